// One-output-channel convolutions of the bf16 path over 64 bf16 input channels, as "tap partial products" on the matrix cores:
//
//   P[tap][pixel] = sum_c W[tap][c] * x[pixel][c]          one v_mfma_f32_32x32x16_bf16 row tile: the taps are the 32 rows
//   out[o]        = sum of P over the (pixel, tap) pairs that reach o        (LDS gather, fp32)
//
// so the 64-channel tensor is read from HBM exactly once (the kernels are HBM-bound: 128 B in per pixel, 4 B out) and the weights keep
// fp32 accuracy as bf16 hi + lo halves (two row groups of the same MFMA, or two MFMAs into one accumulator).
//
//   tap_sum_kernel<false>  Conv2d(64 -> 1, 3x3, pad 1) [+ act] [+ fp32 residual]: conv_last (model/GPEMSR.py:318,455) and
//                          refmodel.decoder.output_layer when it is not preceded by the 64 -> 64 up-block.  9 taps.
//   tap_sum_kernel<true>   ConvTranspose2d(64 -> 64, k3 s2 p1 op1) followed by Conv2d(64 -> 1, 3x3, pad 1) with nothing in between
//                          (the last up-block and the output layer of the VQGAN decoder, model/vqgan.py decoder tail): the two are
//                          ONE linear operator, a stride-2 transposed convolution with a 5x5 kernel and one output channel
//                          (o = 2 i + t - 2, t in 0..4), whose 25 taps are composed on the host (packing.pack_upconv_out).  The
//                          1024^2 x 64 intermediate is never formed.  Two corrections make it exact: the layered form pads the
//                          INTERMEDIATE with zeros, so (a) the routes through intermediate row / column -1 (input row / column 0 with
//                          the k = 0 taps of the transposed kernel) are subtracted again for output row / column 0, and (b) the
//                          bias of the up-block reaches an output only through the taps of the 3x3 that stay inside the image.
#include "bf16_common.h"

namespace gpemsr {

typedef unsigned short tbf16_t;

struct TapParams {
  const void* xv;          // bf16 or fp32 NHWC input, 64 channels
  int n, h, w, ld;
  const void* wfrag;       // bf16 input: [sets][4 k-steps][64 lanes] bf16x8 A-operand fragments (rows = taps, hi / lo halves);
                           // fp32 input: [32 k-steps][64 lanes] floats (v_mfma_f32_32x32x2_f32: lane l = row l % 32, k = l / 32)
  const float* cst;        // 3x3: [0] = bias (or null).  up+out: [dy*3+dx] = sum_co b1[co] w2[co][dy][dx], [9] = b2, then
                           //      [10 + ...] = Wy0 5x64 | Wx0 5x64 | Wc 64 fp32 (border routes, see above)
  int act;
  const float* residual;
  int res_ld;
  float* out;
  int out_ld;
  unsigned char* out_u8;   // 3x3 form, optional: the reference's tensor2img (util/util.py:145-163: clamp to [0,1], x255, round half even)
                           // of the same value, [n][h][w] -- the uint8 image leaves the LAST kernel of the network, no extra pass
  int tiles_x, tiles_y;
  long long total;         // tiles
};

__device__ __forceinline__ float tdot64(const float* xp, const float* wp) {
  float s = 0.f;
#pragma unroll 4
  for (int c4 = 0; c4 < 16; ++c4) {
    const float4 v = *reinterpret_cast<const float4*>(xp + 4 * c4), w4 = *reinterpret_cast<const float4*>(wp + 4 * c4);
    s = fmaf(v.x, w4.x, s); s = fmaf(v.y, w4.y, s); s = fmaf(v.z, w4.z, s); s = fmaf(v.w, w4.w, s);
  }
  return s;
}
__device__ __forceinline__ float tdot64(const tbf16_t* xp, const float* wp) {
  float s = 0.f;
#pragma unroll 2
  for (int c8 = 0; c8 < 8; ++c8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xp + 8 * c8);
    const float4 w0 = *reinterpret_cast<const float4*>(wp + 8 * c8), w1 = *reinterpret_cast<const float4*>(wp + 8 * c8 + 4);
    s = fmaf(xbf_lo(v.x), w0.x, s); s = fmaf(xbf_hi(v.x), w0.y, s); s = fmaf(xbf_lo(v.y), w0.z, s); s = fmaf(xbf_hi(v.y), w0.w, s);
    s = fmaf(xbf_lo(v.z), w1.x, s); s = fmaf(xbf_hi(v.z), w1.y, s); s = fmaf(xbf_lo(v.w), w1.z, s); s = fmaf(xbf_hi(v.w), w1.w, s);
  }
  return s;
}

// TI = tbf16_t: v_mfma_f32_32x32x16_bf16 with hi + lo weight halves.  TI = float (the exact-fp32 path): v_mfma_f32_32x32x2_f32, a lane's
// operand values are channels 2 ks + (lane / 32) of its pixel, picked out of the pixel's sixteen float4 loads.
template <bool UP, typename TI>
__global__ __launch_bounds__(256, 2) void tap_sum_kernel(TapParams P) {
  constexpr bool F32 = sizeof(TI) == 4;
  const TI* const Px = reinterpret_cast<const TI*>(P.xv);
  constexpr int TAPS = UP ? 25 : 9;
  constexpr int HR = UP ? 10 : 18;          // halo rows per tile; 64 halo columns; the tile owns (HR - 2) x 62 pixels
  constexpr int TR = HR - 2, TC = 62;
  constexpr int PER = HR * 2 / 4;           // 32-pixel column tiles of the MFMA per wave
  constexpr int SETS = UP ? 2 : 1;
  __shared__ float ps[TAPS * HR * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lp = lane & 31, hf = lane >> 5;
  // XCD-aware tile order: the 8 XCDs each take one contiguous run of tiles (neighbours share halo rows in that XCD's L2)
  long long tile;
  {
    const long long q = P.total / 8, r = P.total % 8, bid = blockIdx.x;
    const long long xcd = bid % 8, idx = bid / 8;
    tile = xcd * q + (xcd < r ? xcd : r) + idx;
  }
  const int tx = (int)(tile % P.tiles_x), ty = (int)((tile / P.tiles_x) % P.tiles_y), img = (int)(tile / ((long long)P.tiles_x * P.tiles_y));
  const int h = P.h, w = P.w;
  const int y0 = ty * TR - 1, x0 = tx * TC - 1;

  auto store_rows = [&](const f32x16& acc, int mt) {
    float* pp = ps + (mt >> 1) * 64 + (mt & 1) * 32 + lp;
    if (!UP && !F32) {        // rows 0..8 = hi halves, rows 16..24 = lo halves of the same 9 taps
#pragma unroll
      for (int z = 0; z < 4; ++z) pp[(4 * hf + z) * HR * 64] = acc[z] + acc[z + 8];
      if (hf == 0) pp[8 * HR * 64] = acc[4] + acc[12];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + hf * 4 + (r & 3);
        if (row < TAPS) pp[row * HR * 64] = acc[r];
      }
    }
  };
  if (!F32) {
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(P.wfrag);
    bf16x8 wa[SETS][4];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wa[s][ks] = wf[(s * 4 + ks) * 64 + lane];

    bf16x8 xb[PER][4];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int mt = wave + 4 * i, gy = y0 + (mt >> 1), gx = x0 + (mt & 1) * 32 + lp;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const tbf16_t* p = reinterpret_cast<const tbf16_t*>(P.xv) + (((long long)img * h + (ok ? gy : 0)) * w + (ok ? gx : 0)) * P.ld + hf * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) v = *reinterpret_cast<const bf16x8*>(p + ks * 16);
        xb[i][ks] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][ks], xb[i][ks], acc, 0, 0, 0);
        if (SETS == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[SETS - 1][ks], xb[i][ks], acc, 0, 0, 0);
      }
      store_rows(acc, wave + 4 * i);
    }
  } else {
    const float* wf = reinterpret_cast<const float*>(P.wfrag);
    float wa[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wa[ks] = wf[ks * 64 + lane];
#pragma unroll 1
    for (int i = 0; i < PER; ++i) {
      const int mt = wave + 4 * i, gy = y0 + (mt >> 1), gx = x0 + (mt & 1) * 32 + lp;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const float* p = reinterpret_cast<const float*>(P.xv) + (((long long)img * h + (ok ? gy : 0)) * w + (ok ? gx : 0)) * P.ld;
      float4 xv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(p + 4 * j);
        xv[j] = v;
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {       // k-steps 2 j (channels 4 j + hf) and 2 j + 1 (channels 4 j + 2 + hf)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * j], hf ? xv[j].y : xv[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * j + 1], hf ? xv[j].w : xv[j].z, acc, 0, 0, 0);
      }
      store_rows(acc, mt);
    }
  }
  __syncthreads();

  if (!UP) {
    const float bs0 = P.cst ? P.cst[0] : 0.f;
    for (int e = threadIdx.x; e < TR * TC; e += 256) {
      const int ly = e / TC, lx = e - ly * TC;
      const int oy = ty * TR + ly, ox = tx * TC + lx;
      if (oy >= h || ox >= w) continue;
      float s = bs0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) s += ps[((ky * 3 + kx) * HR + ly + ky) * 64 + lx + kx];
      s = apply_act(s, P.act);
      const long long pix = ((long long)img * h + oy) * w + ox;
      if (P.residual) s += P.residual[pix * P.res_ld];
      P.out[pix * P.out_ld] = s;
      if (P.out_u8) P.out_u8[pix] = (unsigned char)rintf(fminf(fmaxf(s, 0.f), 1.f) * 255.0f);
    }
  } else {
    const int OH = 2 * h, OW = 2 * w;
    const float* fix = P.cst + 10;
    float bs[10];
#pragma unroll
    for (int z = 0; z < 10; ++z) bs[z] = P.cst[z];
    for (int e = threadIdx.x; e < TR * TC; e += 256) {
      const int li = e / TC, lj = e - li * TC;
      const int i = ty * TR + li, j = tx * TC + lj;
      if (i >= h || j >= w) continue;
      float o[2][2];
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          float s = 0.f;
#pragma unroll
          for (int a = 0; a < 3 - py; ++a)          // input row i + 1 - a carries tap row py + 2 a
#pragma unroll
            for (int b = 0; b < 3 - px; ++b)
              s += ps[(((py + 2 * a) * 5 + px + 2 * b) * HR + li + 2 - a) * 64 + lj + 2 - b];
          o[py][px] = s;
        }
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int oy = 2 * i + py, ox = 2 * j + px;
          float s = o[py][px] + bs[9];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const bool in = oy + dy - 1 >= 0 && oy + dy - 1 < OH && ox + dx - 1 >= 0 && ox + dx - 1 < OW;
              s += in ? bs[dy * 3 + dx] : 0.f;
            }
          if (oy == 0) {          // routes through intermediate row -1: input row 0, k_y = 0, d_y = 0
            for (int t = ox & 1; t < 5; t += 2) {
              const int jj = (ox + 2 - t) >> 1;
              if (jj >= 0 && jj < w) s -= tdot64(Px + (((long long)img * h) * w + jj) * P.ld, fix + t * 64);
            }
          }
          if (ox == 0) {
            for (int t = oy & 1; t < 5; t += 2) {
              const int ii = (oy + 2 - t) >> 1;
              if (ii >= 0 && ii < h) s -= tdot64(Px + (((long long)img * h + ii) * w) * P.ld, fix + 320 + t * 64);
            }
          }
          if (oy == 0 && ox == 0) s += tdot64(Px + ((long long)img * h) * w * P.ld, fix + 640);
          o[py][px] = s;
        }
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const long long pix = ((long long)img * OH + 2 * i + py) * OW + 2 * j;
        if (P.out_ld == 1) *reinterpret_cast<float2*>(P.out + pix) = make_float2(o[py][0], o[py][1]);
        else { P.out[pix * P.out_ld] = o[py][0]; P.out[(pix + 1) * P.out_ld] = o[py][1]; }
      }
    }
  }
}

// Conv2d(16 -> 2, 7x7, pad 3) + fp32 residual: the last layer of every SpyNet level (basicsr SpyNet BasicModule, conv index 8:
// flow update = conv(...) + up-sampled coarser flow).  On the implicit-GEMM kernel two output channels fill 2 of 32 MFMA rows
// (42 TFLOP/s, 1.6 ms at the finest level).  Here the matrix pipe forms ROW SUMS
//     R[ky][co][row][x] = sum_{kx, c} W[co][c][ky][kx] * in[row][x + kx - 3][c]          rows = (ky, co): 14 of 32, K = 7 k-steps (one per kx)
// for every halo row of a 16 x 32 tile, with the input fragments loaded straight from global memory (a k-step's operand is the
// same 1 KiB of pixels shifted by one, so the seven loads of a row hit L1), and the output is the vertical 7-sum
//     out[co][y][x] = bias[co] + sum_ky R[ky][co][y + ky][x]   (+ residual)
// from LDS.  Weights as bf16 hi halves in rows 0..13 and lo halves in rows 16..29 of the same MFMA (fp32-accurate weights for free).
struct Row7Params {
  const void* xv;          // bf16 or fp32 NHWC input, 16 channels
  int n, h, w, ld;
  const void* wfrag;       // bf16: [7 kx][64 lanes] bf16x8 (hi rows 0..13, lo rows 16..29); fp32: [7 kx][8 k-steps][64 lanes] floats
  const float* bias;       // [2] or null
  const float* residual;
  int res_ld;
  float* out;
  int out_ld;
  int tiles_x, tiles_y;
  long long total;
};

template <typename TI>
__global__ __launch_bounds__(256, 2) void rowsum7_kernel(Row7Params P) {
  constexpr bool F32 = sizeof(TI) == 4;
  constexpr int TR = 16, TC = 32, HR = TR + 6;
  __shared__ float rs[14 * HR * TC];                   // 39,424 B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lp = lane & 31, hf = lane >> 5;
  long long tile;
  {
    const long long q = P.total / 8, r = P.total % 8, bid = blockIdx.x;
    const long long xcd = bid % 8, idx = bid / 8;
    tile = xcd * q + (xcd < r ? xcd : r) + idx;
  }
  const int tx = (int)(tile % P.tiles_x), ty = (int)((tile / P.tiles_x) % P.tiles_y), img = (int)(tile / ((long long)P.tiles_x * P.tiles_y));
  const int h = P.h, w = P.w;
  const int y0 = ty * TR - 3, x0 = tx * TC;
  if (!F32) {
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(P.wfrag);
    bf16x8 wa[7];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) wa[kx] = wf[kx * 64 + lane];
    for (int hr = wave; hr < HR; hr += 4) {
      const int gy = y0 + hr;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const bool rowok = gy >= 0 && gy < h;               // wave-uniform
      if (rowok) {
        bf16x8 xb[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int gx = x0 + lp + kx - 3;
          const bool ok = gx >= 0 && gx < w;
          bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (ok) v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const tbf16_t*>(P.xv) + (((long long)img * h + gy) * w + gx) * P.ld + hf * 8);
          xb[kx] = v;
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kx], xb[kx], acc, 0, 0, 0);
      }
      // rows (ky, co) = 2 ky + co: lanes hf = 0 hold rows 0-3 (registers 0-3) and 8-11 (4-7), hf = 1 rows 4-7 and 12-13; lo halves 8 registers on
      float* rp = rs + hr * TC + lp;
#pragma unroll
      for (int z = 0; z < 4; ++z) rp[(4 * hf + z) * HR * TC] = acc[z] + acc[z + 8];
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        const int row = 8 + 4 * hf + z;
        if (row < 14) rp[row * HR * TC] = acc[4 + z] + acc[12 + z];
      }
    }
  } else {
    // exact-fp32 path: v_mfma_f32_32x32x2_f32, 8 k-steps per kx (lane's operand = channel 2 ks + hf of its pixel), fp32 weights
    const float* wf = reinterpret_cast<const float*>(P.wfrag);
    for (int hr = wave; hr < HR; hr += 4) {
      const int gy = y0 + hr;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const bool rowok = gy >= 0 && gy < h;
      if (rowok) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int gx = x0 + lp + kx - 3;
          const bool ok = gx >= 0 && gx < w;
          const float* p = reinterpret_cast<const float*>(P.xv) + (((long long)img * h + gy) * w + (ok ? gx : 0)) * P.ld;
          float4 xv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { float4 v = make_float4(0.f, 0.f, 0.f, 0.f); if (ok) v = *reinterpret_cast<const float4*>(p + 4 * j); xv[j] = v; }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[(kx * 8 + 2 * j) * 64 + lane], hf ? xv[j].y : xv[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[(kx * 8 + 2 * j + 1) * 64 + lane], hf ? xv[j].w : xv[j].z, acc, 0, 0, 0);
          }
        }
      }
      float* rp = rs + hr * TC + lp;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = (r >> 2) * 8 + hf * 4 + (r & 3);
        if (row < 14) rp[row * HR * TC] = acc[r];
      }
    }
  }
  __syncthreads();
  const float b0 = P.bias ? P.bias[0] : 0.f, b1 = P.bias ? P.bias[1] : 0.f;
  for (int e = threadIdx.x; e < TR * TC; e += 256) {
    const int ly = e / TC, lx = e - ly * TC;
    const int oy = ty * TR + ly, ox = x0 + lx;
    if (oy >= h || ox >= w) continue;
    float s0 = b0, s1 = b1;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      s0 += rs[((2 * ky) * HR + ly + ky) * TC + lx];
      s1 += rs[((2 * ky + 1) * HR + ly + ky) * TC + lx];
    }
    const long long pix = ((long long)img * h + oy) * w + ox;
    if (P.residual) { s0 += P.residual[pix * P.res_ld]; s1 += P.residual[pix * P.res_ld + 1]; }
    P.out[pix * P.out_ld] = s0; P.out[pix * P.out_ld + 1] = s1;
  }
}

template <bool UP, typename TI>
static int launch_tap(TapParams& P, hipStream_t st) {
  constexpr int TR = UP ? 8 : 16;
  P.tiles_x = (P.w + 61) / 62;
  P.tiles_y = (P.h + TR - 1) / TR;
  P.total = (long long)P.n * P.tiles_x * P.tiles_y;
  if (P.total >= (1ll << 31)) return fail(GPEMSR_EINVAL, "tap_sum: too many tiles");
  hipLaunchKernelGGL((tap_sum_kernel<UP, TI>), dim3((unsigned)P.total), dim3(256), 0, st, P);
  return check_launch(UP ? "tap_sum_kernel<up>" : "tap_sum_kernel<3x3>");
}

}  // namespace gpemsr

using namespace gpemsr;
#define A16(p) ((reinterpret_cast<uintptr_t>(p) & 15) == 0)

extern "C" int gpemsr_conv_c64_cout1_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                                          const float* residual, int res_ld, float* out, int out_ld, uint8_t* out_u8, void* stream) {
  GP_REQUIRE(x && wfrag && out, "conv_c64_cout1_bf16: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 64 && ld % 8 == 0 && A16(x) && A16(wfrag) && out_ld >= 1, "conv_c64_cout1_bf16: bad geometry / alignment");
  TapParams P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = wfrag;
  P.cst = bias;
  P.act = act; P.residual = residual; P.res_ld = res_ld; P.out = out; P.out_ld = out_ld; P.out_u8 = out_u8;
  return launch_tap<false, tbf16_t>(P, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int gpemsr_conv_c64_cout1_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* bias, int act,
                                         const float* residual, int res_ld, float* out, int out_ld, uint8_t* out_u8, void* stream) {
  GP_REQUIRE(x && wfrag && out, "conv_c64_cout1_f32: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 64 && ld % 4 == 0 && A16(x) && A16(wfrag) && out_ld >= 1, "conv_c64_cout1_f32: bad geometry / alignment");
  TapParams P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld; P.wfrag = wfrag; P.cst = bias;
  P.act = act; P.residual = residual; P.res_ld = res_ld; P.out = out; P.out_ld = out_ld; P.out_u8 = out_u8;
  return launch_tap<false, float>(P, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int gpemsr_upconv_out_c64_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* consts, float* out,
                                          int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && consts && out, "upconv_out_c64_bf16: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 64 && ld % 8 == 0 && A16(x) && A16(wfrag) && out_ld >= 1 && (out_ld > 1 || (reinterpret_cast<uintptr_t>(out) & 7) == 0)
             && (reinterpret_cast<uintptr_t>(consts) & 7) == 0, "upconv_out_c64_bf16: bad geometry / alignment");
  TapParams P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = wfrag;
  P.cst = consts;
  P.out = out; P.out_ld = out_ld;
  return launch_tap<true, tbf16_t>(P, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int gpemsr_upconv_out_c64_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* consts, float* out,
                                         int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && consts && out, "upconv_out_c64_f32: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 64 && ld % 4 == 0 && A16(x) && A16(wfrag) && out_ld >= 1 && (out_ld > 1 || (reinterpret_cast<uintptr_t>(out) & 7) == 0)
             && (reinterpret_cast<uintptr_t>(consts) & 7) == 0, "upconv_out_c64_f32: bad geometry / alignment");
  TapParams P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld; P.wfrag = wfrag; P.cst = consts; P.out = out; P.out_ld = out_ld;
  return launch_tap<true, float>(P, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int gpemsr_conv7_c16_cout2_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, const float* residual,
                                           int res_ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && out, "conv7_c16_cout2_bf16: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 16 && ld % 8 == 0 && A16(x) && A16(wfrag) && out_ld >= 2 && (!residual || res_ld >= 2),
             "conv7_c16_cout2_bf16: bad geometry / alignment");
  Row7Params P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = wfrag; P.bias = bias; P.residual = residual; P.res_ld = res_ld; P.out = out; P.out_ld = out_ld;
  P.tiles_x = (w + 31) / 32; P.tiles_y = (h + 15) / 16;
  P.total = (long long)n * P.tiles_x * P.tiles_y;
  if (P.total >= (1ll << 31)) return fail(GPEMSR_EINVAL, "conv7_c16_cout2_bf16: too many tiles");
  hipLaunchKernelGGL(rowsum7_kernel<tbf16_t>, dim3((unsigned)P.total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("rowsum7_kernel");
}

extern "C" int gpemsr_conv7_c16_cout2_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* bias, const float* residual,
                                          int res_ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && out, "conv7_c16_cout2_f32: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && ld >= 16 && ld % 4 == 0 && A16(x) && out_ld >= 2 && (!residual || res_ld >= 2), "conv7_c16_cout2_f32: bad geometry / alignment");
  Row7Params P{};
  P.xv = x; P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = wfrag; P.bias = bias; P.residual = residual; P.res_ld = res_ld; P.out = out; P.out_ld = out_ld;
  P.tiles_x = (w + 31) / 32; P.tiles_y = (h + 15) / 16;
  P.total = (long long)n * P.tiles_x * P.tiles_y;
  if (P.total >= (1ll << 31)) return fail(GPEMSR_EINVAL, "conv7_c16_cout2_f32: too many tiles");
  hipLaunchKernelGGL(rowsum7_kernel<float>, dim3((unsigned)P.total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("rowsum7_kernel");
}
