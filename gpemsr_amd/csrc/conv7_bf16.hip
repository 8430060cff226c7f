// Conv2d(32 -> 16, 7x7, stride 1, pad 3) on bf16 NHWC tensors: SpyNet's fourth BasicModule convolution (basicsr
// spynet_arch.BasicModule via R:model/GPEMSR.py:67,99; 32 -> 16 channels at every pyramid level, 512 x 512 at the finest).
//
// On the implicit-GEMM ring kernel (conv_bf16.hip) 16 output channels fill HALF of the 32-row tile of v_mfma_f32_32x32x16_bf16: the
// layer ran at 0.55 PFLOP/s.  Here it runs on v_mfma_f32_16x16x32_bf16, whose shape IS the layer: 16 couts x 16 pixels x K = 32 = one
// filter tap over all 32 input channels, nothing padded.
//   * weights resident in LDS: [49 taps][4 k-groups][16 couts][8] = 50,176 bytes, one 1-KiB fragment per tap;
//   * tile = 16 x 32 output pixels, halo image 22 x 38 pixels x 32 channels (53.5 KB, the swizzled [pixel][32 ch] image of
//     conv_bf16.hip), double buffered by tile and filled by four loader waves with LDS-DMA one whole tile ahead;
//   * each of eight multiplying waves owns two output rows x 32 pixels (four 16 x 16 accumulator tiles = 16 registers) and walks the
//     EIGHT halo rows its two rows read: a pixel fragment of halo row h serves tap row ky = h of the upper output row and ky = h - 1 of
//     the lower one, and the seven weight fragments of a tap row stay in registers for one more halo row -- 112 pixel + 49 weight
//     fragment reads for 196 MFMAs (0.82 ds_read_b128 per MFMA; the LDS array sustains one per 16-cycle MFMA and SIMD);
//   * accumulators are D^T (lane = pixel, registers = 4 consecutive couts): bias, activation, bf16 pack, 8-byte store per lane.
#include "bf16_common.h"

namespace gpemsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct C7Params {
  const unsigned short* x; int n, h, w, ld;
  const unsigned short* wfrag; const float* bias; int act;
  unsigned short* out; int out_ld;
  int tiles_x, tiles_y; unsigned mg_x, mg_y;
  int ntiles;
};

constexpr int C7_TH = 16, C7_HW = 38, C7_HH = 22, C7_HPX = C7_HW * C7_HH;      // halo 22 x 38
constexpr int C7_ABYTES = C7_HPX * 64;                                            // 53,504
constexpr int C7_WBYTES = 49 * 1024;                                              // 50,176

__global__ __launch_bounds__(768, 3) void conv7_c32_cout16_kernel(C7Params P) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const unsigned xsm_lds = xlds_addr(xsm);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T_me = (P.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  auto tile_geo = [&](int j, int& img, int& oy0, int& ox0) {
    int t = (int)blockIdx.x + j * (int)gridDim.x;
    {   // XCD-aware remap (bijective): consecutive logical tiles of concurrently running workgroups share an XCD / L2 (halo rows in common)
      const int q = P.ntiles / 8, r = P.ntiles % 8, xcd = t % 8;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t / 8;
    }
    int tx, ty;
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    xdivmod(t, P.tiles_y, P.mg_y, t, ty);
    img = t; oy0 = ty * C7_TH; ox0 = tx * 32;
  };
  auto tile_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 8) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    const int dtid = tid - 512, dwave = wave - 8;
    const unsigned lds0 = xuni(xsm_lds + (unsigned)dwave * 1024u);
    const unsigned pixb = (unsigned)P.ld * 2u;
    {
      const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.wfrag));
      for (int i = 0; i < 13; ++i) {                    // 49 KiB = 12.25 rounds of 256 x 16 bytes
        const int e = dtid + i * 256;
        if (e < C7_WBYTES / 16) xglds16((unsigned)e * 16u, wp, lds0 + (unsigned)i * 4096u);
      }
    }
    auto issue_tile = [&](int j) {
      int img, oy0, ox0;
      tile_geo(j, img, oy0, ox0);
      const unsigned short* sp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.x + (long long)img * P.h * P.w * P.ld));
      const unsigned la = xuni(lds0 + (unsigned)(C7_WBYTES + (j & 1) * C7_ABYTES));
      char* ab = xsm + C7_WBYTES + (j & 1) * C7_ABYTES;
      constexpr int NSLOT = (C7_HPX * 4 + 255) / 256;   // 14 slots per loader thread (pixel coordinates recomputed per slot: no slot arrays)
#pragma unroll 2
      for (int i = 0; i < NSLOT; ++i) {
        const int e = dtid + i * 256;
        if (e < C7_HPX * 4) {
          const int hp = e >> 2;
          const int hy = hp / C7_HW, hx = hp - hy * C7_HW;
          const int iy = oy0 - 3 + hy, ix = ox0 - 3 + hx;
          if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) {
            const unsigned q = (unsigned)((e & 3) ^ ((hp >> 2) & 3));
            xglds16((unsigned)(iy * P.w + ix) * pixb + 16u * q, sp, la + (unsigned)i * 4096u);
          } else {
            *reinterpret_cast<float4*>(ab + e * 16) = make_float4(0.f, 0.f, 0.f, 0.f);       // zero padding
          }
        }
      }
    };
    issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile_barrier();
    for (int j = 0; j < T_me; ++j) {
      if (j + 1 < T_me) issue_tile(j + 1);              // buffer (j + 1) & 1: read by tile j - 1, left by every wave at the last barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tile_barrier();
    }
    return;
  }

  // ------------------------------------------------ multiplying waves: output rows 2 wave, 2 wave + 1 of the tile ------------------------------------------------
  const int l16 = lane & 15, kg = lane >> 4;           // pixel within a 16-pixel group / k-group (8 input channels) of this lane
  const unsigned wfrag = xsm_lds + (unsigned)(kg * 256 + l16 * 16);
  float b4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b4[i] = P.bias ? P.bias[4 * kg + i] : 0.f;
  const int act = P.act;

  tile_barrier();
  for (int j = 0; j < T_me; ++j) {
    int img, oy0, ox0;
    tile_geo(j, img, oy0, ox0);
    const unsigned abase = xsm_lds + (unsigned)(C7_WBYTES + (j & 1) * C7_ABYTES);
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g) { acc[r][g][0] = b4[0]; acc[r][g][1] = b4[1]; acc[r][g][2] = b4[2]; acc[r][g][3] = b4[3]; }
    bf16x8 wprev[7], wcur[7];
#pragma unroll
    for (int hrow = 0; hrow < 8; ++hrow) {
      if (hrow < 7) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) wcur[kx] = xlds_read16(wfrag + (unsigned)((hrow * 7 + kx) * 1024));
      }
      // pixel fragments of halo row 2 wave + hrow: 16 pixels starting at 16 g + kx, this lane's k-group (swizzled piece)
      const int hp_row = (2 * wave + hrow) * C7_HW + l16;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int hp = hp_row + 16 * g + kx;
          const bf16x8 f = xlds_read16(abase + (unsigned)(hp * 64 + ((kg ^ ((hp >> 2) & 3)) * 16)));
          if (hrow < 7) acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[kx], f, acc[0][g], 0, 0, 0);      // upper row, ky = hrow
          if (hrow >= 1) acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wprev[kx], f, acc[1][g], 0, 0, 0);    // lower row, ky = hrow - 1
        }
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) wprev[kx] = wcur[kx];
    }
    // epilogue: lane = pixel 16 g + l16 of row r, couts 4 kg .. + 3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = oy0 + 2 * wave + r;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int ox = ox0 + 16 * g + l16;
        float v[4] = {acc[r][g][0], acc[r][g][1], acc[r][g][2], acc[r][g][3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (act == GPEMSR_ACT_RELU) v[i] = fmaxf(v[i], 0.f);
          else if (act == GPEMSR_ACT_LRELU) v[i] = fmaxf(v[i], 0.1f * v[i]);
        }
        if (oy < P.h && ox < P.w) {
          unsigned short* op = P.out + (((long long)img * P.h + oy) * P.w + ox) * P.out_ld + 4 * kg;
          *reinterpret_cast<uint2*>(op) = make_uint2(xcvt_pk_bf16(v[0], v[1]), xcvt_pk_bf16(v[2], v[3]));
        }
      }
    }
    tile_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Conv2d(8 -> 32, 7x7, stride 1, pad 3): the FIRST convolution of every SpyNet BasicModule (input = [ref (3), warped supp (3), flow (2)],
// R:model/GPEMSR.py:67,99 via basicsr spynet_arch).  On the ring kernel the 8 input channels were zero-padded to a 16-channel chunk and
// every tap was one K = 16 k-step, half of it zeros (1.24 ms at the finest level).  Here FOUR taps share one v_mfma_f32_16x16x32_bf16:
// k-group g of the instruction (8 of its 32 k slots) is tap 4 j + g with that tap's 8 real channels, so the 49 taps are 13 MFMAs per
// 16 pixels x 16 couts.  The pixel operand of a lane is the 16-byte pixel at (row + ky(tap), x + kx(tap)) of a [halo pixel][8 ch] image
// -- every lane reads its own tap's pixel, no im2col anywhere; taps 49..51 of the last group carry zero weights (their pixel address is
// tap 48's).  Weights resident: [13][2 cout tiles][64 lanes][8] = 26 KB.  Tile 16 x 32 pixels, halo 22 x 38 x 16 B = 13.4 KB, three
// buffers (images are issued two tiles ahead: a tile is only ~3k cycles long); wave = two rows x 32 pixels x 32 couts (32 accumulator
// registers).  The layer is bound by its 64-byte-per-pixel output (0.28 ms at the finest level at 4.8 TB/s).
// x: bf16 NHWC, pixel stride ld >= 8 elements, the first 8 channels of every pixel are used (spynet_prep_bf16's 16-channel tensor).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int C8_ABYTES = C7_HPX * 16;                                            // 13,376
constexpr int C8_WBYTES = 13 * 2 * 1024;                                          // 26,624

__global__ __launch_bounds__(768, 3) void conv7_c8_cout32_kernel(C7Params P) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const unsigned xsm_lds = xlds_addr(xsm);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T_me = (P.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  auto tile_geo = [&](int j, int& img, int& oy0, int& ox0) {
    int t = (int)blockIdx.x + j * (int)gridDim.x;
    {
      const int q = P.ntiles / 8, r = P.ntiles % 8, xcd = t % 8;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t / 8;
    }
    int tx, ty;
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    xdivmod(t, P.tiles_y, P.mg_y, t, ty);
    img = t; oy0 = ty * C7_TH; ox0 = tx * 32;
  };
  auto tile_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 8) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    const int dtid = tid - 512, dwave = wave - 8;
    const unsigned lds0 = xuni(xsm_lds + (unsigned)dwave * 1024u);
    const unsigned pixb = (unsigned)P.ld * 2u;
    {
      const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.wfrag));
      for (int i = 0; i < 7; ++i) {                     // 26 KiB = 6.5 rounds of 256 x 16 bytes
        const int e = dtid + i * 256;
        if (e < C8_WBYTES / 16) xglds16((unsigned)e * 16u, wp, lds0 + (unsigned)i * 4096u);
      }
    }
    int issued = 0, mark[3] = {0, 0, 0};               // DMA instructions this wave has issued; ... when tile j's image (buffer j % 3) was complete
    auto issue_tile = [&](int j, int buf) {            // buf = j % 3 (kept by the caller: no run-time modulo)
      int img, oy0, ox0;
      tile_geo(j, img, oy0, ox0);
      const unsigned short* sp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.x + (long long)img * P.h * P.w * P.ld));
      const unsigned la = xuni(lds0 + (unsigned)(C8_WBYTES + buf * C8_ABYTES));
      char* ab = xsm + C8_WBYTES + buf * C8_ABYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {                     // 836 pixels: 3.3 rounds of 256
        const int hp = dtid + i * 256;
        bool in = false;
        unsigned off = 0u;
        if (hp < C7_HPX) {
          const int hy = hp / C7_HW, hx = hp - hy * C7_HW;
          const int iy = oy0 - 3 + hy, ix = ox0 - 3 + hx;
          in = iy >= 0 && iy < P.h && ix >= 0 && ix < P.w;
          off = (unsigned)(iy * P.w + ix) * pixb;
          if (!in) *reinterpret_cast<float4*>(ab + hp * 16) = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding
        }
        if (in) xglds16(off, sp, la + (unsigned)i * 4096u);
        issued += (__ballot(in) != 0ull) ? 1 : 0;
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) if (b == buf) mark[b] = issued;
    };
    issue_tile(0, 0);
    if (T_me > 1) issue_tile(1, 1);
    xwait_vmcnt(issued - mark[0]);                     // weights + tile 0 (tile 1 may still be in flight)
    tile_barrier();
    int bnext = 2 % 3, bneed = 1;                      // buffer of tile j + 2 / of tile j + 1
    for (int j = 0; j < T_me; ++j) {
      // buffer (j + 2) % 3 was read by tile j - 1, which every multiplying wave left at the previous barrier
      if (j + 2 < T_me) issue_tile(j + 2, bnext);
      if (j + 1 < T_me) {
        int need = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) if (b == bneed) need = mark[b];
        xwait_vmcnt(issued - need);                    // tile j + 1 has landed; tile j + 2 stays in flight
      }
      tile_barrier();
      bnext = bnext == 2 ? 0 : bnext + 1; bneed = bneed == 2 ? 0 : bneed + 1;
    }
    return;
  }

  // ------------------------------------------------ multiplying waves: output rows 2 wave, 2 wave + 1 ------------------------------------------------
  const int l16 = lane & 15, kg = lane >> 4;
  unsigned toff[13];                                   // byte offset of THIS lane's tap (4 j + kg) inside the halo image
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    int t = 4 * j + kg;
    t = t < 49 ? t : 48;                               // (zero weights: any valid pixel)
    toff[j] = (unsigned)(((t / 7) * C7_HW + (t % 7)) * 16);
  }
  const unsigned wfrag = xsm_lds + (unsigned)(lane * 16);
  float b4[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[m][i] = P.bias ? P.bias[16 * m + 4 * kg + i] : 0.f;
  const int act = P.act;

  tile_barrier();
  int buf = 0;
  for (int j = 0; j < T_me; ++j) {
    int img, oy0, ox0;
    tile_geo(j, img, oy0, ox0);
    const unsigned abase = xsm_lds + (unsigned)(C8_WBYTES + buf * C8_ABYTES) + (unsigned)(((2 * wave) * C7_HW + l16) * 16);
    f32x4 acc[2][2][2];                                // [row][pixel group][cout tile]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) { acc[r][g][m][0] = b4[m][0]; acc[r][g][m][1] = b4[m][1]; acc[r][g][m][2] = b4[m][2]; acc[r][g][m][3] = b4[m][3]; }
#pragma unroll
    for (int jj = 0; jj < 13; ++jj) {
      const bf16x8 w0 = xlds_read16(wfrag + (unsigned)((2 * jj) * 1024)), w1 = xlds_read16(wfrag + (unsigned)((2 * jj + 1) * 1024));
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const bf16x8 f = xlds_read16(abase + toff[jj] + (unsigned)((r * C7_HW + 16 * g) * 16));
          acc[r][g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, f, acc[r][g][0], 0, 0, 0);
          acc[r][g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f, acc[r][g][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = oy0 + 2 * wave + r;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int ox = ox0 + 16 * g + l16;
        if (oy < P.h && ox < P.w) {
          unsigned short* op = P.out + (((long long)img * P.h + oy) * P.w + ox) * P.out_ld + 4 * kg;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float v[4] = {acc[r][g][m][0], acc[r][g][m][1], acc[r][g][m][2], acc[r][g][m][3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (act == GPEMSR_ACT_RELU) v[i] = fmaxf(v[i], 0.f);
              else if (act == GPEMSR_ACT_LRELU) v[i] = fmaxf(v[i], 0.1f * v[i]);
            }
            *reinterpret_cast<uint2*>(op + 16 * m) = make_uint2(xcvt_pk_bf16(v[0], v[1]), xcvt_pk_bf16(v[2], v[3]));
          }
        }
      }
    }
    tile_barrier();
    buf = buf == 2 ? 0 : buf + 1;
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_conv7_c32_cout16_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                                            void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && out && n > 0 && h > 0 && w > 0, "conv7_c32_cout16_bf16: null pointer / empty input");
  GP_REQUIRE(ld >= 32 && ld % 8 == 0 && out_ld >= 16 && out_ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wfrag)) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(out) & 7) == 0, "conv7_c32_cout16_bf16: needs ld %% 8 == 0, out_ld %% 4 == 0, 16-byte aligned x / weights, 8-byte aligned out");
  GP_REQUIRE(act == GPEMSR_ACT_NONE || act == GPEMSR_ACT_RELU || act == GPEMSR_ACT_LRELU, "conv7_c32_cout16_bf16: activation %d unsupported", act);
  GP_REQUIRE((long long)h * w * ld * 2 < (1ll << 32), "conv7_c32_cout16_bf16: image too large for 32-bit byte offsets");
  C7Params P{};
  P.x = reinterpret_cast<const unsigned short*>(x); P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = reinterpret_cast<const unsigned short*>(wfrag); P.bias = bias; P.act = act;
  P.out = reinterpret_cast<unsigned short*>(out); P.out_ld = out_ld;
  P.tiles_x = cdiv(w, 32); P.tiles_y = cdiv(h, C7_TH);
  P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y;
  const long long nt = (long long)n * P.tiles_x * P.tiles_y;
  GP_REQUIRE(nt < (1ll << 31), "conv7_c32_cout16_bf16: grid too large");
  P.ntiles = (int)nt;
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_c32_cout16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv7_c32_cout16_bf16: cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
  const int cus = device_cus();
  const int grid = P.ntiles < cus ? P.ntiles : cus;    // persistent: one workgroup per CU (157 KB of LDS)
  const size_t lds = (size_t)C7_WBYTES + 2 * (size_t)C7_ABYTES;
  hipLaunchKernelGGL(conv7_c32_cout16_kernel, dim3(grid), dim3(768), lds, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("conv7_c32_cout16_kernel");
}

extern "C" int gpemsr_conv7_c8_cout32_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                                           void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && wfrag && out && n > 0 && h > 0 && w > 0, "conv7_c8_cout32_bf16: null pointer / empty input");
  GP_REQUIRE(ld >= 8 && ld % 8 == 0 && out_ld >= 32 && out_ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wfrag)) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(out) & 7) == 0, "conv7_c8_cout32_bf16: needs ld %% 8 == 0, out_ld %% 4 == 0, 16-byte aligned x / weights, 8-byte aligned out");
  GP_REQUIRE(act == GPEMSR_ACT_NONE || act == GPEMSR_ACT_RELU || act == GPEMSR_ACT_LRELU, "conv7_c8_cout32_bf16: activation %d unsupported", act);
  GP_REQUIRE((long long)h * w * ld * 2 < (1ll << 32), "conv7_c8_cout32_bf16: image too large for 32-bit byte offsets");
  C7Params P{};
  P.x = reinterpret_cast<const unsigned short*>(x); P.n = n; P.h = h; P.w = w; P.ld = ld;
  P.wfrag = reinterpret_cast<const unsigned short*>(wfrag); P.bias = bias; P.act = act;
  P.out = reinterpret_cast<unsigned short*>(out); P.out_ld = out_ld;
  P.tiles_x = cdiv(w, 32); P.tiles_y = cdiv(h, C7_TH);
  P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y;
  const long long nt = (long long)n * P.tiles_x * P.tiles_y;
  GP_REQUIRE(nt < (1ll << 31), "conv7_c8_cout32_bf16: grid too large");
  P.ntiles = (int)nt;
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_c8_cout32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv7_c8_cout32_bf16: cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
  const int cus = device_cus();
  const int grid = P.ntiles < cus ? P.ntiles : cus;
  const size_t lds = (size_t)C8_WBYTES + 3 * (size_t)C8_ABYTES;
  hipLaunchKernelGGL(conv7_c8_cout32_kernel, dim3(grid), dim3(768), lds, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("conv7_c8_cout32_kernel");
}
