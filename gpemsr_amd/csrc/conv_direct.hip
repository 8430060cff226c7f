// Direct (VALU) convolution for layers whose channel counts are too small for the
// matrix pipe: cout <= 16.  HBM/L2-bound byte work, so the design goal is
// coalesced float4 traffic, not MFMA:
//   * "vec" kernel (cin in {16,64}, ld%4==0): LPP = cin/4 lanes cooperate on one
//     output pixel, each lane owns 4 input channels (one float4 per tap, so a
//     wave reads whole 64..256-B pixel rows), the cross-channel sum is finished
//     with wavefront shuffles (__shfl_xor over the LPP lanes).
//   * scalar kernel (anything else, e.g. the 2->16 stride-4 flow conv): one thread
//     per (pixel, cout).
// Weights are read in the packed [tap][cout][cin_pad8] layout of gpemsr_conv2d
// (cin padded to a multiple of 8) and cached in LDS.
// Replaces: model/GPEMSR.py:70-75 (flowdsconv*), :250 (refmaskconv3), :318 (conv_last),
// decoder.output_layer (model/decoder.py:33), SpyNet's last 16->2 7x7 conv.
#include "common.h"

namespace gpemsr {

struct DirectParams {
  const float* x; int n, h, w, ld, cin, cin_pad;
  const float* weight; const float* bias; int cout, ksize, stride, pad, oh, ow;
  int act; const float* residual; int res_ld; float* out; int out_ld;
  long long npix;
};

template <int LPP, int COUT>
__global__ __launch_bounds__(256) void conv_direct_vec_kernel(DirectParams P) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // [tap][COUT][cin_pad]
  const int ntap = P.ksize * P.ksize;
  for (int i = threadIdx.x; i < ntap * P.cout * P.cin_pad; i += 256) wsm[i] = P.weight[i];
  __syncthreads();
  const int sub = threadIdx.x % LPP;
  const long long per_iter = (long long)gridDim.x * (256 / LPP);
  const long long niter = (P.npix + per_iter - 1) / per_iter;
  for (long long it = 0; it < niter; ++it) {
    // every lane runs every iteration (the shuffles below need full participation)
    const long long pix = it * per_iter + (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    const bool valid = pix < P.npix;
    const long long pp = valid ? pix : 0;
    const int ox = (int)(pp % P.ow);
    const int oy = (int)((pp / P.ow) % P.oh);
    const int img = (int)(pp / ((long long)P.ow * P.oh));
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int ky = 0; ky < P.ksize; ++ky) {
      const int iy = oy * P.stride - P.pad + ky;
      if (iy < 0 || iy >= P.h) continue;
      for (int kx = 0; kx < P.ksize; ++kx) {
        const int ix = ox * P.stride - P.pad + kx;
        if (ix < 0 || ix >= P.w) continue;
        const float4 v = *reinterpret_cast<const float4*>(P.x + (((long long)img * P.h + iy) * P.w + ix) * P.ld + 4 * sub);
        const float* wp = wsm + (ky * P.ksize + kx) * P.cout * P.cin_pad + 4 * sub;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          if (co < P.cout) {
            const float4 wv = *reinterpret_cast<const float4*>(wp + co * P.cin_pad);
            acc[co] = fmaf(v.x, wv.x, acc[co]); acc[co] = fmaf(v.y, wv.y, acc[co]);
            acc[co] = fmaf(v.z, wv.z, acc[co]); acc[co] = fmaf(v.w, wv.w, acc[co]);
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int m = LPP / 2; m >= 1; m >>= 1) acc[co] += __shfl_xor(acc[co], m);
    if (valid) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        if (co < P.cout && (co % LPP) == sub) {
          float v = apply_act(acc[co] + (P.bias ? P.bias[co] : 0.f), P.act);
          if (P.residual) v += P.residual[pix * P.res_ld + co];
          P.out[pix * P.out_ld + co] = v;
        }
      }
    }
  }
}

// 64 -> 1 channels, 3x3, stride 1 (decoder.output_layer and conv_last at 1024^2: pure read traffic, 256 B in per 4 B out).
// 16 lanes own the 64 channels of a pixel (one float4 each); a lane group walks RUN consecutive output columns of one
// row and loads every input pixel ONCE per filter row -- each loaded float4 feeds the three outputs it overlaps -- so a
// pixel is fetched 3 x (RUN+2)/RUN times instead of 9 (the vec kernel above), all of it 256-B coalesced rows.
template <int RUN>
__global__ __launch_bounds__(256) void conv_c64_cout1_kernel(DirectParams P) {
  __shared__ __attribute__((aligned(16))) float wsm[9 * 64];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) wsm[i] = P.weight[(i / 64) * P.cin_pad + (i % 64)];    // cout == 1
  __syncthreads();
  const int sub = threadIdx.x & 15;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(wsm + t * 64 + 4 * sub);
  const int runs_x = (P.w + RUN - 1) / RUN;
  const long long nrun = (long long)P.n * P.h * runs_x;
  const float bias = P.bias ? P.bias[0] : 0.f;
  for (long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); r < ((nrun + 15) / 16) * 16; r += (long long)gridDim.x * 16) {
    const bool live = r < nrun;                       // every lane runs every iteration (shuffles below)
    const long long rr = live ? r : 0;
    const int x0 = (int)(rr % runs_x) * RUN;
    const int oy = (int)((rr / runs_x) % P.h);
    const int img = (int)(rr / ((long long)runs_x * P.h));
    float acc[RUN];
#pragma unroll
    for (int j = 0; j < RUN; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy - 1 + ky;
      if (iy < 0 || iy >= P.h) continue;
      const float* rowp = P.x + (((long long)img * P.h + iy) * P.w) * P.ld + 4 * sub;
#pragma unroll
      for (int c = 0; c < RUN + 2; ++c) {             // input column x0 - 1 + c feeds outputs c-2, c-1, c (taps kx = 2, 1, 0)
        const int ix = x0 - 1 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ix >= 0 && ix < P.w) v = *reinterpret_cast<const float4*>(rowp + (long long)ix * P.ld);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int j = c - kx;
          if (j >= 0 && j < RUN) {
            const float4 w4 = wv[ky * 3 + kx];
            acc[j] = fmaf(v.x, w4.x, acc[j]); acc[j] = fmaf(v.y, w4.y, acc[j]);
            acc[j] = fmaf(v.z, w4.z, acc[j]); acc[j] = fmaf(v.w, w4.w, acc[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RUN; ++j)
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m);
    if (live) {
#pragma unroll
      for (int j = 0; j < RUN; ++j) {
        const int ox = x0 + j;
        if ((j & 15) == sub && ox < P.w) {
          const long long pix = ((long long)img * P.h + oy) * P.w + ox;
          float v = apply_act(acc[j] + bias, P.act);
          if (P.residual) v += P.residual[pix * P.res_ld];
          P.out[pix * P.out_ld] = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv_direct_scalar_kernel(DirectParams P) {
  const long long total = P.npix * P.cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int co = (int)(e % P.cout);
    const long long pix = e / P.cout;
    const int ox = (int)(pix % P.ow);
    const int oy = (int)((pix / P.ow) % P.oh);
    const int img = (int)(pix / ((long long)P.ow * P.oh));
    float acc = 0.f;
    for (int ky = 0; ky < P.ksize; ++ky) {
      const int iy = oy * P.stride - P.pad + ky;
      if (iy < 0 || iy >= P.h) continue;
      for (int kx = 0; kx < P.ksize; ++kx) {
        const int ix = ox * P.stride - P.pad + kx;
        if (ix < 0 || ix >= P.w) continue;
        const float* xp = P.x + (((long long)img * P.h + iy) * P.w + ix) * P.ld;
        const float* wp = P.weight + ((long long)(ky * P.ksize + kx) * P.cout + co) * P.cin_pad;
        for (int ci = 0; ci < P.cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
      }
    }
    float v = apply_act(acc + (P.bias ? P.bias[co] : 0.f), P.act);
    if (P.residual) v += P.residual[pix * P.res_ld + co];
    P.out[pix * P.out_ld + co] = v;
  }
}

// 1 -> COUT (multiple of 64... here any multiple of 4) 3x3 stride-1 "stem" convolution: HBM-write-bound (4*COUT bytes out per
// 4 bytes in).  One lane per (pixel, 4 output channels): the 3x3 input neighbourhood is 9 cached scalar loads shared by
// the 16 lanes of a pixel, weights sit in LDS as [tap][cout], the store is a coalesced float4 (256 B per pixel).
// Used for VGG conv1_1 on the 1-channel 1024^2 images (model/GPEMSR.py:386,390), conv_first, refmaskconv1, indexer stem.
__global__ __launch_bounds__(256) void conv_stem1_kernel(const float* x, int n, int h, int w, const float* weight, const float* bias,
                                                         int cout, int cin_pad, int act, float* out, int out_ld) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // [9][cout] then bias[cout]
  for (int i = threadIdx.x; i < 9 * cout; i += 256) wsm[i] = weight[((long long)(i / cout) * cout + (i % cout)) * cin_pad];
  for (int i = threadIdx.x; i < cout; i += 256) wsm[9 * cout + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  // a thread owns 4 output channels of SRUN consecutive pixels of one row: the 9 weight float4s stay in registers and the
  // 3 x (SRUN+2) input window is read once -- the kernel is then bound by its 256-B-per-pixel output stream
  constexpr int SRUN = 4;
  const int c4 = cout >> 2;
  const int runs = (w + SRUN - 1) / SRUN;
  const long long total = (long long)n * h * runs * c4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c4);
    const long long rp = e / c4;
    const int ox0 = (int)(rp % runs) * SRUN, oy = (int)((rp / runs) % h);
    const long long img = rp / ((long long)runs * h);
    const float* xp = x + img * h * w;
    float4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(wsm + t * cout + 4 * j);
    const float4 bv = *reinterpret_cast<const float4*>(wsm + 9 * cout + 4 * j);
    float win[3][SRUN + 2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy - 1 + ky;
#pragma unroll
      for (int q = 0; q < SRUN + 2; ++q) {
        const int ix = ox0 - 1 + q;
        win[ky][q] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? xp[(long long)iy * w + ix] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < SRUN; ++r) {
      if (ox0 + r >= w) break;
      float4 acc = bv;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = win[ky][r + kx];
          const float4 ww = wv[ky * 3 + kx];
          acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y); acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
        }
      acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
      const long long pix = (img * h + oy) * w + ox0 + r;
      *reinterpret_cast<float4*>(out + pix * out_ld + 4 * j) = acc;
    }
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_conv2d_stem1(const float* x, int n, int h, int w, const float* weight, const float* bias, int cout,
                                   int act, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && weight && out && n > 0 && h > 0 && w > 0, "conv2d_stem1: bad args");
  GP_REQUIRE(cout % 4 == 0 && cout <= 512 && out_ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0), "conv2d_stem1: cout/out alignment");
  const long long total = (long long)n * h * ((w + 3) / 4) * (cout / 4);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(conv_stem1_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), (size_t)10 * cout * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, weight, bias, cout, 8, act, out, out_ld);
  return check_launch("conv_stem1_kernel");
}

extern "C" int gpemsr_conv2d_direct(const float* x, int n, int h, int w, int ld, int cin,
                                    const float* weight, const float* bias, int cout, int ksize, int stride,
                                    int act, const float* residual, int res_ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && weight && out, "conv2d_direct: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && cout <= 16, "conv2d_direct: bad geometry (cout<=16)");
  GP_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1), "conv2d_direct: ksize=%d", ksize);
  GP_REQUIRE(stride == 1 || stride == 2 || stride == 4, "conv2d_direct: stride=%d", stride);
  DirectParams P{};
  P.x = x; P.n = n; P.h = h; P.w = w; P.ld = ld; P.cin = cin; P.cin_pad = (cin + 7) / 8 * 8;
  P.weight = weight; P.bias = bias; P.cout = cout; P.ksize = ksize; P.stride = stride; P.pad = ksize / 2;
  P.oh = (h + 2 * P.pad - ksize) / stride + 1; P.ow = (w + 2 * P.pad - ksize) / stride + 1;
  P.act = act; P.residual = residual; P.res_ld = res_ld; P.out = out; P.out_ld = out_ld;
  P.npix = (long long)n * P.oh * P.ow;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (cin == 16 || cin == 64);
  if (vec_ok && cin == 64 && cout == 1 && ksize == 3 && stride == 1 && w >= 8) {
    constexpr int RUN = 8;
    const long long nrun = (long long)n * h * ((w + RUN - 1) / RUN);
    const long long blocks = (nrun + 15) / 16;
    hipLaunchKernelGGL(conv_c64_cout1_kernel<RUN>, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, P);
    return check_launch("conv_c64_cout1_kernel");
  }
  const size_t lds = (size_t)ksize * ksize * cout * P.cin_pad * sizeof(float);
  if (vec_ok && lds <= 60 * 1024) {
    const int lpp = cin / 4;
    const long long groups = (P.npix + (256 / lpp) - 1) / (256 / lpp);
    const int grid = (int)(groups < 8192 ? groups : 8192);
#define GP_LAUNCH_VEC(L, C) hipLaunchKernelGGL((conv_direct_vec_kernel<L, C>), dim3(grid), dim3(256), lds, st, P)
    if (lpp == 16) { if (cout == 1) GP_LAUNCH_VEC(16, 1); else if (cout <= 2) GP_LAUNCH_VEC(16, 2); else GP_LAUNCH_VEC(16, 16); }
    else           { if (cout == 1) GP_LAUNCH_VEC(4, 1);  else if (cout <= 2) GP_LAUNCH_VEC(4, 2);  else GP_LAUNCH_VEC(4, 16); }
#undef GP_LAUNCH_VEC
    return check_launch("conv_direct_vec_kernel");
  }
  const long long total = P.npix * cout;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(conv_direct_scalar_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, P);
  return check_launch("conv_direct_scalar_kernel");
}
