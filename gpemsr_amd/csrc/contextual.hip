// Contextual (CX) loss forward of the stage-3 training step (train_stage3.py:352-359) -- the element-wise / reduction
// part around the one big GEMM.  Reference arithmetic: model/contextual.py
//   compute_cosine_distance :115-138   y_mu = mean_{n,h,w} y;  x^ = normalize_c(x - y_mu), y^ = normalize_c(y - y_mu);
//                                      dist[n,i,j] = clamp(1 - <x^[n,i,:], y^[n,j,:]>, 0)
//   compute_relative_distance :109-112 dist~ = dist / (min_j dist + 1e-5)
//   compute_cx :103-106                w = exp((1 - dist~)/h);  cx = w / (sum_j w + 1e-5)
//   contextual_loss :44-52             r[n,j] = max_i cx[n,i,j] (i* = argmax);  c[n,j] = exp((1 - dist[n,i*,j])/h);
//                                      CX[n] = sum_j r*c / sum_j c;  loss = mean_n(-log(CX[n] + 1e-5))
// Data layout: features are NHWC, i.e. already the [pixel][channel] rows the GEMM wants; the similarity matrix
// S[n][i][j] = <x^_i, y^_j> comes from gpemsr_conv2d (1x1, per-image "weights" = y^, conv_mfma.hip) with j as the
// channel axis, so rows (i) are contiguous: the row pass is a coalesced one-workgroup-per-row kernel and the column
// max walks rows with consecutive lanes on consecutive j.  Everything here is HBM-bound (P*P floats per image read
// twice, written once); nothing is reshaped into a GEMM.
#include "common.h"
#include <math.h>

namespace gpemsr {

__device__ __forceinline__ float cx_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float cx_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// per-channel partial sums of y over a slab of pixels: part[block][c]  (fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void cx_chan_partial_kernel(const float* y, long long pixels, int c, int ld, int slab, float* part) {
  const long long p0 = (long long)blockIdx.x * slab;
  const long long p1 = p0 + slab < pixels ? p0 + slab : pixels;
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float s = 0.f;
    for (long long p = p0; p < p1; ++p) s += y[p * ld + ch];
    part[(long long)blockIdx.x * c + ch] = s;
  }
}
__global__ __launch_bounds__(256) void cx_chan_final_kernel(const float* part, int nblk, int c, float inv_count, float* mean) {
  for (int ch = blockIdx.x * 256 + threadIdx.x; ch < c; ch += gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(long long)b * c + ch];
    mean[ch] = s * inv_count;
  }
}

// one wave per pixel: out = (x - mu) / max(||x - mu||_2, 1e-12)   (F.normalize(p=2, dim=1), eps 1e-12)
__global__ __launch_bounds__(256) void cx_center_normalize_kernel(const float* x, const float* mu, long long pixels, int c, int ld,
                                                                  float* out, int out_ld) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= pixels) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + p * ld;
  float ss = 0.f;
  for (int ch = lane; ch < c; ch += 64) { const float v = xr[ch] - mu[ch]; ss += v * v; }
  ss = cx_wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  float* orow = out + p * out_ld;
  for (int ch = lane; ch < c; ch += 64) orow[ch] = (xr[ch] - mu[ch]) / nrm;
}

// one workgroup per row of S (cols = P_y): cx row (Eq. 3-4 of the paper)
template <int MAXV>
__global__ __launch_bounds__(256) void cx_rows_kernel(const float* sim, int cols, float inv_h, float* cx) {
  const float* row = sim + (long long)blockIdx.x * cols;
  float* orow = cx + (long long)blockIdx.x * cols;
  const int c4 = cols >> 2;
  float4 v[MAXV];
  float m = INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      float4 s = *reinterpret_cast<const float4*>(row + 4 * e);
      s.x = fmaxf(1.f - s.x, 0.f); s.y = fmaxf(1.f - s.y, 0.f); s.z = fmaxf(1.f - s.z, 0.f); s.w = fmaxf(1.f - s.w, 0.f);
      v[i] = s;
      m = fminf(fminf(fminf(m, s.x), fminf(s.y, s.z)), s.w);
    }
  }
  __shared__ float red[4];
  m = cx_wave_min(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  __syncthreads();
  const float den = m + 1e-5f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      v[i].x = expf((1.f - v[i].x / den) * inv_h); v[i].y = expf((1.f - v[i].y / den) * inv_h);
      v[i].z = expf((1.f - v[i].z / den) * inv_h); v[i].w = expf((1.f - v[i].w / den) * inv_h);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  sum = cx_wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]) + 1e-5f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      v[i].x /= sum; v[i].y /= sum; v[i].z /= sum; v[i].w /= sum;
      *reinterpret_cast<float4*>(orow + 4 * e) = v[i];
    }
  }
}

// column max over i of cx[n,i,j] (first maximum wins) in slabs of rows, then the slabs are merged in order
__global__ __launch_bounds__(256) void cx_colmax_partial_kernel(const float* cx, int rows, int cols, int slab, float* pmax, int* pidx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.z, sb = blockIdx.y;
  if (j >= cols) return;
  const int i0 = sb * slab, i1 = i0 + slab < rows ? i0 + slab : rows;
  const float* base = cx + (long long)n * rows * cols + j;
  float best = -INFINITY; int bi = i0;
  for (int i = i0; i < i1; ++i) {
    const float v = base[(long long)i * cols];
    if (v > best) { best = v; bi = i; }
  }
  const long long o = ((long long)n * gridDim.y + sb) * cols + j;
  pmax[o] = best; pidx[o] = bi;
}
__global__ __launch_bounds__(256) void cx_colmax_final_kernel(const float* pmax, const int* pidx, const float* sim, int rows, int cols,
                                                              int nslab, float inv_h, float* rmax, float* cw) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (j >= cols) return;
  float best = -INFINITY; int bi = 0;
  for (int sb = 0; sb < nslab; ++sb) {
    const long long o = ((long long)n * nslab + sb) * cols + j;
    if (pmax[o] > best) { best = pmax[o]; bi = pidx[o]; }
  }
  const float d = fmaxf(1.f - sim[((long long)n * rows + bi) * cols + j], 0.f);
  rmax[(long long)n * cols + j] = best;
  cw[(long long)n * cols + j] = expf((1.f - d) * inv_h);
}

// one workgroup per image: CX[n] = sum_j r*c / sum_j c
__global__ __launch_bounds__(256) void cx_image_kernel(const float* rmax, const float* cw, int cols, float* cxn) {
  const float* r = rmax + (long long)blockIdx.x * cols;
  const float* c = cw + (long long)blockIdx.x * cols;
  float a = 0.f, b = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) { a += r[j] * c[j]; b += c[j]; }
  __shared__ float ra[4], rb[4];
  a = cx_wave_sum(a); b = cx_wave_sum(b);
  if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = a; rb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) cxn[blockIdx.x] = ((ra[0] + ra[1]) + (ra[2] + ra[3])) / ((rb[0] + rb[1]) + (rb[2] + rb[3]));
}
__global__ void cx_loss_kernel(const float* cxn, int n, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += -logf(cxn[i] + 1e-5f);
    loss[0] = s / (float)n;
  }
}

// (x - mean_c) / std_c on a 3-channel NHWC image (ContextualLoss.forward, model/contextual.py:222-224)
__global__ __launch_bounds__(256) void normalize3_kernel(const float* x, long long pixels, int ld, float m0, float m1, float m2,
                                                         float s0, float s1, float s2, float* out, int out_ld) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const float* xr = x + p * ld;
    float* o = out + p * out_ld;
    o[0] = (xr[0] - m0) / s0; o[1] = (xr[1] - m1) / s1; o[2] = (xr[2] - m2) / s2;
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_cx_channel_mean(const float* y, int64_t pixels, int c, int ld, float* workspace, int64_t workspace_floats,
                                      float* mean, void* stream) {
  GP_REQUIRE(y && workspace && mean && pixels > 0 && c > 0 && ld >= c, "cx_channel_mean: bad args");
  const int slab = 1024;
  const long long nblk = (pixels + slab - 1) / slab;
  GP_REQUIRE(nblk * c <= workspace_floats && nblk < (1 << 30), "cx_channel_mean: workspace too small (need %lld floats)", nblk * c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cx_chan_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, st, y, (long long)pixels, c, ld, slab, workspace);
  hipLaunchKernelGGL(cx_chan_final_kernel, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, st, workspace, (int)nblk, c,
                     1.f / (float)pixels, mean);
  return check_launch("cx_channel_mean");
}

extern "C" int gpemsr_cx_center_normalize(const float* x, const float* mean, int64_t pixels, int c, int ld, float* out, int out_ld,
                                          void* stream) {
  GP_REQUIRE(x && mean && out && pixels > 0 && c > 0 && ld >= c && out_ld >= c, "cx_center_normalize: bad args");
  hipLaunchKernelGGL(cx_center_normalize_kernel, dim3((unsigned)((pixels + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, mean, (long long)pixels, c, ld, out, out_ld);
  return check_launch("cx_center_normalize");
}

extern "C" int gpemsr_cx_rows(const float* sim, int64_t rows, int cols, float band_width, float* cx, void* stream) {
  GP_REQUIRE(sim && cx && rows > 0 && rows < (1ll << 31) && band_width > 0.f, "cx_rows: bad args");
  GP_REQUIRE(cols % 4 == 0 && cols <= 256 * 4 * 16, "cx_rows: cols=%d unsupported (multiple of 4, <= 16384)", cols);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (cols <= 256 * 4 * 4) hipLaunchKernelGGL(cx_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, st, sim, cols, 1.f / band_width, cx);
  else hipLaunchKernelGGL(cx_rows_kernel<16>, dim3((unsigned)rows), dim3(256), 0, st, sim, cols, 1.f / band_width, cx);
  return check_launch("cx_rows");
}

extern "C" int gpemsr_cx_reduce(const float* cx, const float* sim, int n, int rows, int cols, float band_width, float* workspace,
                                int64_t workspace_floats, float* rmax, float* cw, float* cx_image, float* loss, void* stream) {
  GP_REQUIRE(cx && sim && workspace && rmax && cw && cx_image && loss && n > 0 && rows > 0 && cols > 0 && band_width > 0.f,
             "cx_reduce: bad args");
  const int slab = 128;
  const int nslab = (rows + slab - 1) / slab;
  GP_REQUIRE(2ll * n * nslab * cols <= workspace_floats, "cx_reduce: workspace too small (need %lld floats)", 2ll * n * nslab * cols);
  GP_REQUIRE(nslab <= 65535 && n <= 65535, "cx_reduce: grid too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* pmax = workspace;
  int* pidx = reinterpret_cast<int*>(workspace + (long long)n * nslab * cols);
  const unsigned gx = (unsigned)((cols + 255) / 256);
  hipLaunchKernelGGL(cx_colmax_partial_kernel, dim3(gx, nslab, n), dim3(256), 0, st, cx, rows, cols, slab, pmax, pidx);
  hipLaunchKernelGGL(cx_colmax_final_kernel, dim3(gx, n), dim3(256), 0, st, pmax, pidx, sim, rows, cols, nslab, 1.f / band_width, rmax, cw);
  hipLaunchKernelGGL(cx_image_kernel, dim3(n), dim3(256), 0, st, rmax, cw, cols, cx_image);
  hipLaunchKernelGGL(cx_loss_kernel, dim3(1), dim3(64), 0, st, cx_image, n, loss);
  return check_launch("cx_reduce");
}

extern "C" int gpemsr_normalize3(const float* x, int64_t pixels, int ld, const float* mean3, const float* std3, float* out, int out_ld,
                                 void* stream) {
  GP_REQUIRE(x && out && mean3 && std3 && pixels > 0 && ld >= 3 && out_ld >= 3, "normalize3: bad args");
  // mean3 / std3 are HOST pointers (the registered buffers of ContextualLoss)
  const long long blocks = (pixels + 255) / 256;
  hipLaunchKernelGGL(normalize3_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, (long long)pixels, ld, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], out, out_ld);
  return check_launch("normalize3");
}
