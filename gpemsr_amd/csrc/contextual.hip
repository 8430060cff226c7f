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


// ---------------------------------------------------------------------------------------------------------------
// Backward of contextual_loss w.r.t. the similarity matrix S (the x side only: y = VGG(ref_img) is a constant of
// the training step).  With g_n = dL/dCX[n] = -scale / (N (CX[n] + 1e-5)),  B_n = sum_j c_j:
//   d/d r_j = g c_j / B            -> lands on cx[i*_j, j]           (torch.max backward: the arg-max element)
//   d/d c_j = g (r_j - CX) / B     -> d dist[i*_j, j] += (d/d c_j) c_j (-1/h)
//   row i, J_i = { j : i*_j = i }:  q_i = sum_{j in J_i} (d/d r_j) cx_ij
//     dw_ij = ([j in J_i] d/d r_j - q_i) / (sum_j w + 1e-5);   dt_ij = dw_ij w_ij (-1/h)
//     d dist_ij += dt_ij / (dmin_i + 1e-5);   d dmin_i = -sum_j dt_ij dist_ij / (dmin_i + 1e-5)^2 -> first arg-min j
//   dS_ij = -(d dist_ij) where 1 - S_ij >= 0 (clamp(min=0) passes the gradient at equality), else 0.
// Rows with empty J_i get an all-zero gradient.
__global__ __launch_bounds__(256) void cx_bwd_cols_kernel(const float* pmax, const int* pidx, int nslab, const float* rmax, const float* cw,
                                                          const float* cxn, int n_img, int cols, float inv_h, float scale, int* idx, float* gr,
                                                          float* gd) {
  // per image: B = sum_j c_j (fixed order); per column: merge the row-slab arg-maxes of cx_colmax_partial_kernel (first
  // maximum wins, slabs in order) and derive the two coefficients
  const int n = blockIdx.y;
  const float* c = cw + (long long)n * cols;
  __shared__ float red[4];
  float b = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) b += c[j];
  b = cx_wave_sum(b);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  const float B = (red[0] + red[1]) + (red[2] + red[3]);
  const float cxi = cxn[n];
  const float g = -scale / ((float)n_img * (cxi + 1e-5f));
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  float best = -INFINITY; int bi = 0;
  for (int sb = 0; sb < nslab; ++sb) {
    const long long o = ((long long)n * nslab + sb) * cols + j;
    if (pmax[o] > best) { best = pmax[o]; bi = pidx[o]; }
  }
  const long long o = (long long)n * cols + j;
  idx[o] = bi;
  gr[o] = g * c[j] / B;
  gd[o] = g * (rmax[o] - cxi) / B * c[j] * (-inv_h);
}

template <int MAXV>
__global__ __launch_bounds__(256) void cx_bwd_rows_kernel(const float* sim, const int* idx, const float* gr, const float* gd, int rows,
                                                          int cols, float inv_h, float* dsim) {
  const long long rowid = blockIdx.x;
  const int n = (int)(rowid / rows), i = (int)(rowid % rows);
  const float* row = sim + rowid * cols;
  float* orow = dsim + rowid * cols;
  const int* ix = idx + (long long)n * cols;
  const float* grn = gr + (long long)n * cols;
  const float* gdn = gd + (long long)n * cols;
  const int c4 = cols >> 2;
  __shared__ float red[4];
  __shared__ int redi[4];
  // does any column pick this row?
  int cnt = 0;
  for (int j = threadIdx.x; j < cols; j += 256) cnt += (ix[j] == i);
  cnt = __syncthreads_count(cnt);
  if (cnt == 0) {
    for (int e = threadIdx.x; e < c4; e += 256) *reinterpret_cast<float4*>(orow + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float d[MAXV][4], wv[MAXV][4];
  float m = INFINITY;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4) {
      const float4 s = *reinterpret_cast<const float4*>(row + 4 * e);
      d[k][0] = fmaxf(1.f - s.x, 0.f); d[k][1] = fmaxf(1.f - s.y, 0.f); d[k][2] = fmaxf(1.f - s.z, 0.f); d[k][3] = fmaxf(1.f - s.w, 0.f);
      m = fminf(fminf(fminf(m, d[k][0]), fminf(d[k][1], d[k][2])), d[k][3]);
    }
  }
  m = cx_wave_min(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  __syncthreads();
  // first arg-min column
  int jm = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4)
#pragma unroll
      for (int q = 3; q >= 0; --q) if (d[k][q] == m) jm = min(jm, 4 * e + q);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) jm = min(jm, __shfl_xor(jm, o, 64));
  if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = jm;
  __syncthreads();
  jm = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
  const float den = m + 1e-5f;
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[k][q] = expf((1.f - d[k][q] / den) * inv_h);
      sum += (wv[k][0] + wv[k][1]) + (wv[k][2] + wv[k][3]);
    }
  }
  sum = cx_wave_sum(sum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float sw = (red[0] + red[1]) + (red[2] + red[3]) + 1e-5f;
  // q_i = sum_{j in J_i} gr_j * cx_ij
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (ix[4 * e + r] == i) q += grn[4 * e + r] * (wv[k][r] / sw);
  }
  q = cx_wave_sum(q);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  q = (red[0] + red[1]) + (red[2] + red[3]);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool inj = ix[4 * e + r] == i;
        const float dw = ((inj ? grn[4 * e + r] : 0.f) - q) / sw;
        const float dt = dw * wv[k][r] * (-inv_h);
        acc += dt * d[k][r];
        wv[k][r] = dt / den + (inj ? gdn[4 * e + r] : 0.f);        // d dist_ij (without the dmin term)
      }
  }
  acc = cx_wave_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float ddmin = -((red[0] + red[1]) + (red[2] + red[3])) / (den * den);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < c4) {
      const float4 s = *reinterpret_cast<const float4*>(row + 4 * e);
      const float sv[4] = {s.x, s.y, s.z, s.w};
      float o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float dd = wv[k][r] + ((4 * e + r) == jm ? ddmin : 0.f);
        o4[r] = (1.f - sv[r] >= 0.f) ? -dd : 0.f;
      }
      *reinterpret_cast<float4*>(orow + 4 * e) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
  }
}

// F.normalize(x - mu) backward: dx += (g - x^ <g, x^>) / max(||x - mu||, 1e-12); one wave per pixel
__global__ __launch_bounds__(256) void cx_center_normalize_bwd_kernel(const float* x, const float* mu, const float* g, long long pixels, int c,
                                                                      int ld, int g_ld, float* dx, int dx_ld) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= pixels) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + p * ld;
  const float* gr = g + p * g_ld;
  float ss = 0.f, dot = 0.f;
  for (int ch = lane; ch < c; ch += 64) { const float v = xr[ch] - mu[ch]; ss += v * v; dot += v * gr[ch]; }
  ss = cx_wave_sum(ss); dot = cx_wave_sum(dot);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  const float k = dot / (nrm * nrm);                 // <g, x^> / nrm, with x^ = v / nrm
  float* drow = dx + p * dx_ld;
  for (int ch = lane; ch < c; ch += 64) drow[ch] += (gr[ch] - (xr[ch] - mu[ch]) * k) / nrm;
}

// a 1-channel image seen by VGG as three identical channels, normalised: out[p][c] = (x[p] - mean_c) / std_c
// (train_stage3.py:356-358 expand(-1,-1,3,..) + model/contextual.py:222-224), and its backward dx[p] += sum_c g[p][c] / std_c
__global__ __launch_bounds__(256) void gray_normalize3_kernel(const float* x, long long pixels, float m0, float m1, float m2, float s0, float s1,
                                                              float s2, float* out) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const float v = x[p];
    out[3 * p] = (v - m0) / s0; out[3 * p + 1] = (v - m1) / s1; out[3 * p + 2] = (v - m2) / s2;
  }
}
__global__ __launch_bounds__(256) void gray_normalize3_bwd_kernel(const float* g, long long pixels, float s0, float s1, float s2, float* dx) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256)
    dx[p] += (g[3 * p] / s0 + g[3 * p + 1] / s1) + g[3 * p + 2] / s2;
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

extern "C" int gpemsr_cx_backward(const float* sim, const float* cx, const float* rmax, const float* cw, const float* cx_image, int n,
                                  int rows, int cols, float band_width, float scale, int32_t* idx_ws, float* coef_ws, float* dsim,
                                  void* stream) {
  GP_REQUIRE(sim && cx && rmax && cw && cx_image && idx_ws && coef_ws && dsim && n > 0 && rows > 0 && band_width > 0.f, "cx_backward: bad args");
  GP_REQUIRE(cols % 4 == 0 && cols <= 256 * 4 * 16 && n <= 65535, "cx_backward: cols=%d unsupported (multiple of 4, <= 16384)", cols);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // coef_ws: [gr | gd | slab maxima | slab arg-maxes], 2*n*cols + 2*n*nslab*cols floats with nslab = ceil(rows/128)
  const int slab = 128;
  const int nslab = (rows + slab - 1) / slab;
  GP_REQUIRE(nslab <= 65535, "cx_backward: grid too large");
  float* gr = coef_ws;
  float* gd = coef_ws + (long long)n * cols;
  float* pmax = gd + (long long)n * cols;
  int* pidx = reinterpret_cast<int*>(pmax + (long long)n * nslab * cols);
  const unsigned gx = (unsigned)((cols + 255) / 256);
  hipLaunchKernelGGL(cx_colmax_partial_kernel, dim3(gx, nslab, n), dim3(256), 0, st, cx, rows, cols, slab, pmax, pidx);
  hipLaunchKernelGGL(cx_bwd_cols_kernel, dim3(gx, n), dim3(256), 0, st, pmax, pidx, nslab, rmax, cw, cx_image, n, cols, 1.f / band_width, scale,
                     idx_ws, gr, gd);
  const unsigned nrows = (unsigned)((long long)n * rows);
  if (cols <= 256 * 4 * 4) hipLaunchKernelGGL(cx_bwd_rows_kernel<4>, dim3(nrows), dim3(256), 0, st, sim, idx_ws, gr, gd, rows, cols, 1.f / band_width, dsim);
  else hipLaunchKernelGGL(cx_bwd_rows_kernel<16>, dim3(nrows), dim3(256), 0, st, sim, idx_ws, gr, gd, rows, cols, 1.f / band_width, dsim);
  return check_launch("cx_backward");
}

extern "C" int gpemsr_cx_center_normalize_bwd(const float* x, const float* mean, const float* g, int64_t pixels, int c, int ld, int g_ld,
                                              float* dx, int dx_ld, void* stream) {
  GP_REQUIRE(x && mean && g && dx && pixels > 0 && c > 0, "cx_center_normalize_bwd: bad args");
  hipLaunchKernelGGL(cx_center_normalize_bwd_kernel, dim3((unsigned)((pixels + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, mean, g, (long long)pixels, c, ld, g_ld, dx, dx_ld);
  return check_launch("cx_center_normalize_bwd");
}

extern "C" int gpemsr_gray_normalize3(const float* x, int64_t pixels, const float* mean3, const float* std3, float* out, void* stream) {
  GP_REQUIRE(x && out && mean3 && std3 && pixels > 0, "gray_normalize3: bad args");
  const long long blocks = (pixels + 255) / 256;
  hipLaunchKernelGGL(gray_normalize3_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, (long long)pixels, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  return check_launch("gray_normalize3");
}

extern "C" int gpemsr_gray_normalize3_bwd(const float* g, int64_t pixels, const float* std3, float* dx, void* stream) {
  GP_REQUIRE(g && dx && std3 && pixels > 0, "gray_normalize3_bwd: bad args");
  const long long blocks = (pixels + 255) / 256;
  hipLaunchKernelGGL(gray_normalize3_bwd_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     g, (long long)pixels, std3[0], std3[1], std3[2], dx);
  return check_launch("gray_normalize3_bwd");
}
