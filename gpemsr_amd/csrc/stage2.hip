// Stage-2 (indexer) training step, R:train_stage2.py:351-366: the gradient kernels the VQGAN-style blocks add to the
// stage-3 set (backward.hip / wgrad.hip): GroupNorm(32, eps) + ReLU backward, row-softmax backward (attention) and
// CrossEntropyLoss (mean) forward + backward.  All HBM-bound, two-stage fixed-order reductions, no float atomics.
#include "common.h"

namespace gpemsr {

__device__ __forceinline__ float s2_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float s2_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// GroupNorm backward, stage 1: per (image, part) per-channel S1 = sum dN, S2 = sum dN * xhat over the part's pixels, with
// dN = dy masked by the ReLU (y = gamma*xhat + beta > 0) when relu != 0.  grid (parts, n); thread <-> channel.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* x, int ld, const float* dy, int dy_ld, int hw, int c, int groups,
                                                             const float* mr, const float* gamma, const float* beta, int relu, int parts,
                                                             float* ws) {
  const int part = blockIdx.x, img = blockIdx.y, cpg = c / groups;
  const int per = (hw + parts - 1) / parts;
  const int p0 = part * per, p1 = min(hw, p0 + per);
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    const int g = ch / cpg;
    const float mean = mr[2 * (img * groups + g)], rstd = mr[2 * (img * groups + g) + 1];
    const float ga = gamma[ch], be = beta[ch];
    float s1 = 0.f, s2 = 0.f;
    for (int p = p0; p < p1; ++p) {
      const long long pix = (long long)img * hw + p;
      const float xh = (x[pix * ld + ch] - mean) * rstd;
      float d = dy[pix * dy_ld + ch];
      if (relu && !(ga * xh + be > 0.f)) d = 0.f;
      s1 += d; s2 += d * xh;
    }
    float* o = ws + (((long long)img * parts + part) * c + ch) * 2;
    o[0] = s1; o[1] = s2;
  }
}
// stage 2: one block per image: S[n][c][2] = sum over parts; then per group A = sum_c gamma*S1, B = sum_c gamma*S2
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(const float* ws, int c, int groups, int parts, const float* gamma, float* S, float* AB) {
  const int img = blockIdx.x, cpg = c / groups;
  __shared__ float sa[1024], sb[1024];
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < parts; ++p) {
      const float* o = ws + (((long long)img * parts + p) * c + ch) * 2;
      s1 += o[0]; s2 += o[1];
    }
    S[((long long)img * c + ch) * 2] = s1; S[((long long)img * c + ch) * 2 + 1] = s2;
    sa[ch] = gamma[ch] * s1; sb[ch] = gamma[ch] * s2;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < groups; g += 256) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < cpg; ++k) { a += sa[g * cpg + k]; b += sb[g * cpg + k]; }
    AB[((long long)img * groups + g) * 2] = a; AB[((long long)img * groups + g) * 2 + 1] = b;
  }
}
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(const float* S, int n, int c, float* dgamma, float* dbeta) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  float dg = 0.f, db = 0.f;
  for (int i = 0; i < n; ++i) { db += S[((long long)i * c + ch) * 2]; dg += S[((long long)i * c + ch) * 2 + 1]; }
  dgamma[ch] += dg; dbeta[ch] += db;
}
// dx += rstd * (gamma*dN - (A + xhat*B) / m),  m = cpg * hw
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* x, int ld, const float* dy, int dy_ld, long long total, int hw, int c,
                                                           int groups, const float* mr, const float* gamma, const float* beta, int relu,
                                                           const float* AB, float* dx, int dx_ld) {
  const int cpg = c / groups;
  const float inv_m = 1.f / ((float)cpg * (float)hw);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    const long long pix = e / c;
    const int img = (int)(pix / hw), g = ch / cpg;
    const float mean = mr[2 * (img * groups + g)], rstd = mr[2 * (img * groups + g) + 1];
    const float ga = gamma[ch];
    const float xh = (x[pix * ld + ch] - mean) * rstd;
    float d = dy[pix * dy_ld + ch];
    if (relu && !(ga * xh + beta[ch] > 0.f)) d = 0.f;
    const float A = AB[((long long)img * groups + g) * 2], B = AB[((long long)img * groups + g) * 2 + 1];
    dx[pix * dx_ld + ch] += rstd * (ga * d - (A + xh * B) * inv_m);
  }
}

// softmax backward on rows: ds = p * (dp - sum_j dp*p)   (written over dp)
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* p, float* dp, int cols) {
  const float* pr = p + (long long)blockIdx.x * cols;
  float* dr = dp + (long long)blockIdx.x * cols;
  const int c4 = cols >> 2;
  float4 pv[MAXV], dv[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      pv[i] = *reinterpret_cast<const float4*>(pr + 4 * e);
      dv[i] = *reinterpret_cast<const float4*>(dr + 4 * e);
      s += (pv[i].x * dv[i].x + pv[i].y * dv[i].y) + (pv[i].z * dv[i].z + pv[i].w * dv[i].w);
    }
  }
  __shared__ float red[4];
  s = s2_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4)
      *reinterpret_cast<float4*>(dr + 4 * e) = make_float4(pv[i].x * (dv[i].x - s), pv[i].y * (dv[i].y - s), pv[i].z * (dv[i].z - s),
                                                           pv[i].w * (dv[i].w - s));
  }
}

// CrossEntropyLoss (mean over rows): one wave per row.  row_loss[r] = logsumexp(x_r) - x_r[t_r];
// dlogits[r][j] = scale * (softmax(x_r)[j] - [j == t_r]) / rows   (dlogits may be NULL)
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* logits, const int* target, long long rows, int cols, float gscale,
                                                            float* row_loss, float* dlogits) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* x = logits + r * cols;
  float m = -INFINITY;
  for (int j = lane; j < cols; j += 64) m = fmaxf(m, x[j]);
  m = s2_wave_max(m);
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) s += expf(x[j] - m);
  s = s2_wave_sum(s);
  const int t = target[r];
  if (lane == 0) row_loss[r] = (m + logf(s)) - x[t];
  if (dlogits) {
    float* d = dlogits + r * cols;
    for (int j = lane; j < cols; j += 64) d[j] = gscale * (expf(x[j] - m) / s - (j == t ? 1.f : 0.f));
  }
}
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* v, long long rows, float* out) {
  // single block, fixed order: thread t sums rows t, t+256, ... in double, then an LDS tree
  __shared__ double red[256];
  double s = 0.0;
  for (long long r = threadIdx.x; r < rows; r += 256) s += (double)v[r];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)rows);
}

}  // namespace gpemsr

using namespace gpemsr;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int gpemsr_groupnorm_bwd(const float* x, int ld, const float* dy, int dy_ld, int n, int hw, int c, int groups,
                                    const float* mean_rstd, const float* gamma, const float* beta, int relu, float* ws,
                                    int64_t ws_floats, float* dx, int dx_ld, float* dgamma, float* dbeta, void* stream) {
  GP_REQUIRE(x && dy && mean_rstd && gamma && beta && ws && dx && n > 0 && hw > 0 && c > 0 && c <= 1024 && groups > 0 && c % groups == 0,
             "groupnorm_bwd: bad args (c <= 1024)");
  int parts = hw / 64; if (parts < 1) parts = 1; if (parts > 64) parts = 64;
  const long long need = (long long)n * parts * c * 2 + (long long)n * c * 2 + (long long)n * groups * 2;
  GP_REQUIRE(need <= ws_floats, "groupnorm_bwd: workspace too small (need %lld floats)", need);
  float* S = ws + (long long)n * parts * c * 2;
  float* AB = S + (long long)n * c * 2;
  hipStream_t st = ST(stream);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(parts, n), dim3(256), 0, st, x, ld, dy, dy_ld, hw, c, groups, mean_rstd, gamma, beta, relu, parts, ws);
  hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(n), dim3(256), 0, st, ws, c, groups, parts, gamma, S, AB);
  if (dgamma && dbeta) hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((c + 255) / 256), dim3(256), 0, st, S, n, c, dgamma, dbeta);
  const long long total = (long long)n * hw * c;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, st, x, ld, dy, dy_ld, total, hw, c,
                     groups, mean_rstd, gamma, beta, relu, AB, dx, dx_ld);
  return check_launch("groupnorm_bwd");
}

extern "C" int gpemsr_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int cols, void* stream) {
  GP_REQUIRE(p && dp && rows > 0 && rows < (1ll << 31) && cols % 4 == 0 && cols <= 256 * 4 * 16, "softmax_bwd_rows: cols=%d unsupported", cols);
  if (cols <= 256 * 4 * 4) hipLaunchKernelGGL(softmax_bwd_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, ST(stream), p, dp, cols);
  else hipLaunchKernelGGL(softmax_bwd_rows_kernel<16>, dim3((unsigned)rows), dim3(256), 0, ST(stream), p, dp, cols);
  return check_launch("softmax_bwd_rows");
}

extern "C" int gpemsr_cross_entropy(const float* logits, const int32_t* target, int64_t rows, int cols, float grad_scale, float* row_loss,
                                    float* loss, float* dlogits, void* stream) {
  GP_REQUIRE(logits && target && row_loss && loss && rows > 0 && cols > 0, "cross_entropy: bad args");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), logits, target, (long long)rows, cols,
                     grad_scale / (float)rows, row_loss, dlogits);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, ST(stream), row_loss, (long long)rows, loss);
  return check_launch("cross_entropy");
}
