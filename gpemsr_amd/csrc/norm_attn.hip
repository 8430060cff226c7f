// GroupNorm statistics/apply, row softmax, row argmax, row gather (VQGAN prior).
// All HBM-bound: float4 coalesced traffic, wave shuffles + LDS for reductions,
// deterministic two-stage reductions (no float atomics) so results are
// bit-stable run to run.  Replaces model/blocks.py:5-6,13-28,61-83 pieces and
// model/codebook.py:34-43.
#include "common.h"

namespace gpemsr {

// ---- GroupNorm stats, stage 1: per (image, part) per-channel sum / sumsq ----
// grid (parts, n); thread t owns float4 column (t % c4) and pixel rows t / c4 + k*(256/c4)
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* x, int hw, int c, int ld, int parts, float* ws) {
  const int c4 = c >> 2;
  const int col = threadIdx.x % c4, row = threadIdx.x / c4, rows = 256 / c4;
  const int part = blockIdx.x, img = blockIdx.y;
  const int per = (hw + parts - 1) / parts;
  const int p0 = part * per, p1 = min(hw, p0 + per);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  const float* xp = x + (long long)img * hw * ld + 4 * col;
  for (int p = p0 + row; p < p1; p += rows) {
    const float4 v = *reinterpret_cast<const float4*>(xp + (long long)p * ld);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
  }
  __shared__ float4 ss[256], sq[256];
  ss[threadIdx.x] = s; sq[threadIdx.x] = q;
  __syncthreads();
  if (row == 0) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int r = 0; r < rows; ++r) {       // fixed order -> deterministic
      const float4 u = ss[r * c4 + col], w = sq[r * c4 + col];
      a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
      b[0] += w.x; b[1] += w.y; b[2] += w.z; b[3] += w.w;
    }
    float* o = ws + (((long long)img * parts + part) * c + 4 * col) * 2;
    for (int k = 0; k < 4; ++k) { o[2 * k] = (float)a[k]; o[2 * k + 1] = (float)b[k]; }
  }
}

// stage 2: one wave per (image, group): lanes stride over the parts x channels-per-group partials, fixed-order
// double-precision shuffle reduction (deterministic); mean, rstd (biased variance, eps inside the sqrt)
__global__ __launch_bounds__(256) void gn_final_kernel(const float* ws, int n, int hw, int c, int groups, int parts, float eps, float* mr) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n * groups) return;
  const int lane = threadIdx.x & 63;
  const int img = i / groups, g = i % groups, cpg = c / groups;
  double s = 0, q = 0;
  for (int e = lane; e < parts * cpg; e += 64) {
    const int p = e / cpg, k = e % cpg;
    const float* o = ws + (((long long)img * parts + p) * c + g * cpg + k) * 2;
    s += o[0]; q += o[1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m); q += __shfl_xor(q, m); }
  if (lane == 0) {
    const double cnt = (double)hw * cpg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0) var = 0;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* x, long long total4, int hw, int c, int ld, int groups,
                                                       const float* mr, const float* gamma, const float* beta, int relu,
                                                       const float* residual, int res_ld, float* out, int out_ld) {
  const int c4 = c >> 2, cpg = c / groups;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
    const int col = (int)(e % c4);
    const long long pix = e / c4;
    const int img = (int)(pix / hw);
    const int ch = 4 * col;
    const float4 v = *reinterpret_cast<const float4*>(x + pix * ld + ch);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + ch);
    const float4 b4 = *reinterpret_cast<const float4*>(beta + ch);
    float r[4] = {v.x, v.y, v.z, v.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int grp = (ch + k) / cpg;
      const float mean = mr[2 * (img * groups + grp)], rstd = mr[2 * (img * groups + grp) + 1];
      float y = (r[k] - mean) * rstd * gg[k] + bb[k];
      if (relu) y = y > 0.f ? y : 0.f;
      r[k] = y;
    }
    if (residual) {
      const float4 rr = *reinterpret_cast<const float4*>(residual + pix * res_ld + ch);
      r[0] += rr.x; r[1] += rr.y; r[2] += rr.z; r[3] += rr.w;
    }
    *reinterpret_cast<float4*>(out + pix * out_ld + ch) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ---- block reductions ----
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// one workgroup per row; cols <= 256*MAXV*4
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* x, int cols, int ld) {
  float* row = x + (long long)blockIdx.x * ld;
  const int c4 = cols >> 2;
  float4 v[MAXV];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      v[i] = *reinterpret_cast<const float4*>(row + 4 * e);
      m = fmaxf(fmaxf(fmaxf(m, v[i].x), fmaxf(v[i].y, v[i].z)), v[i].w);
    }
  }
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c4) {
      v[i].x *= inv; v[i].y *= inv; v[i].z *= inv; v[i].w *= inv;
      *reinterpret_cast<float4*>(row + 4 * e) = v[i];
    }
  }
}

// one wave per row; ties -> lowest index (torch CPU topk/argmax behaviour)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, long long rows, int cols, int32_t* idx) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* row = x + r * cols;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < cols; c += 64) {
    const float v = row[c];
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m); const int oi = __shfl_xor(bi, m);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) idx[r] = bi;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* table, int dim, const int32_t* idx, long long rows,
                                                          float* out, int out_ld) {
  const int d4 = dim >> 2;
  const long long total = rows * d4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / d4; const int j = (int)(e % d4);
    *reinterpret_cast<float4*>(out + r * out_ld + 4 * j) =
        *reinterpret_cast<const float4*>(table + (long long)idx[r] * dim + 4 * j);
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_groupnorm_stats(const float* x, int n, int hw, int c, int ld, int groups, float eps,
                                      float* ws, int parts, float* mean_rstd, void* stream) {
  GP_REQUIRE(x && ws && mean_rstd, "groupnorm_stats: null pointer");
  GP_REQUIRE(c % 4 == 0 && (c / 4) <= 256 && 256 % (c / 4) == 0, "groupnorm_stats: c=%d unsupported", c);
  GP_REQUIRE(c % groups == 0 && ld % 4 == 0 && parts >= 1, "groupnorm_stats: bad groups/ld/parts");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(parts, n), dim3(256), 0, st, x, hw, c, ld, parts, ws);
  hipLaunchKernelGGL(gn_final_kernel, dim3(cdiv(n * groups, 4)), dim3(256), 0, st, ws, n, hw, c, groups, parts, eps, mean_rstd);
  return check_launch("groupnorm_stats");
}

// scale = rstd * gamma, shift = beta - mean * rstd * gamma per (image, channel): GroupNorm's apply as a per-channel affine map (the
// form gn_apply16_cols_kernel uses), for convolutions that apply it to their source while staging it (gpemsr_conv16_desc.a_scale)
__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const float* mr, const float* gamma, const float* beta, int n, int c, int groups,
                                                             float* scale, float* shift) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n * c) return;
  const int img = e / c, ch = e % c, grp = ch / (c / groups);
  const float mean = mr[2 * (img * groups + grp)], rstd = mr[2 * (img * groups + grp) + 1];
  const float g = gamma[ch], b = beta[ch];
  scale[e] = rstd * g;
  shift[e] = b - mean * rstd * g;
}

extern "C" int gpemsr_groupnorm_scale_shift(const float* mean_rstd, const float* gamma, const float* beta, int n, int c, int groups,
                                            float* scale, float* shift, void* stream) {
  GP_REQUIRE(mean_rstd && gamma && beta && scale && shift && n > 0 && c > 0 && groups > 0 && c % groups == 0, "groupnorm_scale_shift: bad args");
  hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(cdiv((long long)n * c, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mean_rstd, gamma, beta,
                     n, c, groups, scale, shift);
  return check_launch("groupnorm_scale_shift");
}

extern "C" int gpemsr_groupnorm_finish(const float* ws, int n, int hw, int c, int groups, int parts, float eps, float* mean_rstd, void* stream) {
  GP_REQUIRE(ws && mean_rstd && n > 0 && hw > 0 && c % groups == 0 && parts >= 1, "groupnorm_finish: bad args");
  hipLaunchKernelGGL(gn_final_kernel, dim3(cdiv(n * groups, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ws, n, hw, c, groups, parts, eps, mean_rstd);
  return check_launch("groupnorm_finish");
}

extern "C" int gpemsr_groupnorm_apply(const float* x, int n, int hw, int c, int ld, int groups, const float* mean_rstd,
                                      const float* gamma, const float* beta, int relu,
                                      const float* residual, int res_ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && mean_rstd && gamma && beta && out, "groupnorm_apply: null pointer");
  GP_REQUIRE(c % 4 == 0 && ld % 4 == 0 && out_ld % 4 == 0 && (!residual || res_ld % 4 == 0), "groupnorm_apply: alignment");
  const long long total4 = (long long)n * hw * (c / 4);
  const long long blocks = (total4 + 255) / 256;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, total4, hw, c, ld, groups, mean_rstd, gamma, beta, relu,
                     residual, res_ld, out, out_ld);
  return check_launch("groupnorm_apply");
}

extern "C" int gpemsr_softmax_rows(float* x, int64_t rows, int cols, void* stream) {
  GP_REQUIRE(x && rows > 0 && rows < (1ll << 31), "softmax_rows: bad rows");
  GP_REQUIRE(cols % 4 == 0 && cols <= 256 * 4 * 16, "softmax_rows: cols=%d unsupported", cols);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (cols <= 256 * 4 * 4) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, st, x, cols, cols);
  else hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3((unsigned)rows), dim3(256), 0, st, x, cols, cols);
  return check_launch("softmax_rows");
}

extern "C" int gpemsr_softmax_rows_ld(float* x, int64_t rows, int cols, int ld, void* stream) {
  GP_REQUIRE(x && rows > 0 && rows < (1ll << 31) && ld >= cols && ld % 4 == 0, "softmax_rows_ld: bad rows / ld");
  GP_REQUIRE(cols % 4 == 0 && cols <= 256 * 4 * 16, "softmax_rows_ld: cols=%d unsupported", cols);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (cols <= 256 * 4 * 4) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, st, x, cols, ld);
  else hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3((unsigned)rows), dim3(256), 0, st, x, cols, ld);
  return check_launch("softmax_rows_ld");
}

extern "C" int gpemsr_argmax_rows(const float* x, int64_t rows, int cols, int32_t* idx, void* stream) {
  GP_REQUIRE(x && idx && rows > 0 && cols > 0, "argmax_rows: bad args");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, (long long)rows, cols, idx);
  return check_launch("argmax_rows");
}

extern "C" int gpemsr_gather_rows(const float* table, int dim, const int32_t* idx, int64_t rows, float* out, int out_ld,
                                  void* stream) {
  GP_REQUIRE(table && idx && out && dim % 4 == 0 && out_ld % 4 == 0, "gather_rows: bad args");
  const long long total = rows * (dim / 4);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), table, dim, idx, (long long)rows, out, out_ld);
  return check_launch("gather_rows");
}
