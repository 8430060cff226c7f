// HBM-bound kernels of the bf16 data path (precision = "bf16"): the same operators as norm_attn.hip / resample.hip /
// fused.hip / conv_direct.hip, on bf16 NHWC tensors.  Arithmetic is fp32 in registers; only the HBM format changes
// (2 B per element each way).  1-channel images (LR slices, prior image, masks), flows, deformable offsets and logits
// stay fp32, so several kernels take one fp32 and one bf16 operand.  Every kernel moves 8 or 16 bytes per lane along C.
// Replaces the same reference calls as its fp32 counterpart (cited per entry point in include/gpemsr_hip.h).
#include "common.h"

namespace gpemsr {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bfl(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfh(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ bf16_t to_bf16(float a) { return (bf16_t)(pk_bf16(a, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float from_bf16(bf16_t a) { return __uint_as_float((unsigned)a << 16); }

// 8 consecutive bf16 (16 B) <-> 8 floats
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = bfl(u.x); v[1] = bfh(u.x); v[2] = bfl(u.y); v[3] = bfh(u.y); v[4] = bfl(u.z); v[5] = bfh(u.z); v[6] = bfl(u.w); v[7] = bfh(u.w);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
}
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

inline unsigned grid16(long long total) {
  const long long b = (total + 255) / 256;
  return (unsigned)(b < 32768 ? (b < 1 ? 1 : b) : 32768);
}

// ---- casts (module boundary, tests) ----
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* x, long long pixels, int c, int x_ld, bf16_t* out, int out_ld) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    out[p * out_ld + ch] = to_bf16(x[p * x_ld + ch]);
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* x, long long pixels, int c, int x_ld, float* out, int out_ld) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    out[p * out_ld + ch] = from_bf16(x[p * x_ld + ch]);
  }
}

// x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|: the A operands of a "three bf16 products" GEMM
// (hi*Whi + lo*Whi + hi*Wlo) that keeps an fp32 tensor's precision on the bf16 matrix pipe (the indexer's logits, engine.py).
__global__ __launch_bounds__(256) void split_f32_bf16x2_kernel(const float* x, long long pixels, int c4, int x_ld, bf16_t* hi, int hi_ld, bf16_t* lo, int lo_ld) {
  const long long total = pixels * c4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c4; const int ch = 4 * (int)(e % c4);
    const float4 v = *reinterpret_cast<const float4*>(x + p * x_ld + ch);
    const float f[4] = {v.x, v.y, v.z, v.w};
    bf16_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { h[k] = to_bf16(f[k]); l[k] = to_bf16(f[k] - from_bf16(h[k])); }
    *reinterpret_cast<uint2*>(hi + p * hi_ld + ch) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(lo + p * lo_ld + ch) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

// ---- GroupNorm: statistics pass for tensors no convolution just produced, and the apply pass ----
// grid (parts, n); thread t owns 8 channels (t % c8) of pixel rows t / c8 + k*(256/c8)
__global__ __launch_bounds__(256) void gn_partial16_kernel(const bf16_t* x, int hw, int c, int ld, int parts, float* ws) {
  const int c8 = c >> 3;
  const int col = threadIdx.x % c8, row = threadIdx.x / c8, rows = 256 / c8;
  const int part = blockIdx.x, img = blockIdx.y;
  const int per = (hw + parts - 1) / parts;
  const int p0 = part * per, p1 = min(hw, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
  const bf16_t* xp = x + (long long)img * hw * ld + 8 * col;
  if (row < rows)
    for (int p = p0 + row; p < p1; p += rows) {
      float v[8];
      ld8(xp + (long long)p * ld, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s[k] += v[k]; q[k] = fmaf(v[k], v[k], q[k]); }
    }
  __shared__ float ss[256 * 8], sq[256 * 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { ss[threadIdx.x * 8 + k] = s[k]; sq[threadIdx.x * 8 + k] = q[k]; }
  __syncthreads();
  if (row == 0 && col < c8) {
    for (int k = 0; k < 8; ++k) {
      double a = 0, b = 0;
      for (int r = 0; r < rows; ++r) { a += ss[(r * c8 + col) * 8 + k]; b += sq[(r * c8 + col) * 8 + k]; }   // fixed order
      float* o = ws + (((long long)img * parts + part) * c + 8 * col + k) * 2;
      o[0] = (float)a; o[1] = (float)b;
    }
  }
}

// y = relu?((x - mean) * rstd * gamma + beta) (+ residual); bf16 in / out; 8 channels per thread (one group per 8 channels or wider)
__global__ __launch_bounds__(256) void gn_apply16_kernel(const bf16_t* x, long long total8, int hw, int c, int ld, int groups,
                                                         const float* mr, const float* gamma, const float* beta, int relu,
                                                         const bf16_t* residual, int res_ld, bf16_t* out, int out_ld) {
  const int c8 = c >> 3, cpg = c / groups;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long long)gridDim.x * 256) {
    const int col = (int)(e % c8);
    const long long pix = e / c8;
    const int img = (int)(pix / hw);
    const int ch = 8 * col;
    float v[8], g[8], b[8];
    ld8(x + pix * ld + ch, v);
    ld8f(gamma + ch, g); ld8f(beta + ch, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int grp = (ch + k) / cpg;
      const float mean = mr[2 * (img * groups + grp)], rstd = mr[2 * (img * groups + grp) + 1];
      float y = (v[k] - mean) * rstd * g[k] + b[k];
      if (relu) y = y > 0.f ? y : 0.f;
      v[k] = y;
    }
    if (residual) {
      float r[8];
      ld8(residual + pix * res_ld + ch, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
    st8(out + pix * out_ld + ch, v);
  }
}

// The same for channel counts whose 16-byte columns divide the workgroup (c/8 in {1,2,4,...,256}): grid (pixel blocks, n); a
// thread keeps ONE 8-channel column for its whole life, so scale = rstd*gamma and shift = beta - mean*rstd*gamma are formed once
// and a pixel costs one 16-byte load, 8 FMAs and one 16-byte store (the generic kernel above spends ~20 dependent scalar loads and
// two 64-bit divisions per 16 bytes: 2.5 TB/s on the VQGAN blocks).  UNR independent pixels are in flight per thread.
template <int UNR>
__global__ __launch_bounds__(256) void gn_apply16_cols_kernel(const bf16_t* x, int hw, int c, int ld, int groups, const float* mr, const float* gamma,
                                                              const float* beta, int relu, const bf16_t* residual, int res_ld, bf16_t* out, int out_ld,
                                                              int pix_per_block) {
  const int c8 = c >> 3, cpg = c / groups;
  const int col = threadIdx.x % c8, row = threadIdx.x / c8, rows = 256 / c8;
  const int img = blockIdx.y, ch = 8 * col;
  float sc[8], sh[8];
  {
    float g[8], b[8];
    ld8f(gamma + ch, g); ld8f(beta + ch, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int grp = (ch + k) / cpg;
      const float mean = mr[2 * (img * groups + grp)], rstd = mr[2 * (img * groups + grp) + 1];
      sc[k] = rstd * g[k]; sh[k] = b[k] - mean * rstd * g[k];
    }
  }
  const int p0 = blockIdx.x * pix_per_block, p1 = min(hw, p0 + pix_per_block);
  const bf16_t* xp = x + (long long)img * hw * ld + ch;
  const bf16_t* rp = residual ? residual + (long long)img * hw * res_ld + ch : nullptr;
  bf16_t* op = out + (long long)img * hw * out_ld + ch;
  for (int p = p0 + row; p < p1; p += rows * UNR) {
    uint4 raw[UNR], rr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int q = p + u * rows;
      if (q < p1) {
        raw[u] = *reinterpret_cast<const uint4*>(xp + (long long)q * ld);
        if (rp) rr[u] = *reinterpret_cast<const uint4*>(rp + (long long)q * res_ld);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int q = p + u * rows;
      if (q >= p1) continue;
      const unsigned in[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
      const unsigned rs[4] = {rr[u].x, rr[u].y, rr[u].z, rr[u].w};
      unsigned o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = fmaf(__uint_as_float(in[k] << 16), sc[2 * k], sh[2 * k]);
        float b = fmaf(__uint_as_float(in[k] & 0xFFFF0000u), sc[2 * k + 1], sh[2 * k + 1]);
        if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        if (rp) { a += __uint_as_float(rs[k] << 16); b += __uint_as_float(rs[k] & 0xFFFF0000u); }
        o[k] = pk_bf16(a, b);
      }
      *reinterpret_cast<uint4*>(op + (long long)q * out_ld) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ---- row softmax: S (fp32 or bf16) -> P bf16; one workgroup per row, cols <= 256 * 8 * MAXV ----
template <int MAXV, typename TS>
__global__ __launch_bounds__(256) void softmax16_kernel(const TS* s, int cols, int s_ld, bf16_t* p, int p_ld) {
  const TS* row = s + (long long)blockIdx.x * s_ld;
  bf16_t* orow = p + (long long)blockIdx.x * p_ld;
  const int c8 = cols >> 3;
  float v[MAXV][8];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c8) {
      if (sizeof(TS) == 4) ld8f(reinterpret_cast<const float*>(row) + 8 * e, v[i]);
      else ld8(reinterpret_cast<const bf16_t*>(row) + 8 * e, v[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) m = fmaxf(m, v[i][k]);
    }
  }
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[i][k] = expf(v[i][k] - m); sum += v[i][k]; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < c8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[i][k] *= inv;
      st8(orow + 8 * e, v[i]);
    }
  }
}

// codebook gather: out[r][:] = bf16(table[idx[r]][:])   (fp32 table, model/codebook.py:41)
__global__ __launch_bounds__(256) void gather_rows16_kernel(const float* table, int dim, const int32_t* idx, long long rows, bf16_t* out, int out_ld) {
  const int d8 = dim >> 3;
  const long long total = rows * d8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / d8; const int j = (int)(e % d8);
    float v[8];
    ld8f(table + (long long)idx[r] * dim + 8 * j, v);
    st8(out + r * out_ld + 8 * j, v);
  }
}

// NHWC rows [n][rows][c] -> the B-operand layout of gpemsr_conv2d_bf16's 1x1 form: [n][c/8][rows][8]
// perm16: position p of every 16-row group holds row 8 ((p & 7) >> 2) + 4 (p >> 3) + (p & 3), i.e. the rows in the order 0-3, 8-11, 4-7,
// 12-15 -- the order in which a 32x32 MFMA accumulator holds them, so that a product whose COLUMNS these rows become (v^T of the
// attention block) comes out in the k-slot order of gpemsr_flash_attention_bf16's second product (csrc/attn_bf16.hip)
__global__ __launch_bounds__(256) void pack_rows16_kernel(const bf16_t* src, int rows, int c, int ld, long long img_stride, bf16_t* dst, long long total, int perm16) {
  const int c8 = c >> 3;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e % rows);
    const long long t = e / rows;
    const int j = (int)(t % c8);
    const long long img = t / c8;
    const int p = r & 15;
    const int rs = perm16 ? (r & ~15) + 8 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3) : r;
    *reinterpret_cast<uint4*>(dst + ((img * c8 + j) * rows + r) * 8) = *reinterpret_cast<const uint4*>(src + img * img_stride + (long long)rs * ld + 8 * j);
  }
}

// nn.MaxPool2d(2, 2), floor mode (R:model/VGG.py slices 2-3 of the loss network), 8 channels per thread
__global__ __launch_bounds__(256) void maxpool2_16_kernel(const bf16_t* x, int n, int h, int w, int c, int ld, bf16_t* out, int out_ld) {
  const int oh = h >> 1, ow = w >> 1, c8 = c >> 3;
  const long long total = (long long)n * oh * ow * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c8);
    long long t = e / c8;
    const int ox = (int)(t % ow); t /= ow;
    const int oy = (int)(t % oh);
    const long long img = t / oh;
    const bf16_t* p = x + ((img * h + 2 * oy) * w + 2 * ox) * (long long)ld + 8 * j;
    float a[8], b[8], cc[8], d[8];
    ld8(p, a); ld8(p + ld, b); ld8(p + (long long)w * ld, cc); ld8(p + (long long)w * ld + ld, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(cc[k], d[k]));
    st8(out + ((img * oh + oy) * ow + ox) * (long long)out_ld + 8 * j, a);
  }
}

// ---- resampling ----
__device__ __forceinline__ void src_index16(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float s = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - i0;
}

// 8 channels per thread
__global__ __launch_bounds__(256) void bilinear16_kernel(const bf16_t* x, int n, int h, int w, int c, int ld, int oh, int ow,
                                                         int align, float sh, float sw, float mul, bf16_t* out, int out_ld) {
  const int c8 = c >> 3;
  const long long total = (long long)n * oh * ow * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = 8 * (int)(e % c8);
    long long p = e / c8;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    int y0, y1, x0, x1; float ly, lx;
    src_index16(oy, sh, align, h, y0, y1, ly);
    src_index16(ox, sw, align, w, x0, x1, lx);
    const bf16_t* b = x + (long long)img * h * w * ld + ch;
    float v00[8], v01[8], v10[8], v11[8], r[8];
    ld8(b + ((long long)y0 * w + x0) * ld, v00); ld8(b + ((long long)y0 * w + x1) * ld, v01);
    ld8(b + ((long long)y1 * w + x0) * ld, v10); ld8(b + ((long long)y1 * w + x1) * ld, v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k])) * mul;
    st8(out + (((long long)img * oh + oy) * ow + ox) * out_ld + ch, r);
  }
}

__global__ __launch_bounds__(256) void pool3s2_16_kernel(const bf16_t* x, int n, int h, int w, int c, int ld, bf16_t* out, int out_ld) {
  const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1, c8 = c >> 3;
  const long long total = (long long)n * oh * ow * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = 8 * (int)(e % c8);
    long long p = e / c8;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    float mx[8], sm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mx[k] = -INFINITY; sm[k] = 0.f; }
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= h) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= w) continue;
        float v[8];
        ld8(x + (((long long)img * h + iy) * w + ix) * ld + ch, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) { mx[k] = fmaxf(mx[k], v[k]); sm[k] += v[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[k] *= (1.f / 9.f);      // AvgPool2d default count_include_pad=True
    bf16_t* o = out + (((long long)img * oh + oy) * ow + ox) * out_ld;
    st8(o + ch, mx);
    st8(o + c + ch, sm);
  }
}

// One SpyNet level input (see gpemsr_spynet_prep): fp32 frames / flow in, bf16 16-channel conv input out (channels 8..15 zero)
__global__ __launch_bounds__(256) void spynet_prep16_kernel(const float* ref, const float* supp, const float* fc, int n, int h, int w,
                                                            float m0, float m1, float m2, float s0, float s1, float s2,
                                                            float* up, bf16_t* inp) {
  const long long total = (long long)n * h * w;
  const int ch2 = h / 2, cw2 = w / 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int xq = (int)(e % w);
    const int yq = (int)((e / w) % h);
    const int img = (int)(e / ((long long)w * h));
    float fx = 0.f, fy = 0.f;
    if (fc) {
      const int uh = 2 * ch2, uw = 2 * cw2;
      const float sh = uh > 1 ? (float)(ch2 - 1) / (float)(uh - 1) : 0.f;
      const float sw = uw > 1 ? (float)(cw2 - 1) / (float)(uw - 1) : 0.f;
      const int yu = yq < uh ? yq : uh - 1, xu = xq < uw ? xq : uw - 1;
      int y0, y1, x0, x1; float ly, lx;
      src_index16(yu, sh, 1, ch2, y0, y1, ly);
      src_index16(xu, sw, 1, cw2, x0, x1, lx);
      const float* b = fc + (long long)img * ch2 * cw2 * 2;
      const float hy = 1.f - ly, hx = 1.f - lx;
#define GP_F(yy, xx, k) b[((long long)(yy) * cw2 + (xx)) * 2 + (k)]
      fx = (hy * (hx * GP_F(y0, x0, 0) + lx * GP_F(y0, x1, 0)) + ly * (hx * GP_F(y1, x0, 0) + lx * GP_F(y1, x1, 0))) * 2.f;
      fy = (hy * (hx * GP_F(y0, x0, 1) + lx * GP_F(y0, x1, 1)) + ly * (hx * GP_F(y1, x0, 1) + lx * GP_F(y1, x1, 1))) * 2.f;
#undef GP_F
    }
    up[2 * e] = fx; up[2 * e + 1] = fy;
    float sx = fminf(fmaxf((float)xq + fx, 0.f), (float)(w - 1));
    float sy = fminf(fmaxf((float)yq + fy, 0.f), (float)(h - 1));
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const float lx = sx - x0, ly = sy - y0;
    const float* sp = supp + (long long)img * h * w;
    const int x1 = x0 + 1, y1 = y0 + 1;
    float wv = 0.f;
    wv += (1.f - ly) * (1.f - lx) * sp[(long long)y0 * w + x0];
    if (x1 <= w - 1) wv += (1.f - ly) * lx * sp[(long long)y0 * w + x1];
    if (y1 <= h - 1) wv += ly * (1.f - lx) * sp[(long long)y1 * w + x0];
    if (x1 <= w - 1 && y1 <= h - 1) wv += ly * lx * sp[(long long)y1 * w + x1];
    const float rv = ref[e];
    const float o[8] = {(rv - m0) / s0, (rv - m1) / s1, (rv - m2) / s2, (wv - m0) / s0, (wv - m1) / s1, (wv - m2) / s2, fx, fy};
    st8(inp + e * 16, o);
    *reinterpret_cast<uint4*>(inp + e * 16 + 8) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Modulated deformable column gather: x bf16, offsets / mask logits fp32 (coordinates keep full precision), columns bf16
__global__ __launch_bounds__(256) void dcn_columns16_kernel(const bf16_t* x, int n, int h, int w, int c, int ld,
                                                            const float* om, int om_ld, int groups, bf16_t* col) {
  const int cg = c / groups;            // 8
  const int K = 9;
  const long long total = (long long)n * h * w * groups * K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int g = (int)(e % groups);
    long long t = e / groups;
    const int k = (int)(t % K); t /= K;
    const long long pix = t;
    const int xq = (int)(pix % w);
    const int yq = (int)((pix / w) % h);
    const int img = (int)(pix / ((long long)w * h));
    const float* o = om + pix * om_ld;
    const float dy = o[g * 2 * K + 2 * k], dx = o[g * 2 * K + 2 * k + 1];
    const float ml = o[2 * groups * K + g * K + k];
    const float m = 1.f / (1.f + expf(-ml));
    const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dx;
    // BRANCH-FREE: the four corner rows are always fetched (clamped addresses, all four loads in flight together); corners outside the
    // image and sampling points outside (-1, h) x (-1, w) get weight 0 -- the same sums in the same order as the branchy form
    const bool inside = py > -1.f && py < (float)h && px > -1.f && px < (float)w;
    const float fy = floorf(py), fx = floorf(px);
    const float ly = py - fy, lx = px - fx;
    const int y0 = (int)fmaxf(fminf(fy, (float)h), -2.f), x0 = (int)fmaxf(fminf(fx, (float)w), -2.f);
    const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
    const int ys[4] = {y0, y0, y0 + 1, y0 + 1}, xs[4] = {x0, x0 + 1, x0, x0 + 1};
    float v[4][8], wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = inside && ys[q] >= 0 && ys[q] <= h - 1 && xs[q] >= 0 && xs[q] <= w - 1;
      const int yc = min(max(ys[q], 0), h - 1), xc = min(max(xs[q], 0), w - 1);
      ld8(x + (((long long)img * h + yc) * w + xc) * ld + g * cg, v[q]);
      wq[q] = ok ? wts[q] : 0.f;
    }
    float acc[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) acc[z] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int z = 0; z < 8; ++z) acc[z] += wq[q] * v[q][z];
#pragma unroll
    for (int z = 0; z < 8; ++z) acc[z] *= m;
    st8(col + pix * (long long)(K * c) + k * c + g * cg, acc);
  }
}

// ---- GPEMSR-specific fusions ----
// 16x16-patch cosine of two bf16 feature maps (un-fused fallback of gpemsr_vgg_mask_bf16; model/GPEMSR.py:387-395)
__global__ __launch_bounds__(256) void patch_cosine16_kernel(const bf16_t* a, const bf16_t* b, int h, int w, int c, float* out) {
  const int pw = w / 16, ph = h / 16;
  const int px = blockIdx.x % pw, py = (blockIdx.x / pw) % ph, img = blockIdx.x / (pw * ph);
  const int c8 = c >> 3;
  const int total = 256 * c8;
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int j = e % c8, p = e / c8;
    const long long off = (((long long)img * h + py * 16 + (p >> 4)) * w + px * 16 + (p & 15)) * c + 8 * j;
    float u[8], v[8];
    ld8(a + off, u); ld8(b + off, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) { dot += u[k] * v[k]; na += u[k] * u[k]; nb += v[k] * v[k]; }
  }
  __shared__ float red[3][4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { dot += __shfl_xor(dot, o); na += __shfl_xor(na, o); nb += __shfl_xor(nb, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dot; red[1][threadIdx.x >> 6] = na; red[2][threadIdx.x >> 6] = nb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float d = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float x = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float y = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    out[blockIdx.x] = d / (fmaxf(sqrtf(x), 1e-12f) * fmaxf(sqrtf(y), 1e-12f));
  }
}

// ThreeDA temporal gate, c == 64: 8 lanes x 8 channels per (b,t,pixel)
__global__ __launch_bounds__(256) void temporal_gate16_kernel(const bf16_t* aligned, const bf16_t* emb, const bf16_t* emb_ref,
                                                              int b, int t, int hw, int c, bf16_t* af) {
  const int sub = threadIdx.x & 7;
  const long long items = (long long)b * t * hw;
  const long long per_iter = (long long)gridDim.x * 32;
  const long long niter = (items + per_iter - 1) / per_iter;
  for (long long it = 0; it < niter; ++it) {
    const long long item = it * per_iter + (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool valid = item < items;
    const long long ii = valid ? item : 0;
    const int p = (int)(ii % hw);
    const int ti = (int)((ii / hw) % t);
    const int bi = (int)(ii / ((long long)hw * t));
    float e1[8], e0[8];
    ld8(emb + ii * c + 8 * sub, e1);
    ld8(emb_ref + ((long long)bi * hw + p) * c + 8 * sub, e0);
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += e1[k] * e0[k];
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) d += __shfl_xor(d, m);
    const float g = 1.f / (1.f + expf(-d));
    if (valid) {
      float v[8];
      ld8(aligned + ii * c + 8 * sub, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= g;
      st8(af + ((long long)bi * hw + p) * ((long long)t * c) + ti * c + 8 * sub, v);
    }
  }
}

// TT > 0: the frame count at compile time (the network's 5): loops unroll and in[][] stays in registers -- with run-time bounds the
// array is indexed dynamically and lives in scratch (272 bytes per lane)
template <int TT>
__global__ __launch_bounds__(256) void frame_mix16_kernel(const bf16_t* af, long long pixels, int t_rt, int c, const float* m,
                                                          const float* bias, bf16_t* out) {
  const int t = TT > 0 ? TT : t_rt;
  const int c8 = c >> 3;
  const long long total = pixels * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c8);
    const long long p = e / c8;
    const bf16_t* ip = af + p * ((long long)t * c) + 8 * j;
    float in[TT > 0 ? TT : 8][8];
#pragma unroll
    for (int k = 0; k < t; ++k) ld8(ip + k * c, in[k]);
#pragma unroll
    for (int i = 0; i < t; ++i) {
      float s[8];
#pragma unroll
      for (int z = 0; z < 8; ++z) s[z] = bias[i];
#pragma unroll
      for (int k = 0; k < t; ++k) {
        const float wv = m[i * t + k];
#pragma unroll
        for (int z = 0; z < 8; ++z) s[z] = fmaf(wv, in[k][z], s[z]);
      }
#pragma unroll
      for (int z = 0; z < 8; ++z) s[z] = s[z] > 0.f ? s[z] : 0.1f * s[z];
      st8(out + p * ((long long)t * c) + i * c + 8 * j, s);
    }
  }
}

__global__ __launch_bounds__(256) void threeda_combine16_kernel(const bf16_t* feat, const bf16_t* attn, const bf16_t* add,
                                                                const bf16_t* f2, const bf16_t* f3, long long count8, bf16_t* out) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count8; e += (long long)gridDim.x * 256) {
    float f[8], a[8], d[8], u[8], v[8], r[8];
    ld8(feat + 8 * e, f); ld8(attn + 8 * e, a); ld8(add + 8 * e, d); ld8(f2 + 8 * e, u); ld8(f3 + 8 * e, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = f[k] * (1.f / (1.f + expf(-a[k]))) * 2.f + d[k] + u[k] + v[k];
    st8(out + 8 * e, r);
  }
}

__global__ __launch_bounds__(256) void copy_channels16_kernel(const bf16_t* src, int src_ld, bf16_t* dst, int dst_ld, long long pixels, int c8) {
  const long long total = pixels * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c8; const int j = (int)(e % c8);
    *reinterpret_cast<uint4*>(dst + p * dst_ld + 8 * j) = *reinterpret_cast<const uint4*>(src + p * src_ld + 8 * j);
  }
}

// fp32 channels -> bf16 channel slice (assembling the flow / frame concat buffer of the POD offset convs)
__global__ __launch_bounds__(256) void copy_channels_f32_bf16_kernel(const float* src, int src_ld, bf16_t* dst, int dst_ld, long long pixels, int c) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    dst[p * dst_ld + ch] = to_bf16(src[p * src_ld + ch]);
  }
}

// ---- small direct convolutions with mixed formats ----
// 1 -> COUT stem (fp32 1-channel image in, bf16 out): conv_first, indexer stem, refmaskconv1, VGG conv1_1 (un-fused path)
__global__ __launch_bounds__(256) void conv_stem1_16_kernel(const float* x, int n, int h, int w, const float* weight, const float* bias,
                                                            int cout, int cin_pad, int act, bf16_t* out, int out_ld) {
  extern __shared__ __attribute__((aligned(16))) float wsm16[];   // [9][cout] then bias[cout]
  for (int i = threadIdx.x; i < 9 * cout; i += 256) wsm16[i] = weight[((long long)(i / cout) * cout + (i % cout)) * cin_pad];
  for (int i = threadIdx.x; i < cout; i += 256) wsm16[9 * cout + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  constexpr int SRUN = 4;
  const int c8 = cout >> 3;
  const int runs = (w + SRUN - 1) / SRUN;
  const long long total = (long long)n * h * runs * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c8);
    const long long rp = e / c8;
    const int ox0 = (int)(rp % runs) * SRUN, oy = (int)((rp / runs) % h);
    const long long img = rp / ((long long)runs * h);
    const float* xp = x + img * h * w;
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
      float acc[8];
#pragma unroll
      for (int z = 0; z < 8; ++z) acc[z] = wsm16[9 * cout + 8 * j + z];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = win[ky][r + kx];
          const float* ww = wsm16 + (ky * 3 + kx) * cout + 8 * j;
#pragma unroll
          for (int z = 0; z < 8; ++z) acc[z] = fmaf(v, ww[z], acc[z]);
        }
#pragma unroll
      for (int z = 0; z < 8; ++z) acc[z] = apply_act(acc[z], act);
      st8(out + ((img * h + oy) * w + ox0 + r) * out_ld + 8 * j, acc);
    }
  }
}

// 64 -> 1, 3x3 stride 1, bf16 in -> fp32 out (+ fp32 residual): decoder.output_layer and conv_last at 1024^2.  8 lanes own
// the 64 channels of a pixel (16 B each); a lane group walks RUN consecutive output columns (see conv_c64_cout1_kernel).
template <int RUN>
__global__ __launch_bounds__(256) void conv_c64_cout1_16_kernel(const bf16_t* x, int n, int h, int w, int ld, const float* weight, int cin_pad,
                                                                const float* bias, int act, const float* residual, int res_ld,
                                                                float* out, int out_ld) {
  __shared__ __attribute__((aligned(16))) float wsm[9 * 64];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) wsm[i] = weight[(i / 64) * cin_pad + (i % 64)];
  __syncthreads();
  const int sub = threadIdx.x & 7;
  float wv[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int z = 0; z < 8; ++z) wv[t][z] = wsm[t * 64 + 8 * sub + z];
  const int runs_x = (w + RUN - 1) / RUN;
  const long long nrun = (long long)n * h * runs_x;
  const float bs = bias ? bias[0] : 0.f;
  for (long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); r < ((nrun + 31) / 32) * 32; r += (long long)gridDim.x * 32) {
    const bool live = r < nrun;
    const long long rr = live ? r : 0;
    const int x0 = (int)(rr % runs_x) * RUN;
    const int oy = (int)((rr / runs_x) % h);
    const int img = (int)(rr / ((long long)runs_x * h));
    float acc[RUN];
#pragma unroll
    for (int j = 0; j < RUN; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy - 1 + ky;
      if (iy < 0 || iy >= h) continue;
      const bf16_t* rowp = x + (((long long)img * h + iy) * w) * ld + 8 * sub;
#pragma unroll
      for (int c = 0; c < RUN + 2; ++c) {
        const int ix = x0 - 1 + c;
        float v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) v[z] = 0.f;
        if (ix >= 0 && ix < w) ld8(rowp + (long long)ix * ld, v);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int j = c - kx;
          if (j >= 0 && j < RUN) {
#pragma unroll
            for (int z = 0; z < 8; ++z) acc[j] = fmaf(v[z], wv[ky * 3 + kx][z], acc[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RUN; ++j)
#pragma unroll
      for (int m = 4; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m);
    if (live) {
#pragma unroll
      for (int j = 0; j < RUN; ++j) {
        const int ox = x0 + j;
        if ((j & 7) == sub && ox < w) {
          const long long pix = ((long long)img * h + oy) * w + ox;
          float v = apply_act(acc[j] + bs, act);
          if (residual) v += residual[pix * res_ld];
          out[pix * out_ld] = v;
        }
      }
    }
  }
}

// generic tiny conv (cout <= 16, any k <= 7, stride 1/2/4): one thread per (pixel, cout); TI / TO in {float, bf16_t}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_direct16_kernel(const TI* x, int n, int h, int w, int ld, int cin, int cin_pad, const float* weight,
                                                            const float* bias, int cout, int ksize, int stride, int oh, int ow, int act,
                                                            TO* out, int out_ld) {
  const int pad = ksize / 2;
  const long long total = (long long)n * oh * ow * cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int co = (int)(e % cout);
    const long long pix = e / cout;
    const int ox = (int)(pix % ow);
    const int oy = (int)((pix / ow) % oh);
    const int img = (int)(pix / ((long long)ow * oh));
    float acc = 0.f;
    for (int ky = 0; ky < ksize; ++ky) {
      const int iy = oy * stride - pad + ky;
      if (iy < 0 || iy >= h) continue;
      for (int kx = 0; kx < ksize; ++kx) {
        const int ix = ox * stride - pad + kx;
        if (ix < 0 || ix >= w) continue;
        const TI* xp = x + (((long long)img * h + iy) * w + ix) * ld;
        const float* wp = weight + ((long long)(ky * ksize + kx) * cout + co) * cin_pad;
        for (int ci = 0; ci < cin; ++ci) {
          const float xv = sizeof(TI) == 4 ? (float)reinterpret_cast<const float*>(xp)[ci] : from_bf16(reinterpret_cast<const bf16_t*>(xp)[ci]);
          acc = fmaf(xv, wp[ci], acc);
        }
      }
    }
    const float v = apply_act(acc + (bias ? bias[co] : 0.f), act);
    if (sizeof(TO) == 4) reinterpret_cast<float*>(out)[pix * out_ld + co] = v;
    else reinterpret_cast<bf16_t*>(out)[pix * out_ld + co] = to_bf16(v);
  }
}

// small 3x3 convolution, cin in {2 (fp32 flow), 16 (bf16)} -> cout <= 16, stride 1/2/4 (the flow down-convs, R:model/GPEMSR.py:70-75):
// one thread per output pixel holds all couts; the weights sit in LDS as [tap][cin][16 couts] and are read as broadcasts.
template <typename TI, int CIN, typename TO>
__global__ __launch_bounds__(256) void conv_small16_kernel(const TI* x, int n, int h, int w, int ld, int cin_pad, const float* weight, const float* bias,
                                                           int cout, int stride, int oh, int ow, int act, TO* out, int out_ld, int vec_out) {
  __shared__ __attribute__((aligned(16))) float wsm[9 * CIN * 16];
  for (int i = threadIdx.x; i < 9 * CIN * 16; i += 256) {
    const int co = i & 15, ci = (i >> 4) % CIN, tap = i / (16 * CIN);
    wsm[i] = co < cout ? weight[((long long)tap * cout + co) * cin_pad + ci] : 0.f;
  }
  __syncthreads();
  float bs[16];
#pragma unroll
  for (int co = 0; co < 16; ++co) bs[co] = (bias && co < cout) ? bias[co] : 0.f;
  const long long total = (long long)n * oh * ow;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int ox = (int)(pix % ow), oy = (int)((pix / ow) % oh), img = (int)(pix / ((long long)ow * oh));
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = bs[co];
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {              // (fully unrolled, the nine taps' inputs were all kept live: 356 bytes of scratch per lane)
      const int iy = oy * stride - 1 + ky;
      if (iy < 0 || iy >= h) continue;
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride - 1 + kx;
        if (ix < 0 || ix >= w) continue;
        const TI* xp = x + (((long long)img * h + iy) * w + ix) * ld;
        float v[CIN];
        if (sizeof(TI) == 4) {
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) v[ci] = reinterpret_cast<const float*>(xp)[ci];
        } else {
#pragma unroll
          for (int c8 = 0; c8 < CIN / 8; ++c8) {
            float t[8];
            ld8(reinterpret_cast<const bf16_t*>(xp) + 8 * c8, t);
#pragma unroll
            for (int z = 0; z < 8; ++z) v[8 * c8 + z] = t[z];
          }
        }
        const float* wp = wsm + (ky * 3 + kx) * CIN * 16;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wp + ci * 16 + 4 * c4);
            acc[4 * c4] = fmaf(v[ci], w4.x, acc[4 * c4]); acc[4 * c4 + 1] = fmaf(v[ci], w4.y, acc[4 * c4 + 1]);
            acc[4 * c4 + 2] = fmaf(v[ci], w4.z, acc[4 * c4 + 2]); acc[4 * c4 + 3] = fmaf(v[ci], w4.w, acc[4 * c4 + 3]);
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = apply_act(acc[co], act);
    TO* op = out + pix * out_ld;
    if (sizeof(TO) == 2 && vec_out) {
      float lo[8], hi[8];
#pragma unroll
      for (int z = 0; z < 8; ++z) { lo[z] = acc[z]; hi[z] = acc[8 + z]; }
      st8(reinterpret_cast<bf16_t*>(op), lo); st8(reinterpret_cast<bf16_t*>(op) + 8, hi);
    } else {
#pragma unroll
      for (int co = 0; co < 16; ++co)
        if (co < cout) {
          if (sizeof(TO) == 4) reinterpret_cast<float*>(op)[co] = acc[co];
          else reinterpret_cast<bf16_t*>(op)[co] = to_bf16(acc[co]);
        }
    }
  }
}

}  // namespace gpemsr

using namespace gpemsr;
#define ST(s) reinterpret_cast<hipStream_t>(s)
#define A16(p) ((reinterpret_cast<uintptr_t>(p) & 15) == 0)

extern "C" int gpemsr_cast_f32_bf16(const float* x, int64_t pixels, int c, int x_ld, void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && pixels > 0 && c > 0, "cast_f32_bf16: bad args");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid16(pixels * c)), dim3(256), 0, ST(stream), x, (long long)pixels, c, x_ld, reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("cast_f32_bf16");
}
extern "C" int gpemsr_cast_bf16_f32(const void* x, int64_t pixels, int c, int x_ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && pixels > 0 && c > 0, "cast_bf16_f32: bad args");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid16(pixels * c)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), (long long)pixels, c, x_ld, out, out_ld);
  return check_launch("cast_bf16_f32");
}

extern "C" int gpemsr_split_f32_bf16x2(const float* x, int64_t pixels, int c, int x_ld, void* hi, int hi_ld, void* lo, int lo_ld, void* stream) {
  GP_REQUIRE(x && hi && lo && pixels > 0 && c > 0, "split_f32_bf16x2: bad args");
  GP_REQUIRE(c % 4 == 0 && x_ld % 4 == 0 && hi_ld % 4 == 0 && lo_ld % 4 == 0 && A16(x) && A16(hi) && A16(lo), "split_f32_bf16x2: channels and row strides must be multiples of 4, bases 16-byte aligned");
  hipLaunchKernelGGL(split_f32_bf16x2_kernel, dim3(grid16(pixels * (c / 4))), dim3(256), 0, ST(stream), x, (long long)pixels, c / 4, x_ld,
                     reinterpret_cast<bf16_t*>(hi), hi_ld, reinterpret_cast<bf16_t*>(lo), lo_ld);
  return check_launch("split_f32_bf16x2");
}

extern "C" int gpemsr_groupnorm_stats_bf16(const void* x, int n, int hw, int c, int ld, float* ws, int parts, void* stream) {
  GP_REQUIRE(x && ws && c % 8 == 0 && (c / 8) <= 256 && ld % 8 == 0 && parts >= 1 && A16(x), "groupnorm_stats_bf16: bad args (c%%8, c<=2048, ld%%8)");
  hipLaunchKernelGGL(gn_partial16_kernel, dim3(parts, n), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), hw, c, ld, parts, ws);
  return check_launch("groupnorm_stats_bf16");
}

extern "C" int gpemsr_groupnorm_apply_bf16(const void* x, int n, int hw, int c, int ld, int groups, const float* mean_rstd,
                                           const float* gamma, const float* beta, int relu, const void* residual, int res_ld,
                                           void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && mean_rstd && gamma && beta && out, "groupnorm_apply_bf16: null pointer");
  GP_REQUIRE(c % 8 == 0 && ld % 8 == 0 && out_ld % 8 == 0 && (!residual || res_ld % 8 == 0) && c % groups == 0 && A16(x) && A16(out) && A16(gamma) && A16(beta),
             "groupnorm_apply_bf16: alignment");
  const int c8 = c / 8;
  if (c8 <= 256 && 256 % c8 == 0) {
    const int rows = 256 / c8;
    // >= 4 waves of work per CU over the whole launch, >= 4 x UNR pixel rows per thread
    int blocks = (2048 + n - 1) / n;
    const int max_blocks = (hw + rows * 16 - 1) / (rows * 16);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    int per = (hw + blocks - 1) / blocks;
    per = (per + rows - 1) / rows * rows;
    blocks = (hw + per - 1) / per;
    hipLaunchKernelGGL(gn_apply16_cols_kernel<4>, dim3(blocks, n), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), hw, c, ld, groups, mean_rstd, gamma, beta,
                       relu, reinterpret_cast<const bf16_t*>(residual), res_ld, reinterpret_cast<bf16_t*>(out), out_ld, per);
    return check_launch("groupnorm_apply_bf16");
  }
  const long long total8 = (long long)n * hw * (c / 8);
  hipLaunchKernelGGL(gn_apply16_kernel, dim3(grid16(total8)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), total8, hw, c, ld, groups,
                     mean_rstd, gamma, beta, relu, reinterpret_cast<const bf16_t*>(residual), res_ld, reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("groupnorm_apply_bf16");
}

extern "C" int gpemsr_softmax_rows_bf16(const void* s, int s_f32, int64_t rows, int cols, int s_ld, void* p, int p_ld, void* stream) {
  GP_REQUIRE(s && p && rows > 0 && rows < (1ll << 31) && cols % 8 == 0 && cols <= 256 * 8 * 8 && s_ld % 8 == 0 && p_ld % 8 == 0 && A16(s) && A16(p),
             "softmax_rows_bf16: bad args (cols %% 8 == 0, <= 16384)");
  hipStream_t st = ST(stream);
  bf16_t* pp = reinterpret_cast<bf16_t*>(p);
  if (s_f32) {
    if (cols <= 2048) hipLaunchKernelGGL((softmax16_kernel<1, float>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const float*>(s), cols, s_ld, pp, p_ld);
    else if (cols <= 4096) hipLaunchKernelGGL((softmax16_kernel<2, float>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const float*>(s), cols, s_ld, pp, p_ld);
    else if (cols <= 8192) hipLaunchKernelGGL((softmax16_kernel<4, float>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const float*>(s), cols, s_ld, pp, p_ld);
    else hipLaunchKernelGGL((softmax16_kernel<8, float>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const float*>(s), cols, s_ld, pp, p_ld);
  } else {
    if (cols <= 2048) hipLaunchKernelGGL((softmax16_kernel<1, bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(s), cols, s_ld, pp, p_ld);
    else if (cols <= 4096) hipLaunchKernelGGL((softmax16_kernel<2, bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(s), cols, s_ld, pp, p_ld);
    else if (cols <= 8192) hipLaunchKernelGGL((softmax16_kernel<4, bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(s), cols, s_ld, pp, p_ld);
    else hipLaunchKernelGGL((softmax16_kernel<8, bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(s), cols, s_ld, pp, p_ld);
  }
  return check_launch("softmax_rows_bf16");
}

extern "C" int gpemsr_gather_rows_bf16(const float* table, int dim, const int32_t* idx, int64_t rows, void* out, int out_ld, void* stream) {
  GP_REQUIRE(table && idx && out && dim % 8 == 0 && out_ld % 8 == 0 && A16(out) && A16(table), "gather_rows_bf16: bad args");
  hipLaunchKernelGGL(gather_rows16_kernel, dim3(grid16(rows * (dim / 8))), dim3(256), 0, ST(stream), table, dim, idx, (long long)rows, reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("gather_rows_bf16");
}

extern "C" int gpemsr_pack_rows_bf16_ex(const void* src, int n, int rows, int c, int ld, int64_t img_stride, void* dst, int perm16, void* stream) {
  GP_REQUIRE(src && dst && n > 0 && rows > 0 && c % 8 == 0 && ld % 8 == 0 && img_stride % 8 == 0 && A16(src) && A16(dst) && (!perm16 || rows % 16 == 0),
             "pack_rows_bf16: bad args (perm16 needs rows %% 16 == 0)");
  const long long total = (long long)n * (c / 8) * rows;
  hipLaunchKernelGGL(pack_rows16_kernel, dim3(grid16(total)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(src), rows, c, ld, (long long)img_stride,
                     reinterpret_cast<bf16_t*>(dst), total, perm16 ? 1 : 0);
  return check_launch("pack_rows_bf16");
}
extern "C" int gpemsr_pack_rows_bf16(const void* src, int n, int rows, int c, int ld, int64_t img_stride, void* dst, void* stream) {
  return gpemsr_pack_rows_bf16_ex(src, n, rows, c, ld, img_stride, dst, 0, stream);
}

extern "C" int gpemsr_maxpool2_bf16(const void* x, int n, int h, int w, int c, int ld, void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && n > 0 && h >= 2 && w >= 2 && c % 8 == 0 && ld % 8 == 0 && out_ld % 8 == 0 && A16(x) && A16(out), "maxpool2_bf16: bad args");
  hipLaunchKernelGGL(maxpool2_16_kernel, dim3(grid16((long long)n * (h / 2) * (w / 2) * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), n, h, w, c, ld,
                     reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("maxpool2_bf16");
}

extern "C" int gpemsr_bilinear_bf16(const void* x, int n, int h, int w, int c, int ld, int oh, int ow, int align_corners, float mul,
                                    void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && n > 0 && h > 0 && w > 0 && c % 8 == 0 && ld % 8 == 0 && out_ld % 8 == 0 && oh > 0 && ow > 0 && A16(x) && A16(out), "bilinear_bf16: bad args");
  float sh, sw;
  if (align_corners) { sh = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f; sw = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f; }
  else { sh = (float)((double)h / (double)oh); sw = (float)((double)w / (double)ow); }
  hipLaunchKernelGGL(bilinear16_kernel, dim3(grid16((long long)n * oh * ow * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), n, h, w, c, ld,
                     oh, ow, align_corners, sh, sw, mul, reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("bilinear_bf16");
}

extern "C" int gpemsr_pool3s2_maxavg_bf16(const void* x, int n, int h, int w, int c, int ld, void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && c % 8 == 0 && ld % 8 == 0 && out_ld % 8 == 0 && out_ld >= 2 * c && A16(x) && A16(out), "pool3s2_bf16: bad args");
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  hipLaunchKernelGGL(pool3s2_16_kernel, dim3(grid16((long long)n * oh * ow * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), n, h, w, c, ld,
                     reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("pool3s2_bf16");
}

extern "C" int gpemsr_spynet_prep_bf16(const float* ref, const float* supp, const float* flow_coarse, int n, int h, int w,
                                       const float* mean3, const float* std3, float* up_flow, void* inp16, void* stream) {
  GP_REQUIRE(ref && supp && up_flow && inp16 && mean3 && std3 && A16(inp16), "spynet_prep_bf16: null pointer / alignment");
  GP_REQUIRE(h >= 2 && w >= 2, "spynet_prep_bf16: level smaller than 2x2");
  hipLaunchKernelGGL(spynet_prep16_kernel, dim3(grid16((long long)n * h * w)), dim3(256), 0, ST(stream), ref, supp, flow_coarse, n, h, w,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], up_flow, reinterpret_cast<bf16_t*>(inp16));
  return check_launch("spynet_prep_bf16");
}

extern "C" int gpemsr_dcn_columns_bf16(const void* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups, void* col, void* stream) {
  GP_REQUIRE(x && om && col && A16(x) && A16(col), "dcn_columns_bf16: null pointer / alignment");
  GP_REQUIRE(groups > 0 && c % groups == 0 && c / groups == 8 && ld % 8 == 0, "dcn_columns_bf16: needs 8 channels per deformable group");
  GP_REQUIRE(om_ld >= 3 * groups * 9, "dcn_columns_bf16: om_ld too small");
  hipLaunchKernelGGL(dcn_columns16_kernel, dim3(grid16((long long)n * h * w * groups * 9)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), n, h, w, c, ld,
                     om, om_ld, groups, reinterpret_cast<bf16_t*>(col));
  return check_launch("dcn_columns_bf16");
}

extern "C" int gpemsr_patch_cosine_bf16(const void* a, const void* b, int n, int h, int w, int c, float* out, void* stream) {
  GP_REQUIRE(a && b && out && h % 16 == 0 && w % 16 == 0 && c % 8 == 0 && A16(a) && A16(b), "patch_cosine_bf16: needs h,w multiples of 16, c%%8==0");
  hipLaunchKernelGGL(patch_cosine16_kernel, dim3(n * (h / 16) * (w / 16)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(a),
                     reinterpret_cast<const bf16_t*>(b), h, w, c, out);
  return check_launch("patch_cosine_bf16");
}

extern "C" int gpemsr_temporal_gate_bf16(const void* aligned, const void* emb, const void* emb_ref, int b, int t, int hw, int c, void* af, void* stream) {
  GP_REQUIRE(aligned && emb && emb_ref && af && c == 64 && A16(aligned) && A16(emb) && A16(emb_ref) && A16(af), "temporal_gate_bf16: c must be 64");
  const long long items = (long long)b * t * hw;
  const long long blocks = (items + 31) / 32;
  hipLaunchKernelGGL(temporal_gate16_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(aligned),
                     reinterpret_cast<const bf16_t*>(emb), reinterpret_cast<const bf16_t*>(emb_ref), b, t, hw, c, reinterpret_cast<bf16_t*>(af));
  return check_launch("temporal_gate_bf16");
}

extern "C" int gpemsr_frame_mix_lrelu_bf16(const void* af, int64_t pixels, int t, int c, const float* m, const float* bias, void* out, void* stream) {
  GP_REQUIRE(af && m && bias && out && t <= 8 && c % 8 == 0 && A16(af) && A16(out), "frame_mix_bf16: bad args");
  if (t == 5)
    hipLaunchKernelGGL(frame_mix16_kernel<5>, dim3(grid16(pixels * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(af), (long long)pixels, t, c, m, bias,
                       reinterpret_cast<bf16_t*>(out));
  else
    hipLaunchKernelGGL(frame_mix16_kernel<0>, dim3(grid16(pixels * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(af), (long long)pixels, t, c, m, bias,
                       reinterpret_cast<bf16_t*>(out));
  return check_launch("frame_mix_bf16");
}

extern "C" int gpemsr_threeda_combine_bf16(const void* feat, const void* attn, const void* attn_add, const void* f2, const void* f3, int64_t count,
                                           void* out, void* stream) {
  GP_REQUIRE(feat && attn && attn_add && f2 && f3 && out && count % 8 == 0, "threeda_combine_bf16: bad args");
  hipLaunchKernelGGL(threeda_combine16_kernel, dim3(grid16(count / 8)), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(feat), reinterpret_cast<const bf16_t*>(attn),
                     reinterpret_cast<const bf16_t*>(attn_add), reinterpret_cast<const bf16_t*>(f2), reinterpret_cast<const bf16_t*>(f3), (long long)(count / 8),
                     reinterpret_cast<bf16_t*>(out));
  return check_launch("threeda_combine_bf16");
}

extern "C" int gpemsr_copy_channels_bf16(const void* src, int src_ld, void* dst, int dst_ld, int64_t pixels, int c, void* stream) {
  GP_REQUIRE(src && dst && pixels > 0 && c > 0 && c % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && A16(src) && A16(dst), "copy_channels_bf16: bad args");
  hipLaunchKernelGGL(copy_channels16_kernel, dim3(grid16(pixels * (c / 8))), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(src), src_ld,
                     reinterpret_cast<bf16_t*>(dst), dst_ld, (long long)pixels, c / 8);
  return check_launch("copy_channels_bf16");
}

extern "C" int gpemsr_copy_channels_f32_bf16(const float* src, int src_ld, void* dst, int dst_ld, int64_t pixels, int c, void* stream) {
  GP_REQUIRE(src && dst && pixels > 0 && c > 0, "copy_channels_f32_bf16: bad args");
  hipLaunchKernelGGL(copy_channels_f32_bf16_kernel, dim3(grid16(pixels * c)), dim3(256), 0, ST(stream), src, src_ld, reinterpret_cast<bf16_t*>(dst), dst_ld, (long long)pixels, c);
  return check_launch("copy_channels_f32_bf16");
}

extern "C" int gpemsr_conv2d_stem1_bf16(const float* x, int n, int h, int w, const float* weight, const float* bias, int cout, int act,
                                        void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && weight && out && n > 0 && h > 0 && w > 0, "conv2d_stem1_bf16: bad args");
  GP_REQUIRE(cout % 8 == 0 && cout <= 512 && out_ld % 8 == 0 && A16(out), "conv2d_stem1_bf16: cout/out alignment");
  const long long total = (long long)n * h * ((w + 3) / 4) * (cout / 8);
  hipLaunchKernelGGL(conv_stem1_16_kernel, dim3(grid16(total) < 65536 ? grid16(total) : 65536), dim3(256), (size_t)10 * cout * sizeof(float), ST(stream),
                     x, n, h, w, weight, bias, cout, 8, act, reinterpret_cast<bf16_t*>(out), out_ld);
  return check_launch("conv_stem1_16_kernel");
}

extern "C" int gpemsr_conv2d_direct_bf16(const void* x, int x_f32, int n, int h, int w, int ld, int cin, const float* weight, const float* bias,
                                         int cout, int ksize, int stride, int act, const float* residual, int res_ld, void* out, int out_f32,
                                         int out_ld, void* stream) {
  GP_REQUIRE(x && weight && out, "conv2d_direct_bf16: null pointer");
  GP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && cout <= 16, "conv2d_direct_bf16: bad geometry (cout<=16)");
  GP_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1) && (stride == 1 || stride == 2 || stride == 4), "conv2d_direct_bf16: ksize/stride");
  const int pad = ksize / 2, cin_pad = (cin + 7) / 8 * 8;
  const int oh = (h + 2 * pad - ksize) / stride + 1, ow = (w + 2 * pad - ksize) / stride + 1;
  hipStream_t st = ST(stream);
  if (!x_f32 && out_f32 && cin == 64 && cout == 1 && ksize == 3 && stride == 1 && w >= 8 && ld % 8 == 0 && A16(x)) {
    constexpr int RUN = 8;
    const long long nrun = (long long)n * h * ((w + RUN - 1) / RUN);
    const long long blocks = (nrun + 31) / 32;
    hipLaunchKernelGGL(conv_c64_cout1_16_kernel<RUN>, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(x),
                       n, h, w, ld, weight, cin_pad, bias, act, residual, res_ld, reinterpret_cast<float*>(out), out_ld);
    return check_launch("conv_c64_cout1_16_kernel");
  }
  GP_REQUIRE(!residual, "conv2d_direct_bf16: residual only in the 64 -> 1 form");
  if (ksize == 3 && cout <= 16 && ((x_f32 && cin == 2) || (!x_f32 && cin == 16 && ld % 8 == 0 && A16(x)))) {
    const unsigned g2 = grid16((long long)n * oh * ow);
    const int vec = (!out_f32 && cout == 16 && out_ld % 8 == 0 && A16(out)) ? 1 : 0;
#define GP_S(TI, CI, TO) hipLaunchKernelGGL((conv_small16_kernel<TI, CI, TO>), dim3(g2), dim3(256), 0, st, reinterpret_cast<const TI*>(x), n, h, w, ld, cin_pad, weight, bias, \
                                            cout, stride, oh, ow, act, reinterpret_cast<TO*>(out), out_ld, vec)
    if (x_f32) { if (out_f32) GP_S(float, 2, float); else GP_S(float, 2, bf16_t); }
    else { if (out_f32) GP_S(bf16_t, 16, float); else GP_S(bf16_t, 16, bf16_t); }
#undef GP_S
    return check_launch("conv_small16_kernel");
  }
  const unsigned grid = grid16((long long)n * oh * ow * cout);
#define GP_D(TI, TO) hipLaunchKernelGGL((conv_direct16_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const TI*>(x), n, h, w, ld, cin, cin_pad, \
                                        weight, bias, cout, ksize, stride, oh, ow, act, reinterpret_cast<TO*>(out), out_ld)
  if (x_f32) { if (out_f32) GP_D(float, float); else GP_D(float, bf16_t); }
  else { if (out_f32) GP_D(bf16_t, float); else GP_D(bf16_t, bf16_t); }
#undef GP_D
  return check_launch("conv_direct16_kernel");
}
