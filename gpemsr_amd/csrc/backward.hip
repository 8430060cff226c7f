// Stage-3 training step, backward side (train_stage3.py:343-366 -> loss_total.backward(); optimizer_G.step()):
// the elementwise / gather / reduction gradients of every op on the trainable part of the path.  The matrix work of
// the backward reuses the forward machinery: data gradients of convolutions are convolutions (gpemsr_conv2d with
// re-packed weights), weight gradients are in wgrad.hip.
//
// Convention: a gradient output marked "+=" is ACCUMULATED into a zero-initialised buffer (a tensor with several
// consumers receives one contribution per consumer).  Everything here is deterministic: fixed summation order, and the one scatter
// (the deformable sampling's input gradient) accumulates in 64-bit fixed point with integer atomics (gpemsr_dcn_columns_bwd_det;
// gpemsr_dcn_columns_bwd is the older float-atomic form, kept for A/B).
#include "common.h"

namespace gpemsr {

inline unsigned bgrid(long long total) {
  const long long b = (total + 255) / 256;
  return (unsigned)(b < 32768 ? (b < 1 ? 1 : b) : 32768);
}

__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  switch (act) {
    case GPEMSR_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case GPEMSR_ACT_LRELU: return y > 0.f ? 1.f : 0.1f;
    case GPEMSR_ACT_SIGMOID: return y * (1.f - y);
    case GPEMSR_ACT_LRELU_SIGMOID: return y * (1.f - y) * (y > 0.5f ? 1.f : 0.1f);
    default: return 1.f;
  }
}

// dz[n][h][w][c] = dy * act'(y); with pixel_shuffle dy / y are [n][2h][2w][c/4] and
// dz[.., y, x, 4*cc + 2*i + j] = (dy * act')[.., 2y+i, 2x+j, cc]   (nn.PixelShuffle(2) backward).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, int dy_ld, const float* y, int y_ld, int n, int h, int w,
                                                      int c, int act, int ps, float* dz, int dz_ld) {
  const long long total = (long long)n * h * w * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    const long long p = e / c;
    long long sp; int sc;
    if (ps) {
      const int xq = (int)(p % w), yq = (int)((p / w) % h);
      const long long img = p / ((long long)w * h);
      const int cc = ch >> 2, i = (ch >> 1) & 1, j = ch & 1;
      sp = (img * (2 * h) + 2 * yq + i) * (2 * w) + 2 * xq + j;
      sc = cc;
    } else { sp = p; sc = ch; }
    const float g = dy[sp * dy_ld + sc];
    const float d = act == GPEMSR_ACT_NONE ? 1.f : act_grad_from_output(y[sp * y_ld + sc], act);
    dz[p * dz_ld + ch] = g * d;
  }
}

// column sums of dz[pixels][c] (bias gradient).  Stage 1: block b sums pixels b, b+G, ... for every channel.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* dz, long long pixels, int c, int ld, float* ws) {
  // thread t owns channel t % cw of pixel lane t / cw (cw = min(c, 256)); lanes are folded in a fixed order through LDS
  const int cw = c >= 256 ? 256 : c;
  const int L = 256 / cw;
  const int chl = threadIdx.x % cw, pl = threadIdx.x / cw;
  __shared__ float red[256];
  for (int c0 = 0; c0 < c; c0 += cw) {
    const int ch = c0 + chl;
    float s = 0.f;
    if (pl < L && ch < c)
      for (long long p = (long long)blockIdx.x * L + pl; p < pixels; p += (long long)gridDim.x * L) s += dz[p * ld + ch];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && ch < c) {
      for (int l = 1; l < L; ++l) s += red[l * cw + chl];
      ws[(long long)blockIdx.x * c + ch] = s;
    }
    __syncthreads();
  }
}
// one wave per channel: lane l sums blocks l, l+64, ..., then a fixed shuffle tree
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* ws, int blocks, int c, float* db) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += ws[(long long)b * c + ch];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) db[ch] += s;
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* src, int src_ld, float* dst, int dst_ld, long long pixels, int c, float alpha) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    dst[p * dst_ld + ch] += alpha * src[p * src_ld + ch];
  }
}

__global__ __launch_bounds__(256) void mul_pix_kernel(const float* x, int x_ld, const float* m, long long pixels, int c, float* out, int out_ld) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    out[p * out_ld + ch] = x[p * x_ld + ch] * m[p];
  }
}

// out = x * m[pixel]: dx += dy * m; dm[pixel] += sum_c dy * x.  One 16-lane group per pixel.
__global__ __launch_bounds__(256) void mul_pix_bwd_kernel(const float* dy, int dy_ld, const float* x, int x_ld, const float* m,
                                                          long long pixels, int c, float* dx, int dx_ld, float* dm) {
  const int sub = threadIdx.x & 15;
  const long long per_iter = (long long)gridDim.x * 16;
  const long long niter = (pixels + per_iter - 1) / per_iter;
  for (long long it = 0; it < niter; ++it) {
    const long long p = it * per_iter + (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = p < pixels;
    float s = 0.f;
    if (valid) {
      const float mv = m[p];
      for (int ch = sub; ch < c; ch += 16) {
        const float g = dy[p * dy_ld + ch];
        s += g * x[p * x_ld + ch];
        if (dx) dx[p * dx_ld + ch] += g * mv;
      }
    }
#pragma unroll
    for (int k = 8; k >= 1; k >>= 1) s += __shfl_xor(s, k);
    if (valid && sub == 0 && dm) dm[p] += s;
  }
}

__device__ __forceinline__ void bsrc_index(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float s = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - i0;
}

// F.interpolate(bilinear) backward in gather form: every input element sums the output pixels whose two source taps
// include it, re-evaluating the forward's source-index formula (bit-identical weights, fixed order).
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* dy, int dy_ld, int n, int h, int w, int c, int oh, int ow,
                                                           int align, float sh, float sw, float mul, float* dx, int dx_ld) {
  const long long total = (long long)n * h * w * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ix = (int)(p % w); p /= w;
    const int iy = (int)(p % h);
    const int img = (int)(p / h);
    // conservative candidate ranges (exact membership is re-checked through bsrc_index)
    int ylo, yhi, xlo, xhi;
    if (align) {
      ylo = sh > 0.f ? (int)floorf((iy - 1) / sh) - 1 : 0; yhi = sh > 0.f ? (int)ceilf((iy + 1) / sh) + 1 : oh - 1;
      xlo = sw > 0.f ? (int)floorf((ix - 1) / sw) - 1 : 0; xhi = sw > 0.f ? (int)ceilf((ix + 1) / sw) + 1 : ow - 1;
    } else {
      ylo = (int)floorf((iy - 0.5f) / sh - 0.5f) - 1; yhi = (int)ceilf((iy + 1.5f) / sh - 0.5f) + 1;
      xlo = (int)floorf((ix - 0.5f) / sw - 0.5f) - 1; xhi = (int)ceilf((ix + 1.5f) / sw - 0.5f) + 1;
    }
    if (iy == 0) ylo = 0;
    if (ix == 0) xlo = 0;
    if (iy == h - 1) yhi = oh - 1;
    if (ix == w - 1) xhi = ow - 1;
    ylo = max(ylo, 0); xlo = max(xlo, 0); yhi = min(yhi, oh - 1); xhi = min(xhi, ow - 1);
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float ly;
      bsrc_index(oy, sh, align, h, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx;
        bsrc_index(ox, sw, align, w, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx != 0.f) row += wx * dy[(((long long)img * oh + oy) * ow + ox) * dy_ld + ch];
      }
      acc += wy * row;
    }
    dx[(((long long)img * h + iy) * w + ix) * dx_ld + ch] += acc * mul;
  }
}

// Modulated deformable sampling backward (torchvision deform_conv2d backward for input / offset / mask, restated):
// thread per (pixel, tap, group) as in the forward.  dom (+=) has one writer per element; dx (+=) is scattered with
// float atomics (up to 4 corners x 8 channels per thread).
__global__ __launch_bounds__(256) void dcn_columns_bwd_kernel(const float* x, int n, int h, int w, int c, int ld, const float* om,
                                                              int om_ld, int groups, const float* dcol, float* dx, int dx_ld,
                                                              float* dom, int dom_ld) {
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
    const float dy = o[g * 2 * K + 2 * k], dxo = o[g * 2 * K + 2 * k + 1];
    const float ml = o[2 * groups * K + g * K + k];
    const float m = 1.f / (1.f + expf(-ml));
    const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dxo;
    const float* dc = dcol + pix * (long long)(K * c) + k * c + g * cg;
    const float4 d0 = *reinterpret_cast<const float4*>(dc), d1 = *reinterpret_cast<const float4*>(dc + 4);
    const float dcv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    float g_m = 0.f, g_py = 0.f, g_px = 0.f;
    if (py > -1.f && py < (float)h && px > -1.f && px < (float)w) {
      const int y0 = (int)floorf(py), x0 = (int)floorf(px);
      const float ly = py - y0, lx = px - x0;
      const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
      const float wdy[4] = {-(1.f - lx), -lx, (1.f - lx), lx};          // d wts / d py
      const float wdx[4] = {-(1.f - ly), (1.f - ly), -ly, ly};          // d wts / d px
      const int ys[4] = {y0, y0, y0 + 1, y0 + 1}, xs[4] = {x0, x0 + 1, x0, x0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (ys[q] >= 0 && ys[q] <= h - 1 && xs[q] >= 0 && xs[q] <= w - 1) {
          const long long sp = ((long long)img * h + ys[q]) * w + xs[q];
          const float* xp = x + sp * ld + g * cg;
          const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
          const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          float dot = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) dot += dcv[r] * xv[r];
          g_m += wts[q] * dot;
          g_py += wdy[q] * dot;
          g_px += wdx[q] * dot;
          if (dx) {
            float* dp = dx + sp * dx_ld + g * cg;
            const float wm = wts[q] * m;
#pragma unroll
            for (int r = 0; r < 8; ++r) unsafeAtomicAdd(dp + r, wm * dcv[r]);      // hardware global_atomic_add_f32 (no CAS loop)
          }
        }
      }
    }
    if (dom) {
      float* dq = dom + pix * dom_ld;
      dq[g * 2 * K + 2 * k] += g_py * m;
      dq[g * 2 * K + 2 * k + 1] += g_px * m;
      dq[2 * groups * K + g * K + k] += g_m * m * (1.f - m);
    }
  }
}


// Tiled flavour of the same backward: one workgroup owns an 8x8-pixel tile (all taps and groups) and accumulates the
// input-gradient scatter in an LDS window of the tile +- DCN_M pixels with LDS float atomics (ds_add_f32); only what
// falls outside the window, and one flush of the window's non-zero entries, go to global atomics -- about an order of
// magnitude fewer L2 atomics than one per (corner, channel).  dom as above.
#define DCN_T 8
#define DCN_M 4
#define DCN_WW (DCN_T + 2 * DCN_M)
__global__ __launch_bounds__(256) void dcn_columns_bwd_tiled_kernel(const float* x, int n, int h, int w, int c, int ld, const float* om,
                                                                    int om_ld, int groups, const float* dcol, float* dx, int dx_ld,
                                                                    float* dom, int dom_ld, int tiles_x, int tiles_y) {
  extern __shared__ float win[];                           // [DCN_WW][DCN_WW][c]
  const int cg = c / groups, K = 9;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, img = blockIdx.x / (tiles_x * tiles_y);
  const int y0t = ty * DCN_T, x0t = tx * DCN_T;
  const int wy0 = y0t - DCN_M, wx0 = x0t - DCN_M;
  for (int e = threadIdx.x; e < DCN_WW * DCN_WW * c; e += 256) win[e] = 0.f;
  __syncthreads();
  const int items = DCN_T * DCN_T * K * groups;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int g = it % groups;
    const int k = (it / groups) % K;
    const int lp = it / (groups * K);
    const int yq = y0t + lp / DCN_T, xq = x0t + lp % DCN_T;
    if (yq >= h || xq >= w) continue;
    const long long pix = ((long long)img * h + yq) * w + xq;
    const float* o = om + pix * om_ld;
    const float dy = o[g * 2 * K + 2 * k], dxo = o[g * 2 * K + 2 * k + 1];
    const float ml = o[2 * groups * K + g * K + k];
    const float m = 1.f / (1.f + expf(-ml));
    const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dxo;
    const float* dc = dcol + pix * (long long)(K * c) + k * c + g * cg;
    const float4 d0 = *reinterpret_cast<const float4*>(dc), d1 = *reinterpret_cast<const float4*>(dc + 4);
    const float dcv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    float g_m = 0.f, g_py = 0.f, g_px = 0.f;
    if (py > -1.f && py < (float)h && px > -1.f && px < (float)w) {
      const int y0 = (int)floorf(py), x0 = (int)floorf(px);
      const float ly = py - y0, lx = px - x0;
      const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
      const float wdy[4] = {-(1.f - lx), -lx, (1.f - lx), lx};
      const float wdx[4] = {-(1.f - ly), (1.f - ly), -ly, ly};
      const int ys[4] = {y0, y0, y0 + 1, y0 + 1}, xs[4] = {x0, x0 + 1, x0, x0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (ys[q] >= 0 && ys[q] <= h - 1 && xs[q] >= 0 && xs[q] <= w - 1) {
          const long long sp = ((long long)img * h + ys[q]) * w + xs[q];
          const float* xp = x + sp * ld + g * cg;
          const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
          const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          float dot = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) dot += dcv[r] * xv[r];
          g_m += wts[q] * dot; g_py += wdy[q] * dot; g_px += wdx[q] * dot;
          if (dx) {
            const float wm = wts[q] * m;
            const int wy = ys[q] - wy0, wx = xs[q] - wx0;
            if (wy >= 0 && wy < DCN_WW && wx >= 0 && wx < DCN_WW) {
              float* lp2 = win + (wy * DCN_WW + wx) * c + g * cg;
#pragma unroll
              for (int r = 0; r < 8; ++r) unsafeAtomicAdd(lp2 + r, wm * dcv[r]);
            } else {
              float* dp = dx + sp * dx_ld + g * cg;
#pragma unroll
              for (int r = 0; r < 8; ++r) unsafeAtomicAdd(dp + r, wm * dcv[r]);
            }
          }
        }
      }
    }
    if (dom) {
      float* dq = dom + pix * dom_ld;
      dq[g * 2 * K + 2 * k] += g_py * m;
      dq[g * 2 * K + 2 * k + 1] += g_px * m;
      dq[2 * groups * K + g * K + k] += g_m * m * (1.f - m);
    }
  }
  if (!dx) return;
  __syncthreads();
  for (int e = threadIdx.x; e < DCN_WW * DCN_WW * c; e += 256) {
    const float v = win[e];
    if (v == 0.f) continue;
    const int ch = e % c, wp = e / c;
    const int yy = wy0 + wp / DCN_WW, xx = wx0 + wp % DCN_WW;
    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
    unsafeAtomicAdd(dx + (((long long)img * h + yy) * w + xx) * dx_ld + ch, v);
  }
}

// BIT-STABLE flavour of the same backward (deterministic reductions).  Float addition is not associative, so a scatter
// through float atomics depends on the order in which the hardware retires them.  Here every contribution wts * m * dcol is turned into a
// 64-bit FIXED-POINT integer -- scale 2^(38 - e), e = exponent of max |dcol| (a device scalar from a deterministic max reduction; |contribution|
// <= max |dcol|, so 2^24 of them still fit) -- and accumulated with INTEGER atomics (ds_add_u64 in the tile's LDS window, global_atomic_add_x2
// outside it and for the window's flush): integer addition is associative, the sum is exact whatever the order; resolution 2^-38 of the largest
// column gradient, i.e. far below fp32's own rounding of the result.  gpemsr_dcn_fix_finish then adds the converted sums into dx in a fixed
// order (one writer per element).  dom as above (one writer per element).
__device__ __forceinline__ float dcn_fix_scale(const float* absmax) {
  const float a = *absmax;
  if (!(a > 0.f) || !isfinite(a)) return 1.f;
  int e; frexpf(a, &e);                                    // a = f * 2^e, f in [0.5, 1)
  return ldexpf(1.f, 38 - e);
}
template <bool WINDOW>
__global__ __launch_bounds__(256) void dcn_columns_bwd_fix_kernel(const float* x, int n, int h, int w, int c, int ld, const float* om,
                                                                  int om_ld, int groups, const float* dcol, const float* absmax,
                                                                  unsigned long long* dx_fix, float* dom, int dom_ld, int tiles_x, int tiles_y) {
  extern __shared__ unsigned long long winq[];            // WINDOW: [DCN_WW][DCN_WW][c]
  const int cg = c / groups, K = 9;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, img = blockIdx.x / (tiles_x * tiles_y);
  const int y0t = ty * DCN_T, x0t = tx * DCN_T;
  const int wy0 = y0t - DCN_M, wx0 = x0t - DCN_M;
  const float scale = dcn_fix_scale(absmax);
  if (WINDOW) {
    for (int e = threadIdx.x; e < DCN_WW * DCN_WW * c; e += 256) winq[e] = 0ull;
    __syncthreads();
  }
  const int items = DCN_T * DCN_T * K * groups;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int g = it % groups;
    const int k = (it / groups) % K;
    const int lp = it / (groups * K);
    const int yq = y0t + lp / DCN_T, xq = x0t + lp % DCN_T;
    if (yq >= h || xq >= w) continue;
    const long long pix = ((long long)img * h + yq) * w + xq;
    const float* o = om + pix * om_ld;
    const float dy = o[g * 2 * K + 2 * k], dxo = o[g * 2 * K + 2 * k + 1];
    const float ml = o[2 * groups * K + g * K + k];
    const float m = 1.f / (1.f + expf(-ml));
    const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dxo;
    const float* dc = dcol + pix * (long long)(K * c) + k * c + g * cg;
    const float4 d0 = *reinterpret_cast<const float4*>(dc), d1 = *reinterpret_cast<const float4*>(dc + 4);
    const float dcv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    float g_m = 0.f, g_py = 0.f, g_px = 0.f;
    if (py > -1.f && py < (float)h && px > -1.f && px < (float)w) {
      const int y0 = (int)floorf(py), x0 = (int)floorf(px);
      const float ly = py - y0, lx = px - x0;
      const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
      const float wdy[4] = {-(1.f - lx), -lx, (1.f - lx), lx};
      const float wdx[4] = {-(1.f - ly), (1.f - ly), -ly, ly};
      const int ys[4] = {y0, y0, y0 + 1, y0 + 1}, xs[4] = {x0, x0 + 1, x0, x0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (ys[q] >= 0 && ys[q] <= h - 1 && xs[q] >= 0 && xs[q] <= w - 1) {
          const long long sp = ((long long)img * h + ys[q]) * w + xs[q];
          const float* xp = x + sp * ld + g * cg;
          const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
          const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          float dot = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) dot += dcv[r] * xv[r];
          g_m += wts[q] * dot; g_py += wdy[q] * dot; g_px += wdx[q] * dot;
          if (dx_fix) {
            const float wm = wts[q] * m * scale;             // (a power of two: the product with it is exact)
            const int wy = ys[q] - wy0, wx = xs[q] - wx0;
            unsigned long long* tp = (WINDOW && wy >= 0 && wy < DCN_WW && wx >= 0 && wx < DCN_WW) ? winq + (wy * DCN_WW + wx) * c + g * cg
                                                                                                    : dx_fix + sp * c + g * cg;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const long long qv = __float2ll_rn(wm * dcv[r]);
              if (qv != 0) atomicAdd(tp + r, (unsigned long long)qv);      // two's complement: the wrapped unsigned sum is the signed sum
            }
          }
        }
      }
    }
    if (dom) {
      float* dq = dom + pix * dom_ld;
      dq[g * 2 * K + 2 * k] += g_py * m;
      dq[g * 2 * K + 2 * k + 1] += g_px * m;
      dq[2 * groups * K + g * K + k] += g_m * m * (1.f - m);
    }
  }
  if (!WINDOW || !dx_fix) return;
  __syncthreads();
  for (int e = threadIdx.x; e < DCN_WW * DCN_WW * c; e += 256) {
    const unsigned long long v = winq[e];
    if (v == 0ull) continue;
    const int ch = e % c, wp = e / c;
    const int yy = wy0 + wp / DCN_WW, xx = wx0 + wp % DCN_WW;
    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
    atomicAdd(dx_fix + (((long long)img * h + yy) * w + xx) * c + ch, v);
  }
}
// dx[pixel][ch] += fixed-point sum / scale; one thread per element, 64-bit -> double -> float (one rounding)
__global__ __launch_bounds__(256) void dcn_fix_finish_kernel(const unsigned long long* dx_fix, const float* absmax, long long pixels, int c,
                                                             float* dx, int dx_ld) {
  const double inv = 1.0 / (double)dcn_fix_scale(absmax);
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long q = (long long)dx_fix[e];
    if (q != 0) dx[(e / c) * dx_ld + (e % c)] += (float)((double)q * inv);
  }
}

// ThreeDA temporal gate backward; one 16-lane group per (b, pixel), looping over the t frames so that the sum into
// d_emb_ref has a fixed order.  c == 64.
__global__ __launch_bounds__(256) void temporal_gate_bwd_kernel(const float* aligned, const float* emb, const float* emb_ref,
                                                                const float* daf, int b, int t, int hw, int c,
                                                                float* d_aligned, float* d_emb, float* d_emb_ref) {
  const int sub = threadIdx.x & 15;
  const long long items = (long long)b * hw;
  const long long per_iter = (long long)gridDim.x * 16;
  const long long niter = (items + per_iter - 1) / per_iter;
  for (long long it = 0; it < niter; ++it) {
    const long long item = it * per_iter + (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = item < items;
    const long long ii = valid ? item : 0;
    const int p = (int)(ii % hw);
    const int bi = (int)(ii / hw);
    const float4 e0 = *reinterpret_cast<const float4*>(emb_ref + ii * c + 4 * sub);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ti = 0; ti < t; ++ti) {
      const long long fi = ((long long)bi * t + ti) * hw + p;
      const float4 e1 = *reinterpret_cast<const float4*>(emb + fi * c + 4 * sub);
      const float4 v = *reinterpret_cast<const float4*>(aligned + fi * c + 4 * sub);
      const float4 dg = *reinterpret_cast<const float4*>(daf + ii * ((long long)t * c) + ti * c + 4 * sub);
      float d = e1.x * e0.x + e1.y * e0.y + e1.z * e0.z + e1.w * e0.w;
      float s = dg.x * v.x + dg.y * v.y + dg.z * v.z + dg.w * v.w;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) { d += __shfl_xor(d, m); s += __shfl_xor(s, m); }
      const float g = 1.f / (1.f + expf(-d));
      const float dd = s * g * (1.f - g);
      if (valid) {
        float4* da = reinterpret_cast<float4*>(d_aligned + fi * c + 4 * sub);
        float4 a = *da; a.x += dg.x * g; a.y += dg.y * g; a.z += dg.z * g; a.w += dg.w * g; *da = a;
        float4* de = reinterpret_cast<float4*>(d_emb + fi * c + 4 * sub);
        float4 q = *de; q.x += dd * e0.x; q.y += dd * e0.y; q.z += dd * e0.z; q.w += dd * e0.w; *de = q;
        acc.x += dd * e1.x; acc.y += dd * e1.y; acc.z += dd * e1.z; acc.w += dd * e1.w;
      }
    }
    if (valid) {
      float4* dr = reinterpret_cast<float4*>(d_emb_ref + ii * c + 4 * sub);
      float4 r = *dr; r.x += acc.x; r.y += acc.y; r.z += acc.z; r.w += acc.w; *dr = r;
    }
  }
}

// Conv3d(t,t,1)+LeakyReLU backward.  out = lrelu(bias_i + sum_k M[i][k] af[k]).  d_af += M^T dz; per-block partial sums of
// dM (t*t) and dbias (t) go to ws[block][t*t + t]; frame_mix_bwd_final adds them up in block order.
// TT > 0: frame count at compile time (5): pm[], in[], dz[] stay in registers (dynamic indexing put them in scratch, 176 B/lane)
template <int TT>
__global__ __launch_bounds__(256) void frame_mix_bwd_kernel(const float* af, const float* out, const float* dout, long long pixels,
                                                            int t_rt, int c, const float* m, float* d_af, float* ws) {
  const int t = TT > 0 ? TT : t_rt;
  const int c4 = c >> 2;
  const long long total = pixels * c4;
  float pm[30];                          // t <= 5: 25 + 5
  const int nred = t * t + t;
#pragma unroll
  for (int i = 0; i < 30; ++i) pm[i] = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c4);
    const long long p = e / c4;
    const long long base = p * ((long long)t * c) + 4 * j;
    float4 in[5], dz[5];
#pragma unroll
    for (int k = 0; k < t; ++k) {
      in[k] = *reinterpret_cast<const float4*>(af + base + k * c);
      const float4 o = *reinterpret_cast<const float4*>(out + base + k * c);
      float4 g = *reinterpret_cast<const float4*>(dout + base + k * c);
      g.x *= o.x > 0.f ? 1.f : 0.1f; g.y *= o.y > 0.f ? 1.f : 0.1f; g.z *= o.z > 0.f ? 1.f : 0.1f; g.w *= o.w > 0.f ? 1.f : 0.1f;
      dz[k] = g;
    }
#pragma unroll
    for (int k = 0; k < t; ++k) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < t; ++i) {
        const float wv = m[i * t + k];
        s.x = fmaf(wv, dz[i].x, s.x); s.y = fmaf(wv, dz[i].y, s.y); s.z = fmaf(wv, dz[i].z, s.z); s.w = fmaf(wv, dz[i].w, s.w);
      }
      float4* dp = reinterpret_cast<float4*>(d_af + base + k * c);
      float4 q = *dp; q.x += s.x; q.y += s.y; q.z += s.z; q.w += s.w; *dp = q;
    }
#pragma unroll
    for (int i = 0; i < t; ++i) {
#pragma unroll
      for (int k = 0; k < t; ++k)
        pm[i * t + k] += dz[i].x * in[k].x + dz[i].y * in[k].y + dz[i].z * in[k].z + dz[i].w * in[k].w;
      pm[t * t + i] += (dz[i].x + dz[i].y) + (dz[i].z + dz[i].w);
    }
  }
  __shared__ float red[4][30];
#pragma unroll
  for (int i = 0; i < (TT > 0 ? TT * TT + TT : 30); ++i) {
    if (i >= nred) break;
    float v = pm[i];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < nred) ws[(long long)blockIdx.x * nred + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void frame_mix_bwd_final_kernel(const float* ws, int blocks, int t, float* dm, float* dbias) {
  const int i = threadIdx.x, nred = t * t + t;
  if (i >= nred) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += ws[(long long)b * nred + i];
  if (i < t * t) dm[i] += s; else dbias[i - t * t] += s;
}

// MaxPool2d(3,2,1) | AvgPool2d(3,2,1) backward (dy = [max grads | avg grads]); gather form: every input element visits
// the <= 4 windows that contain it and re-derives each window's arg-max (first maximum in scan order, as ATen).
__global__ __launch_bounds__(256) void pool3s2_bwd_kernel(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld,
                                                          float* dx, int dx_ld) {
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  const long long total = (long long)n * h * w * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ix = (int)(p % w); p /= w;
    const int iy = (int)(p % h);
    const int img = (int)(p / h);
    float acc = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy >= oh) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox >= ow) continue;
        // window rows 2oy-1..2oy+1 contains iy by construction
        float mx = -INFINITY; int ay = -1, ax = -1;
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = 2 * oy - 1 + ky;
          if (yy < 0 || yy >= h) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = 2 * ox - 1 + kx;
            if (xx < 0 || xx >= w) continue;
            const float v = x[(((long long)img * h + yy) * w + xx) * ld + ch];
            if (v > mx) { mx = v; ay = yy; ax = xx; }
          }
        }
        const float* g = dy + (((long long)img * oh + oy) * ow + ox) * dy_ld;
        if (ay == iy && ax == ix) acc += g[ch];
        acc += g[c + ch] * (1.f / 9.f);
      }
    }
    dx[(((long long)img * h + iy) * w + ix) * dx_ld + ch] += acc;
  }
}

__global__ __launch_bounds__(256) void threeda_combine_bwd_kernel(const float* feat, const float* attn, const float* dout, long long count,
                                                                  float* dfeat, float* dattn, float* dadd, float* df2, float* df3) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float g = dout[e];
    const float s = 1.f / (1.f + expf(-attn[e]));
    dfeat[e] += g * 2.f * s;
    dattn[e] += g * feat[e] * 2.f * s * (1.f - s);
    dadd[e] += g; df2[e] += g; df3[e] += g;
  }
}

// nn.MaxPool2d(2,2) backward: dx[first max of the window] += dy
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld,
                                                           float* dx, int dx_ld) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)n * oh * ow * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    const long long b00 = (((long long)img * h + 2 * oy) * w + 2 * ox);
    const long long pos[4] = {b00, b00 + 1, b00 + w, b00 + w + 1};
    float mx = -INFINITY; int a = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float v = x[pos[q] * ld + ch]; if (v > mx) { mx = v; a = q; } }
    dx[pos[a] * dx_ld + ch] += dy[(((long long)img * oh + oy) * ow + ox) * dy_ld + ch];
  }
}

// dtarget image i += sum over j with idx[j] == i of dsrc image j  (backward of gather_images / copy_images)
__global__ __launch_bounds__(256) void scatter_add_images_kernel(const float* dsrc, const int* idx, float* dtarget, int n_src, int n_dst,
                                                                 long long elems4) {
  const long long total = (long long)n_src * elems4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / elems4); const long long k = e % elems4;
    float4 acc = reinterpret_cast<float4*>(dtarget)[e];
    for (int j = 0; j < n_dst; ++j) {
      if (idx[j] != i) continue;
      const float4 v = reinterpret_cast<const float4*>(dsrc)[(long long)j * elems4 + k];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(dtarget)[e] = acc;
  }
}

// torch.nn.L1Loss()(gt, sr) (mean) and its gradient w.r.t. sr: sign(sr - gt) * gscale / count
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* sr, const float* gt, long long count, float gscale, float* dsr, float* ws) {
  float s = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float d = sr[e] - gt[e];
    s += fabsf(d);
    if (dsr) dsr[e] += (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
  }
  __shared__ float red[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// Codebook.forward's loss and gradients (model/codebook.py:20-31), zq = E[idx]:
//   loss = mean((zq.detach() - z)^2) + beta * mean((zq - z.detach())^2)            (one value: (1 + beta) * mean((zq - z)^2))
//   dz += cz * (z - zq),  dE[idx] += ce * (zq - z)  (several tokens share a code: float atomics),  zq_out = zq
// (the decoder input z + (zq - z).detach() has the VALUE zq; its gradient is added to dz by the caller)
__global__ __launch_bounds__(256) void vq_loss_partial_kernel(const float* z, int z_ld, const float* E, const int* idx, long long rows, int C,
                                                              float cz, float ce, float* dz, int dz_ld, float* dE, float* zq, int zq_ld, float* ws) {
  float s = 0.f;
  const long long count = rows * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const long long row = e / C; const int c = (int)(e - row * C);
    const int k = idx[row];
    const float q = E[(long long)k * C + c];
    const float d = q - z[row * z_ld + c];
    s = fmaf(d, d, s);
    if (dz) dz[row * dz_ld + c] -= cz * d;
    if (dE) atomicAdd(dE + (long long)k * C + c, ce * d);
    zq[row * zq_ld + c] = q;
  }
  __shared__ float red[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sum_scale_kernel(const float* ws, int blocks, float scale, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += (double)ws[b];
    out[0] = (float)(s * scale);
  }
}

// torch.optim.Adam step (no amsgrad): train_stage3.py:163,365
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long count, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    float gr = g[e];
    const float pv = p[e];
    if (wd != 0.f) gr += wd * pv;
    const float mm = b1 * m[e] + (1.f - b1) * gr;
    const float vv = b2 * v[e] + (1.f - b2) * gr * gr;
    m[e] = mm; v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] = pv - (lr / bc1) * (mm / denom);
  }
}

// dst[n][cols][rows] = src[n][rows][cols]  (per-image transpose through a 32x33 LDS tile)
__global__ __launch_bounds__(256) void transpose_kernel(const float* src, float* dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const long long img = blockIdx.z;
  const float* s = src + img * (long long)rows * cols;
  float* d = dst + img * (long long)rows * cols;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = s[(long long)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) d[(long long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

}  // namespace gpemsr

using namespace gpemsr;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int gpemsr_act_bwd(const float* dy, int dy_ld, const float* y, int y_ld, int n, int h, int w, int c, int act,
                              int pixel_shuffle, float* dz, int dz_ld, void* stream) {
  GP_REQUIRE(dy && dz && (y || act == GPEMSR_ACT_NONE) && n > 0 && h > 0 && w > 0 && c > 0, "act_bwd: bad args");
  GP_REQUIRE(!pixel_shuffle || c % 4 == 0, "act_bwd: pixel_shuffle needs c %% 4 == 0");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(bgrid((long long)n * h * w * c)), dim3(256), 0, ST(stream), dy, dy_ld, y, y_ld, n, h, w, c,
                     act, pixel_shuffle, dz, dz_ld);
  return check_launch("act_bwd");
}

extern "C" int gpemsr_bias_grad(const float* dz, int64_t pixels, int c, int ld, float* ws, int64_t ws_floats, float* db, void* stream) {
  GP_REQUIRE(dz && ws && db && pixels > 0 && c > 0, "bias_grad: bad args");
  int blocks = (int)(pixels < 256 ? pixels : 256);
  if ((int64_t)blocks * c > ws_floats) blocks = (int)(ws_floats / c);
  GP_REQUIRE(blocks >= 1, "bias_grad: workspace too small");
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), 0, ST(stream), dz, (long long)pixels, c, ld, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 3) / 4), dim3(256), 0, ST(stream), ws, blocks, c, db);
  return check_launch("bias_grad");
}

extern "C" int gpemsr_axpy(const float* src, int src_ld, float* dst, int dst_ld, int64_t pixels, int c, float alpha, void* stream) {
  GP_REQUIRE(src && dst && pixels > 0 && c > 0, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(bgrid(pixels * c)), dim3(256), 0, ST(stream), src, src_ld, dst, dst_ld, (long long)pixels, c, alpha);
  return check_launch("axpy");
}

extern "C" int gpemsr_mul_pix(const float* x, int x_ld, const float* m, int64_t pixels, int c, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && m && out && pixels > 0 && c > 0, "mul_pix: bad args");
  hipLaunchKernelGGL(mul_pix_kernel, dim3(bgrid(pixels * c)), dim3(256), 0, ST(stream), x, x_ld, m, (long long)pixels, c, out, out_ld);
  return check_launch("mul_pix");
}

extern "C" int gpemsr_mul_pix_bwd(const float* dy, int dy_ld, const float* x, int x_ld, const float* m, int64_t pixels, int c,
                                  float* dx, int dx_ld, float* dm, void* stream) {
  GP_REQUIRE(dy && x && m && pixels > 0 && c > 0, "mul_pix_bwd: bad args");
  const long long blocks = (pixels + 15) / 16;
  hipLaunchKernelGGL(mul_pix_bwd_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, ST(stream), dy, dy_ld, x, x_ld,
                     m, (long long)pixels, c, dx, dx_ld, dm);
  return check_launch("mul_pix_bwd");
}

extern "C" int gpemsr_bilinear_bwd(const float* dy, int dy_ld, int n, int h, int w, int c, int oh, int ow, int align_corners, float mul,
                                   float* dx, int dx_ld, void* stream) {
  GP_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0, "bilinear_bwd: bad args");
  float sh, sw;
  if (align_corners) { sh = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f; sw = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f; }
  else { sh = (float)((double)h / (double)oh); sw = (float)((double)w / (double)ow); }
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(bgrid((long long)n * h * w * c)), dim3(256), 0, ST(stream), dy, dy_ld, n, h, w, c, oh, ow,
                     align_corners, sh, sw, mul, dx, dx_ld);
  return check_launch("bilinear_bwd");
}

extern "C" int gpemsr_dcn_columns_bwd(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                                      const float* dcol, float* dx, int dx_ld, float* dom, int dom_ld, void* stream) {
  GP_REQUIRE(x && om && dcol, "dcn_columns_bwd: null pointer");
  GP_REQUIRE(groups > 0 && c % groups == 0 && c / groups == 8 && ld % 4 == 0, "dcn_columns_bwd: needs 8 channels per deformable group");
  GP_REQUIRE(om_ld >= 3 * groups * 9 && (!dom || dom_ld >= 3 * groups * 9), "dcn_columns_bwd: om_ld too small");
  const size_t lds = (size_t)DCN_WW * DCN_WW * c * sizeof(float);
  const long long tiles = (long long)n * ((h + DCN_T - 1) / DCN_T) * ((w + DCN_T - 1) / DCN_T);
  if (dx && lds <= 64 * 1024 && tiles < (1ll << 31)) {
    hipLaunchKernelGGL(dcn_columns_bwd_tiled_kernel, dim3((unsigned)tiles), dim3(256), lds, ST(stream), x, n, h, w, c, ld, om, om_ld, groups, dcol,
                       dx, dx_ld, dom, dom_ld, (w + DCN_T - 1) / DCN_T, (h + DCN_T - 1) / DCN_T);
    return check_launch("dcn_columns_bwd");
  }
  hipLaunchKernelGGL(dcn_columns_bwd_kernel, dim3(bgrid((long long)n * h * w * groups * 9)), dim3(256), 0, ST(stream), x, n, h, w, c, ld,
                     om, om_ld, groups, dcol, dx, dx_ld, dom, dom_ld);
  return check_launch("dcn_columns_bwd");
}

extern "C" int gpemsr_dcn_columns_bwd_det(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                                          const float* dcol, const float* dcol_absmax, long long* dx_fix, float* dx, int dx_ld,
                                          float* dom, int dom_ld, void* stream) {
  GP_REQUIRE(x && om && dcol && dcol_absmax, "dcn_columns_bwd_det: null pointer");
  GP_REQUIRE((dx == nullptr) == (dx_fix == nullptr), "dcn_columns_bwd_det: dx and its fixed-point workspace come together");
  GP_REQUIRE(groups > 0 && c % groups == 0 && c / groups == 8 && ld % 4 == 0, "dcn_columns_bwd_det: needs 8 channels per deformable group");
  GP_REQUIRE(om_ld >= 3 * groups * 9 && (!dom || dom_ld >= 3 * groups * 9), "dcn_columns_bwd_det: om_ld too small");
  const int tiles_x = (w + DCN_T - 1) / DCN_T, tiles_y = (h + DCN_T - 1) / DCN_T;
  const long long tiles = (long long)n * tiles_x * tiles_y;
  GP_REQUIRE(tiles > 0 && tiles < (1ll << 31), "dcn_columns_bwd_det: grid too large");
  const size_t lds = (size_t)DCN_WW * DCN_WW * c * sizeof(unsigned long long);
  unsigned long long* fix = reinterpret_cast<unsigned long long*>(dx_fix);
  if (dx && lds <= 144 * 1024) {
    static dev_once_t attr{0};
    if (dev_once_begin(attr)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_columns_bwd_fix_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "dcn_columns_bwd_det: cannot raise the dynamic LDS limit");
      dev_once_done(attr);
    }
    hipLaunchKernelGGL(dcn_columns_bwd_fix_kernel<true>, dim3((unsigned)tiles), dim3(256), lds, ST(stream), x, n, h, w, c, ld, om, om_ld, groups, dcol,
                       dcol_absmax, fix, dom, dom_ld, tiles_x, tiles_y);
  } else {
    hipLaunchKernelGGL(dcn_columns_bwd_fix_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, ST(stream), x, n, h, w, c, ld, om, om_ld, groups, dcol,
                       dcol_absmax, fix, dom, dom_ld, tiles_x, tiles_y);
  }
  if (check_launch("dcn_columns_bwd_det") != GPEMSR_OK) return GPEMSR_ELAUNCH;
  if (dx) {
    hipLaunchKernelGGL(dcn_fix_finish_kernel, dim3(bgrid((long long)n * h * w * c)), dim3(256), 0, ST(stream), fix, dcol_absmax, (long long)n * h * w, c, dx, dx_ld);
    return check_launch("dcn_fix_finish");
  }
  return GPEMSR_OK;
}

extern "C" int gpemsr_temporal_gate_bwd(const float* aligned, const float* emb, const float* emb_ref, const float* daf, int b, int t,
                                        int hw, int c, float* d_aligned, float* d_emb, float* d_emb_ref, void* stream) {
  GP_REQUIRE(aligned && emb && emb_ref && daf && d_aligned && d_emb && d_emb_ref, "temporal_gate_bwd: null pointer");
  GP_REQUIRE(c == 64, "temporal_gate_bwd: c must be 64");
  const long long blocks = ((long long)b * hw + 15) / 16;
  hipLaunchKernelGGL(temporal_gate_bwd_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, ST(stream), aligned, emb,
                     emb_ref, daf, b, t, hw, c, d_aligned, d_emb, d_emb_ref);
  return check_launch("temporal_gate_bwd");
}

extern "C" int gpemsr_frame_mix_lrelu_bwd(const float* af, const float* out, const float* dout, int64_t pixels, int t, int c,
                                          const float* m, float* d_af, float* dm, float* dbias, float* ws, int64_t ws_floats,
                                          void* stream) {
  GP_REQUIRE(af && out && dout && m && d_af && dm && dbias && ws && t <= 5 && c % 4 == 0, "frame_mix_bwd: bad args (t <= 5)");
  const int nred = t * t + t;
  long long blocks = (pixels * (c / 4) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks * nred > ws_floats) blocks = ws_floats / nred;
  GP_REQUIRE(blocks >= 1, "frame_mix_bwd: workspace too small");
  if (t == 5) hipLaunchKernelGGL(frame_mix_bwd_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, ST(stream), af, out, dout, (long long)pixels, t, c, m, d_af, ws);
  else hipLaunchKernelGGL(frame_mix_bwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ST(stream), af, out, dout, (long long)pixels, t, c, m, d_af, ws);
  hipLaunchKernelGGL(frame_mix_bwd_final_kernel, dim3(1), dim3(64), 0, ST(stream), ws, (int)blocks, t, dm, dbias);
  return check_launch("frame_mix_bwd");
}

extern "C" int gpemsr_pool3s2_maxavg_bwd(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld, float* dx,
                                         int dx_ld, void* stream) {
  GP_REQUIRE(x && dy && dx && dy_ld >= 2 * c, "pool3s2_bwd: bad args");
  hipLaunchKernelGGL(pool3s2_bwd_kernel, dim3(bgrid((long long)n * h * w * c)), dim3(256), 0, ST(stream), x, n, h, w, c, ld, dy, dy_ld, dx, dx_ld);
  return check_launch("pool3s2_bwd");
}

extern "C" int gpemsr_threeda_combine_bwd(const float* feat, const float* attn, const float* dout, int64_t count, float* dfeat,
                                          float* dattn, float* dadd, float* df2, float* df3, void* stream) {
  GP_REQUIRE(feat && attn && dout && dfeat && dattn && dadd && df2 && df3 && count > 0, "threeda_combine_bwd: bad args");
  hipLaunchKernelGGL(threeda_combine_bwd_kernel, dim3(bgrid(count)), dim3(256), 0, ST(stream), feat, attn, dout, (long long)count, dfeat,
                     dattn, dadd, df2, df3);
  return check_launch("threeda_combine_bwd");
}

extern "C" int gpemsr_maxpool2_bwd(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld, float* dx, int dx_ld,
                                   void* stream) {
  GP_REQUIRE(x && dy && dx && h >= 2 && w >= 2, "maxpool2_bwd: bad args");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(bgrid((long long)n * (h / 2) * (w / 2) * c)), dim3(256), 0, ST(stream), x, n, h, w, c, ld,
                     dy, dy_ld, dx, dx_ld);
  return check_launch("maxpool2_bwd");
}

extern "C" int gpemsr_scatter_add_images(const float* dsrc, const int* idx, float* dtarget, int n_src, int n_dst,
                                         int64_t elems_per_image, void* stream) {
  GP_REQUIRE(dsrc && idx && dtarget && n_src > 0 && n_dst > 0 && elems_per_image % 4 == 0, "scatter_add_images: bad args");
  GP_REQUIRE(((reinterpret_cast<uintptr_t>(dsrc) | reinterpret_cast<uintptr_t>(dtarget)) & 15) == 0, "scatter_add_images: alignment");
  hipLaunchKernelGGL(scatter_add_images_kernel, dim3(bgrid((long long)n_src * (elems_per_image / 4))), dim3(256), 0, ST(stream), dsrc, idx,
                     dtarget, n_src, n_dst, (long long)(elems_per_image / 4));
  return check_launch("scatter_add_images");
}

extern "C" int gpemsr_l1_loss(const float* sr, const float* gt, int64_t count, float grad_scale, float* dsr, float* ws,
                              int64_t ws_floats, float* loss, void* stream) {
  GP_REQUIRE(sr && gt && ws && loss && count > 0 && ws_floats >= 1, "l1_loss: bad args");
  long long blocks = (count + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks > ws_floats) blocks = ws_floats;
  hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), sr, gt, (long long)count,
                     grad_scale / (float)count, dsr, ws);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(64), 0, ST(stream), ws, (int)blocks, 1.f / (float)count, loss);
  return check_launch("l1_loss");
}

extern "C" int gpemsr_vq_codebook_loss(const float* z, int z_ld, const float* embedding, const int32_t* idx, int64_t rows, int dim, float beta,
                                       float grad_scale, float* dz, int dz_ld, float* dembedding, float* zq, int zq_ld, float* ws,
                                       int64_t ws_floats, float* loss, void* stream) {
  GP_REQUIRE(z && embedding && idx && zq && ws && loss && rows > 0 && dim > 0 && ws_floats >= 1, "vq_codebook_loss: bad args");
  const long long count = (long long)rows * dim;
  long long blocks = (count + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks > ws_floats) blocks = ws_floats;
  const float inv = 1.f / (float)count;
  hipLaunchKernelGGL(vq_loss_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), z, z_ld, embedding, idx, (long long)rows, dim,
                     2.f * grad_scale * inv, 2.f * grad_scale * beta * inv, dz, dz_ld, dembedding, zq, zq_ld, ws);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(64), 0, ST(stream), ws, (int)blocks, (1.f + beta) * inv, loss);
  return check_launch("vq_codebook_loss");
}

extern "C" int gpemsr_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, void* stream) {
  GP_REQUIRE(p && g && m && v && count > 0 && step >= 1, "adam_step: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(bgrid(count)), dim3(256), 0, ST(stream), p, g, m, v, (long long)count, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)sqrt(bc2));
  return check_launch("adam_step");
}

extern "C" int gpemsr_transpose_images(const float* src, float* dst, int n, int rows, int cols, void* stream) {
  GP_REQUIRE(src && dst && n > 0 && rows > 0 && cols > 0 && n <= 65535, "transpose_images: bad args");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, n), dim3(256), 0, ST(stream), src, dst, rows, cols);
  return check_launch("transpose_images");
}
