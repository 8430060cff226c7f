// Kernels of the ADVERSARIAL phase of stage-1 (VQGAN) training, R:train_stage1.py:300-345 for current_step > gan_start:
// the PatchGAN discriminator of R:model/discriminator.py:9-32 -- Conv2d(k4, stride 2 / 1, padding 0), InstanceNorm2d (no affine,
// eps 1e-5), LeakyReLU(0.2) -- its backward, and the second-order pieces of the R1 penalty (R:train_stage1.py:360-372: the gradient
// of |d sum(D(x)) / dx|^2 with respect to D's weights).
//
// The 4x4 convolutions run as GEMMs on the matrix cores: gpemsr_im2col4 gathers the 16 taps of every output pixel into a column
// tensor [pixels][16 cin] (the deformable-convolution path does the same with its sampled columns), gpemsr_conv2d's 1x1 form
// multiplies it with the weights, gpemsr_conv2d_wgrad's 1x1 form gives the weight gradient, and gpemsr_col2im4 folds a column
// gradient back onto the input grid in gather form (fixed summation order, no atomics).  Everything here is fp32 NHWC.
#include "common.h"

namespace gpemsr {

// col[(n*oh + oy)*ow + ox][kp]: k = (ky*4 + kx)*c + ci for k < 16c, zero for 16c <= k < kp
__global__ __launch_bounds__(256) void im2col4_kernel(const float* x, int n, int h, int w, int c, int ld, int stride, int oh, int ow, int kp, float* col) {
  const long long total = (long long)n * oh * ow * kp;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int k = (int)(e % kp);
    long long p = e / kp;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const long long img = p / oh;
    float v = 0.f;
    if (k < 16 * c) {
      const int tap = k / c, ci = k - tap * c;
      v = x[((img * h + oy * stride + (tap >> 2)) * w + ox * stride + (tap & 3)) * (long long)ld + ci];
    }
    col[e] = v;
  }
}

// dx[n][iy][ix][ci] (+)= sum over the taps (ky, kx) whose window position (iy - ky) / s, (ix - kx) / s is integral and inside
__global__ __launch_bounds__(256) void col2im4_kernel(const float* dcol, int n, int h, int w, int c, int stride, int oh, int ow, int kp, float* dx, int dx_ld,
                                                      int accumulate) {
  const long long total = (long long)n * h * w * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ci = (int)(e % c);
    long long p = e / c;
    const int ix = (int)(p % w); p /= w;
    const int iy = (int)(p % h);
    const long long img = p / h;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int ty = iy - ky;
      if (ty < 0 || ty % stride != 0) continue;
      const int oy = ty / stride;
      if (oy >= oh) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int tx = ix - kx;
        if (tx < 0 || tx % stride != 0) continue;
        const int ox = tx / stride;
        if (ox >= ow) continue;
        s += dcol[((img * oh + oy) * ow + ox) * (long long)kp + (ky * 4 + kx) * c + ci];
      }
    }
    float* o = dx + ((img * h + iy) * w + ix) * (long long)dx_ld + ci;
    *o = accumulate ? *o + s : s;
  }
}

__global__ __launch_bounds__(256) void lrelu_slope_kernel(const float* x, long long count, float slope, float* y) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float v = x[e];
    y[e] = v > 0.f ? v : slope * v;
  }
}
// dx (+)= dy * (y > 0 ? 1 : slope)   (y = the activation's output: same sign as its input for slope > 0)
__global__ __launch_bounds__(256) void lrelu_slope_bwd_kernel(const float* dy, const float* y, long long count, float slope, float* dx, int accumulate) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float g = dy[e] * (y[e] > 0.f ? 1.f : slope);
    dx[e] = accumulate ? dx[e] + g : g;
  }
}

// out[0] (+)= scale * sum(x) or scale * sum(x^2): one workgroup, fixed order (deterministic)
__global__ __launch_bounds__(1024) void sum_scaled_kernel(const float* x, long long count, float scale, int square, float* out, int accumulate) {
  __shared__ double red[16];
  double s = 0.0;
  for (long long e = threadIdx.x; e < count; e += 1024) { const double v = x[e]; s += square ? v * v : v; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    const float r = (float)(t * (double)scale);
    out[0] = accumulate ? out[0] + r : r;
  }
}

// Second-order terms of InstanceNorm2d (per image and channel: y = (x - mu) r, r = (var + eps)^-1/2, backward
//   dx = r (dy - mean(dy) - y mean(dy y)) ).  Given G = dL/d(dx):
//   gdy = r (G - mean(G) - y mean(G y))                                                       (the backward operator is symmetric)
//   gx  = -(r^2/n) y S_Gu - r^2 m2 G + (r^2 m2/n) S_G + (r^2 m2/n) y S_Gy - (r^2/n) u S_Gy,   u = dy - m1 - y m2,
//         m1 = mean(dy), m2 = mean(dy y), S_G = sum G, S_Gy = sum G y, S_Gu = sum G u.
// One workgroup per (image, 32 channels): 8 pixel lanes x 32 channels, two passes over the image's pixels.
__global__ __launch_bounds__(256) void instnorm_bwd_bwd_kernel(const float* x, const float* dy, const float* G, const float* mean_rstd, int hw, int c,
                                                               float* gx, float* gdy, int accumulate_gx) {
  const int img = blockIdx.y, c0 = blockIdx.x * 32;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int ch = c0 + cl;
  const bool ok = ch < c;
  const float mu = ok ? mean_rstd[2 * ((long long)img * c + ch)] : 0.f, r = ok ? mean_rstd[2 * ((long long)img * c + ch) + 1] : 0.f;
  const long long base = (long long)img * hw * c;
  double s_dy = 0, s_dyy = 0, s_g = 0, s_gy = 0, s_gdy = 0;
  if (ok)
    for (int p = pl; p < hw; p += 8) {
      const long long o = base + (long long)p * c + ch;
      const float yv = (x[o] - mu) * r, d = dy[o], g = G[o];
      s_dy += d; s_dyy += (double)d * yv; s_g += g; s_gy += (double)g * yv; s_gdy += (double)g * d;
    }
  __shared__ double red[5][8][32];
  red[0][pl][cl] = s_dy; red[1][pl][cl] = s_dyy; red[2][pl][cl] = s_g; red[3][pl][cl] = s_gy; red[4][pl][cl] = s_gdy;
  __syncthreads();
  double t[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { t[k] = 0.0; for (int j = 0; j < 8; ++j) t[k] += red[k][j][cl]; }
  if (!ok) return;
  const double n = (double)hw;
  const float m1 = (float)(t[0] / n), m2 = (float)(t[1] / n), mg = (float)(t[2] / n), mgy = (float)(t[3] / n);
  const float S_G = (float)t[2], S_Gy = (float)t[3];
  const float S_Gu = (float)(t[4] - (t[0] / n) * t[2] - (t[1] / n) * t[3]);
  const float r2n = r * r / (float)hw;
  for (int p = pl; p < hw; p += 8) {
    const long long o = base + (long long)p * c + ch;
    const float yv = (x[o] - mu) * r, d = dy[o], g = G[o];
    const float u = d - m1 - yv * m2;
    gdy[o] = r * (g - mg - yv * mgy);
    const float v = -r2n * yv * S_Gu - r * r * m2 * g + r2n * m2 * S_G + r2n * m2 * yv * S_Gy - r2n * u * S_Gy;
    gx[o] = accumulate_gx ? gx[o] + v : v;
  }
}

static inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 8 ? 65535 * 8 : b));
}

}  // namespace gpemsr

using namespace gpemsr;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int gpemsr_im2col4(const float* x, int n, int h, int w, int c, int ld, int stride, float* col, int kp, void* stream) {
  GP_REQUIRE(x && col && n > 0 && h >= 4 && w >= 4 && c > 0 && ld >= c && (stride == 1 || stride == 2) && kp >= 16 * c, "im2col4: bad args");
  const int oh = (h - 4) / stride + 1, ow = (w - 4) / stride + 1;
  hipLaunchKernelGGL(im2col4_kernel, dim3(grid_for((long long)n * oh * ow * kp)), dim3(256), 0, ST(stream), x, n, h, w, c, ld, stride, oh, ow, kp, col);
  return check_launch("im2col4");
}

extern "C" int gpemsr_col2im4(const float* dcol, int n, int h, int w, int c, int stride, int kp, float* dx, int dx_ld, int accumulate, void* stream) {
  GP_REQUIRE(dcol && dx && n > 0 && h >= 4 && w >= 4 && c > 0 && dx_ld >= c && (stride == 1 || stride == 2) && kp >= 16 * c, "col2im4: bad args");
  const int oh = (h - 4) / stride + 1, ow = (w - 4) / stride + 1;
  hipLaunchKernelGGL(col2im4_kernel, dim3(grid_for((long long)n * h * w * c)), dim3(256), 0, ST(stream), dcol, n, h, w, c, stride, oh, ow, kp, dx, dx_ld, accumulate);
  return check_launch("col2im4");
}

extern "C" int gpemsr_lrelu_slope(const float* x, int64_t count, float slope, float* y, void* stream) {
  GP_REQUIRE(x && y && count > 0 && slope > 0.f, "lrelu_slope: bad args (positive slope)");
  hipLaunchKernelGGL(lrelu_slope_kernel, dim3(grid_for(count)), dim3(256), 0, ST(stream), x, (long long)count, slope, y);
  return check_launch("lrelu_slope");
}

extern "C" int gpemsr_lrelu_slope_bwd(const float* dy, const float* y, int64_t count, float slope, float* dx, int accumulate, void* stream) {
  GP_REQUIRE(dy && y && dx && count > 0 && slope > 0.f, "lrelu_slope_bwd: bad args");
  hipLaunchKernelGGL(lrelu_slope_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, ST(stream), dy, y, (long long)count, slope, dx, accumulate);
  return check_launch("lrelu_slope_bwd");
}

extern "C" int gpemsr_sum_scaled(const float* x, int64_t count, float scale, int square, float* out, int accumulate, void* stream) {
  GP_REQUIRE(x && out && count > 0, "sum_scaled: bad args");
  hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(1024), 0, ST(stream), x, (long long)count, scale, square, out, accumulate);
  return check_launch("sum_scaled");
}

extern "C" int gpemsr_instnorm_bwd_bwd(const float* x, const float* dy, const float* g, const float* mean_rstd, int n, int hw, int c, float* gx, float* gdy,
                                       int accumulate_gx, void* stream) {
  GP_REQUIRE(x && dy && g && mean_rstd && gx && gdy && n > 0 && hw > 0 && c > 0, "instnorm_bwd_bwd: bad args");
  hipLaunchKernelGGL(instnorm_bwd_bwd_kernel, dim3((c + 31) / 32, n), dim3(256), 0, ST(stream), x, dy, g, mean_rstd, hw, c, gx, gdy, accumulate_gx);
  return check_launch("instnorm_bwd_bwd");
}
