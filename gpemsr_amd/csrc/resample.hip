// Resampling kernels: bilinear resize, pools, SpyNet level input (normalise + flow
// upsample + warp), modulated-deformable column gather.  All are HBM/L2-bound
// gathers: one thread per output float4 (or pixel), coalesced along C.
#include "common.h"

namespace gpemsr {

// PyTorch upsample_bilinear2d source index (aten UpSample.h area_pixel_compute_source_index)
__device__ __forceinline__ void src_index(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float s = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* x, int n, int h, int w, int c, int ld, int oh, int ow,
                                                       int align, float sh, float sw, float mul, float* out, int out_ld) {
  const long long total = (long long)n * oh * ow * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    int y0, y1, x0, x1; float ly, lx;
    src_index(oy, sh, align, h, y0, y1, ly);
    src_index(ox, sw, align, w, x0, x1, lx);
    const float* b = x + (long long)img * h * w * ld + ch;
    const float v00 = b[((long long)y0 * w + x0) * ld], v01 = b[((long long)y0 * w + x1) * ld];
    const float v10 = b[((long long)y1 * w + x0) * ld], v11 = b[((long long)y1 * w + x1) * ld];
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    out[(((long long)img * oh + oy) * ow + ox) * out_ld + ch] = v * mul;
  }
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)n * oh * ow * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    const float* b = x + (((long long)img * h + 2 * oy) * w + 2 * ox) * ld + ch;
    const float v = ((b[0] + b[ld]) + (b[(long long)w * ld] + b[(long long)(w + 1) * ld])) * 0.25f;
    out[(((long long)img * oh + oy) * ow + ox) * out_ld + ch] = v;
  }
}

// nn.MaxPool2d(2, 2) (torchvision vgg19 features 4/9/18/27), floor mode, 4 channels per thread
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld) {
  const int oh = h / 2, ow = w / 2, c4 = c / 4;
  const long long total = (long long)n * oh * ow * c4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = 4 * (int)(e % c4);
    long long p = e / c4;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    const float* b = x + (((long long)img * h + 2 * oy) * w + 2 * ox) * ld + ch;
    const float4 v00 = *reinterpret_cast<const float4*>(b), v01 = *reinterpret_cast<const float4*>(b + ld);
    const float4 v10 = *reinterpret_cast<const float4*>(b + (long long)w * ld), v11 = *reinterpret_cast<const float4*>(b + (long long)(w + 1) * ld);
    float4 r;
    r.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x)); r.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
    r.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z)); r.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
    *reinterpret_cast<float4*>(out + (((long long)img * oh + oy) * ow + ox) * out_ld + ch) = r;
  }
}

__global__ __launch_bounds__(256) void pool3s2_kernel(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld) {
  const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
  const long long total = (long long)n * oh * ow * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ch = (int)(e % c);
    long long p = e / c;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const int img = (int)(p / oh);
    float mx = -INFINITY, sm = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= h) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= w) continue;
        const float v = x[(((long long)img * h + iy) * w + ix) * ld + ch];
        mx = fmaxf(mx, v); sm += v;
      }
    }
    float* o = out + (((long long)img * oh + oy) * ow + ox) * out_ld;
    o[ch] = mx;
    o[c + ch] = sm / 9.f;        // AvgPool2d default count_include_pad=True
  }
}

// One SpyNet pyramid level: see gpemsr_hip.h
__global__ __launch_bounds__(256) void spynet_prep_kernel(const float* ref, const float* supp, const float* fc, int n, int h, int w,
                                                          float m0, float m1, float m2, float s0, float s1, float s2,
                                                          float* up, float* inp, int inp_ld) {
  const long long total = (long long)n * h * w;
  const int ch2 = h / 2, cw2 = w / 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int xq = (int)(e % w);
    const int yq = (int)((e / w) % h);
    const int img = (int)(e / ((long long)w * h));
    float fx = 0.f, fy = 0.f;
    if (fc) {   // bilinear x2 (align_corners=True) of the [h/2][w/2] flow, *2, replicate-padded to h x w
      const int uh = 2 * ch2, uw = 2 * cw2;
      const float sh = uh > 1 ? (float)(ch2 - 1) / (float)(uh - 1) : 0.f;
      const float sw = uw > 1 ? (float)(cw2 - 1) / (float)(uw - 1) : 0.f;
      const int yu = yq < uh ? yq : uh - 1, xu = xq < uw ? xq : uw - 1;   // F.pad(..., mode='replicate')
      int y0, y1, x0, x1; float ly, lx;
      src_index(yu, sh, 1, ch2, y0, y1, ly);
      src_index(xu, sw, 1, cw2, x0, x1, lx);
      const float* b = fc + (long long)img * ch2 * cw2 * 2;
      const float hy = 1.f - ly, hx = 1.f - lx;
#define GP_F(yy, xx, k) b[((long long)(yy) * cw2 + (xx)) * 2 + (k)]
      fx = (hy * (hx * GP_F(y0, x0, 0) + lx * GP_F(y0, x1, 0)) + ly * (hx * GP_F(y1, x0, 0) + lx * GP_F(y1, x1, 0))) * 2.f;
      fy = (hy * (hx * GP_F(y0, x0, 1) + lx * GP_F(y0, x1, 1)) + ly * (hx * GP_F(y1, x0, 1) + lx * GP_F(y1, x1, 1))) * 2.f;
#undef GP_F
    }
    up[2 * e] = fx; up[2 * e + 1] = fy;
    // flow_warp: grid_sample(bilinear, border, align_corners=True) at (x+fx, y+fy)
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
    float* o = inp + e * inp_ld;
    *reinterpret_cast<float4*>(o) = make_float4((rv - m0) / s0, (rv - m1) / s1, (rv - m2) / s2, (wv - m0) / s0);
    *reinterpret_cast<float4*>(o + 4) = make_float4((wv - m1) / s1, (wv - m2) / s2, fx, fy);
    if (inp_ld == 16) {            // zero channels 8..15: the 16-channel form feeds the split-bf16 7x7 kernel (cin % 16 == 0)
      *reinterpret_cast<float4*>(o + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(o + 12) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Modulated deformable column gather: thread per (pixel, group, tap), 8 channels (cg) per thread.
__global__ __launch_bounds__(256) void dcn_columns_kernel(const float* x, int n, int h, int w, int c, int ld,
                                                          const float* om, int om_ld, int groups, float* col) {
  const int cg = c / groups;            // 8
  const int K = 9;
  const long long total = (long long)n * h * w * groups * K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int g = (int)(e % groups);
    long long t = e / groups;
    const int k = (int)(t % K); t /= K;           // tap-major inside a pixel so writes stay contiguous
    const long long pix = t;
    const int xq = (int)(pix % w);
    const int yq = (int)((pix / w) % h);
    const int img = (int)(pix / ((long long)w * h));
    const float* o = om + pix * om_ld;
    // offset = cat(o1,o2): channel j of the 2*groups*K offset tensor is conv_offset channel j
    const float dy = o[g * 2 * K + 2 * k], dx = o[g * 2 * K + 2 * k + 1];
    const float ml = o[2 * groups * K + g * K + k];
    const float m = 1.f / (1.f + expf(-ml));
    const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dx;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    if (py > -1.f && py < (float)h && px > -1.f && px < (float)w) {
      const int y0 = (int)floorf(py), x0 = (int)floorf(px);
      const float ly = py - y0, lx = px - x0;
      const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
      const int ys[4] = {y0, y0, y0 + 1, y0 + 1}, xs[4] = {x0, x0 + 1, x0, x0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (ys[q] >= 0 && ys[q] <= h - 1 && xs[q] >= 0 && xs[q] <= w - 1) {
          const float* xp = x + (((long long)img * h + ys[q]) * w + xs[q]) * ld + g * cg;
          const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
          acc[0] += wts[q] * a.x; acc[1] += wts[q] * a.y; acc[2] += wts[q] * a.z; acc[3] += wts[q] * a.w;
          acc[4] += wts[q] * b.x; acc[5] += wts[q] * b.y; acc[6] += wts[q] * b.z; acc[7] += wts[q] * b.w;
        }
      }
    }
    float* cp = col + pix * (long long)(K * c) + k * c + g * cg;
    *reinterpret_cast<float4*>(cp) = make_float4(acc[0] * m, acc[1] * m, acc[2] * m, acc[3] * m);
    *reinterpret_cast<float4*>(cp + 4) = make_float4(acc[4] * m, acc[5] * m, acc[6] * m, acc[7] * m);
  }
}

inline unsigned grid_for(long long total) {
  const long long b = (total + 255) / 256;
  return (unsigned)(b < 32768 ? (b < 1 ? 1 : b) : 32768);
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_bilinear(const float* x, int n, int h, int w, int c, int ld, int oh, int ow, int align_corners,
                               float mul, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0, "bilinear: bad args");
  float sh, sw;
  if (align_corners) { sh = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f; sw = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f; }
  else { sh = (float)((double)h / (double)oh); sw = (float)((double)w / (double)ow); }
  hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((long long)n * oh * ow * c)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, c, ld, oh, ow, align_corners, sh, sw, mul, out, out_ld);
  return check_launch("bilinear");
}

extern "C" int gpemsr_avgpool2(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && h >= 2 && w >= 2, "avgpool2: bad args");   // odd sizes floor like F.avg_pool2d
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((long long)n * (h / 2) * (w / 2) * c)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, c, ld, out, out_ld);
  return check_launch("avgpool2");
}

extern "C" int gpemsr_maxpool2(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && h >= 2 && w >= 2 && c % 4 == 0 && ld % 4 == 0 && out_ld % 4 == 0, "maxpool2: bad args (c, ld multiples of 4)");
  GP_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "maxpool2: alignment");
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((long long)n * (h / 2) * (w / 2) * (c / 4))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, c, ld, out, out_ld);
  return check_launch("maxpool2");
}

extern "C" int gpemsr_pool3s2_maxavg(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream) {
  GP_REQUIRE(x && out && out_ld >= 2 * c, "pool3s2: bad args");
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  hipLaunchKernelGGL(pool3s2_kernel, dim3(grid_for((long long)n * oh * ow * c)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, c, ld, out, out_ld);
  return check_launch("pool3s2");
}

extern "C" int gpemsr_spynet_prep(const float* ref, const float* supp, const float* flow_coarse, int n, int h, int w,
                                  const float* mean3, const float* std3, float* up_flow, float* inp8, int inp_ld, void* stream) {
  GP_REQUIRE(ref && supp && up_flow && inp8 && mean3 && std3, "spynet_prep: null pointer");
  GP_REQUIRE(h >= 2 && w >= 2, "spynet_prep: level smaller than 2x2 (got %d x %d); basicsr SpyNet needs inputs >= 64 px", h, w);
  GP_REQUIRE((reinterpret_cast<uintptr_t>(inp8) & 15) == 0 && (inp_ld == 8 || inp_ld == 16), "spynet_prep: inp8 alignment / inp_ld must be 8 or 16");
  // mean3/std3 are HOST pointers (3 floats each): constants of the basicsr SpyNet buffers
  hipLaunchKernelGGL(spynet_prep_kernel, dim3(grid_for((long long)n * h * w)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     ref, supp, flow_coarse, n, h, w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], up_flow, inp8, inp_ld);
  return check_launch("spynet_prep");
}

extern "C" int gpemsr_dcn_columns(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                                  float* col, void* stream) {
  GP_REQUIRE(x && om && col, "dcn_columns: null pointer");
  GP_REQUIRE(groups > 0 && c % groups == 0 && c / groups == 8 && ld % 4 == 0, "dcn_columns: needs 8 channels per deformable group");
  GP_REQUIRE(om_ld >= 3 * groups * 9, "dcn_columns: om_ld too small");
  hipLaunchKernelGGL(dcn_columns_kernel, dim3(grid_for((long long)n * h * w * groups * 9)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, n, h, w, c, ld, om, om_ld, groups, col);
  return check_launch("dcn_columns");
}
