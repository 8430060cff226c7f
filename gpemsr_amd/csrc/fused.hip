// GPEMSR-specific reductions / elementwise fusions (HBM-bound; float4 coalesced,
// wave shuffles for the per-pixel channel reductions).
#include "common.h"

namespace gpemsr {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// model/GPEMSR.py:387-395.  One workgroup per 16x16 patch; 256 threads sweep
// 256 pixels x (c/4) float4 with consecutive lanes on consecutive channels.
__global__ __launch_bounds__(256) void patch_cosine_kernel(const float* a, const float* b, int h, int w, int c, float* out) {
  const int pw = w / 16, ph = h / 16;
  const int px = blockIdx.x % pw, py = (blockIdx.x / pw) % ph, img = blockIdx.x / (pw * ph);
  const int c4 = c >> 2;
  const int total = 256 * c4;
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int j = e % c4, p = e / c4;
    const long long off = (((long long)img * h + py * 16 + (p >> 4)) * w + px * 16 + (p & 15)) * c + 4 * j;
    const float4 u = *reinterpret_cast<const float4*>(a + off), v = *reinterpret_cast<const float4*>(b + off);
    dot += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    na += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    nb += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  __shared__ float red[3][4];
  dot = wsum64(dot); na = wsum64(na); nb = wsum64(nb);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dot; red[1][threadIdx.x >> 6] = na; red[2][threadIdx.x >> 6] = nb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float d = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float x = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float y = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    out[blockIdx.x] = d / (fmaxf(sqrtf(x), 1e-12f) * fmaxf(sqrtf(y), 1e-12f));   // F.normalize eps
  }
}

// ThreeDA temporal gate: c/4 lanes per (b,t,pixel); requires c == 64 (16 lanes)
__global__ __launch_bounds__(256) void temporal_gate_kernel(const float* aligned, const float* emb, const float* emb_ref,
                                                            int b, int t, int hw, int c, float* af) {
  const int sub = threadIdx.x & 15;
  const long long items = (long long)b * t * hw;
  const long long per_iter = (long long)gridDim.x * 16;
  const long long niter = (items + per_iter - 1) / per_iter;
  for (long long it = 0; it < niter; ++it) {
    const long long item = it * per_iter + (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = item < items;
    const long long ii = valid ? item : 0;
    const int p = (int)(ii % hw);
    const int ti = (int)((ii / hw) % t);
    const int bi = (int)(ii / ((long long)hw * t));
    const float4 e1 = *reinterpret_cast<const float4*>(emb + ii * c + 4 * sub);
    const float4 e0 = *reinterpret_cast<const float4*>(emb_ref + ((long long)bi * hw + p) * c + 4 * sub);
    float d = e1.x * e0.x + e1.y * e0.y + e1.z * e0.z + e1.w * e0.w;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) d += __shfl_xor(d, m);
    const float g = 1.f / (1.f + expf(-d));
    if (valid) {
      const float4 v = *reinterpret_cast<const float4*>(aligned + ii * c + 4 * sub);
      *reinterpret_cast<float4*>(af + ((long long)bi * hw + p) * ((long long)t * c) + ti * c + 4 * sub) =
          make_float4(v.x * g, v.y * g, v.z * g, v.w * g);
    }
  }
}

// TT > 0: the frame count at compile time (the network's 5) -- with run-time loop bounds in[] is indexed dynamically and lives in scratch
template <int TT>
__global__ __launch_bounds__(256) void frame_mix_kernel(const float* af, long long pixels, int t_rt, int c, const float* m,
                                                        const float* bias, float* out) {
  const int t = TT > 0 ? TT : t_rt;
  const int c4 = c >> 2;
  const long long total = pixels * c4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % c4);
    const long long p = e / c4;
    const float* ip = af + p * ((long long)t * c) + 4 * j;
    float4 in[TT > 0 ? TT : 8];
#pragma unroll
    for (int k = 0; k < t; ++k) in[k] = *reinterpret_cast<const float4*>(ip + k * c);
#pragma unroll
    for (int i = 0; i < t; ++i) {
      float4 s = make_float4(bias[i], bias[i], bias[i], bias[i]);
#pragma unroll
      for (int k = 0; k < t; ++k) {
        const float wv = m[i * t + k];
        s.x = fmaf(wv, in[k].x, s.x); s.y = fmaf(wv, in[k].y, s.y); s.z = fmaf(wv, in[k].z, s.z); s.w = fmaf(wv, in[k].w, s.w);
      }
      s.x = s.x > 0.f ? s.x : 0.1f * s.x; s.y = s.y > 0.f ? s.y : 0.1f * s.y;
      s.z = s.z > 0.f ? s.z : 0.1f * s.z; s.w = s.w > 0.f ? s.w : 0.1f * s.w;
      *reinterpret_cast<float4*>(out + p * ((long long)t * c) + i * c + 4 * j) = s;
    }
  }
}

__global__ __launch_bounds__(256) void threeda_combine_kernel(const float* feat, const float* attn, const float* add,
                                                              const float* f2, const float* f3, long long count4, float* out) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count4; e += (long long)gridDim.x * 256) {
    const float4 f = reinterpret_cast<const float4*>(feat)[e], a = reinterpret_cast<const float4*>(attn)[e];
    const float4 d = reinterpret_cast<const float4*>(add)[e], u = reinterpret_cast<const float4*>(f2)[e];
    const float4 v = reinterpret_cast<const float4*>(f3)[e];
    float4 r;
    r.x = f.x * (1.f / (1.f + expf(-a.x))) * 2.f + d.x + u.x + v.x;
    r.y = f.y * (1.f / (1.f + expf(-a.y))) * 2.f + d.y + u.y + v.y;
    r.z = f.z * (1.f / (1.f + expf(-a.z))) * 2.f + d.z + u.z + v.z;
    r.w = f.w * (1.f / (1.f + expf(-a.w))) * 2.f + d.w + u.w + v.w;
    reinterpret_cast<float4*>(out)[e] = r;
  }
}

__global__ __launch_bounds__(256) void tensor2img_kernel(const float* x, long long count, uint8_t* out) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    float v = fminf(fmaxf(x[e], 0.f), 1.f);
    out[e] = (uint8_t)rintf(v * 255.0f);          // numpy round = round-half-even
  }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* src, int src_ld, float* dst, int dst_ld, long long pixels, int c) {
  const long long total = pixels * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long p = e / c; const int ch = (int)(e % c);
    dst[p * dst_ld + ch] = src[p * src_ld + ch];
  }
}

// dst image j <- src image (j / div) * mul + add   (frame/tile regrouping of the 5-slice windows)
__global__ __launch_bounds__(256) void copy_images_kernel(const float* src, float* dst, long long n_dst, long long elems4,
                                                          int div, int mul, int add) {
  const long long total = n_dst * elems4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long j = e / elems4, k = e % elems4;
    const long long si = (j / div) * mul + add;
    reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[si * elems4 + k];
  }
}

// dst image j = src image idx[j] (volume mode: the 5 cached frames of every window, edge windows repeat a frame)
__global__ __launch_bounds__(256) void gather_images_kernel(const float* src, const int* idx, float* dst, long long n_dst,
                                                            long long elems4) {
  const long long total = n_dst * elems4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long j = e / elems4, k = e % elems4;
    reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[(long long)idx[j] * elems4 + k];
  }
}

inline unsigned grid_for(long long total) {
  const long long b = (total + 255) / 256;
  return (unsigned)(b < 32768 ? (b < 1 ? 1 : b) : 32768);
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_patch_cosine(const float* a, const float* b, int n, int h, int w, int c, float* out, void* stream) {
  GP_REQUIRE(a && b && out, "patch_cosine: null pointer");
  GP_REQUIRE(h % 16 == 0 && w % 16 == 0 && c % 4 == 0, "patch_cosine: needs h,w multiples of 16 (reflect 'same' padding of "
             "model/GPEMSR.py:14-30 is not implemented; the forward asserts LR sizes that never need it)");
  hipLaunchKernelGGL(patch_cosine_kernel, dim3(n * (h / 16) * (w / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a, b, h, w, c, out);
  return check_launch("patch_cosine");
}

extern "C" int gpemsr_temporal_gate(const float* aligned, const float* emb, const float* emb_ref, int b, int t, int hw, int c,
                                    float* af, void* stream) {
  GP_REQUIRE(aligned && emb && emb_ref && af, "temporal_gate: null pointer");
  GP_REQUIRE(c == 64, "temporal_gate: c must be 64 (16 lanes x float4)");
  const long long items = (long long)b * t * hw;
  const long long blocks = (items + 15) / 16;
  hipLaunchKernelGGL(temporal_gate_kernel, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), aligned, emb, emb_ref, b, t, hw, c, af);
  return check_launch("temporal_gate");
}

extern "C" int gpemsr_frame_mix_lrelu(const float* af, int64_t pixels, int t, int c, const float* m, const float* bias,
                                      float* out, void* stream) {
  GP_REQUIRE(af && m && bias && out && t <= 8 && c % 4 == 0, "frame_mix: bad args");
  if (t == 5)
    hipLaunchKernelGGL(frame_mix_kernel<5>, dim3(grid_for(pixels * (c / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       af, (long long)pixels, t, c, m, bias, out);
  else
    hipLaunchKernelGGL(frame_mix_kernel<0>, dim3(grid_for(pixels * (c / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       af, (long long)pixels, t, c, m, bias, out);
  return check_launch("frame_mix");
}

extern "C" int gpemsr_threeda_combine(const float* feat, const float* attn, const float* attn_add, const float* f2,
                                      const float* f3, int64_t count, float* out, void* stream) {
  GP_REQUIRE(feat && attn && attn_add && f2 && f3 && out && count % 4 == 0, "threeda_combine: bad args");
  hipLaunchKernelGGL(threeda_combine_kernel, dim3(grid_for(count / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     feat, attn, attn_add, f2, f3, (long long)(count / 4), out);
  return check_launch("threeda_combine");
}

extern "C" int gpemsr_tensor2img_u8(const float* x, int64_t count, uint8_t* out, void* stream) {
  GP_REQUIRE(x && out && count > 0, "tensor2img: bad args");
  hipLaunchKernelGGL(tensor2img_kernel, dim3(grid_for(count)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, (long long)count, out);
  return check_launch("tensor2img");
}

extern "C" int gpemsr_copy_channels(const float* src, int src_ld, float* dst, int dst_ld, int64_t pixels, int c, void* stream) {
  GP_REQUIRE(src && dst && pixels > 0 && c > 0, "copy_channels: bad args");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(pixels * c)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, src_ld, dst, dst_ld, (long long)pixels, c);
  return check_launch("copy_channels");
}

extern "C" int gpemsr_copy_images(const float* src, float* dst, int64_t n_dst, int64_t elems_per_image, int div, int mul, int add,
                                  void* stream) {
  GP_REQUIRE(src && dst && n_dst > 0 && elems_per_image > 0 && elems_per_image % 4 == 0 && div > 0, "copy_images: bad args");
  GP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "copy_images: alignment");
  hipLaunchKernelGGL(copy_images_kernel, dim3(grid_for(n_dst * (elems_per_image / 4))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, (long long)n_dst, (long long)(elems_per_image / 4), div, mul, add);
  return check_launch("copy_images");
}

extern "C" int gpemsr_gather_images(const float* src, const int* idx, float* dst, int64_t n_dst, int64_t elems_per_image, void* stream) {
  GP_REQUIRE(src && idx && dst && n_dst > 0 && elems_per_image > 0 && elems_per_image % 4 == 0, "gather_images: bad args");
  GP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "gather_images: alignment");
  hipLaunchKernelGGL(gather_images_kernel, dim3(grid_for(n_dst * (elems_per_image / 4))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, idx, dst, (long long)n_dst, (long long)(elems_per_image / 4));
  return check_launch("gather_images");
}
