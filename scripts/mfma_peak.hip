// Dev probe: sustained v_mfma_f32_32x32x2_f32 rate with (a) nothing else, (b) ds_read_b128 fragments per 8 MFMAs,
// (c) a workgroup barrier every 24 MFMAs, (d) both; at 2..5 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const int lane = threadIdx.x & 63;
  const float* pa = sm + (lane & 31) * 8 + (lane >> 5) * 4;
  float4 a0 = make_float4(1, 2, 3, 4), a1 = a0, b0 = a0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (MODE & 1) {
        a0 = *reinterpret_cast<const float4*>(pa + ((i + t) & 7) * 256);
        a1 = *reinterpret_cast<const float4*>(pa + 2048 + ((i + t) & 7) * 256);
        b0 = *reinterpret_cast<const float4*>(pa + 1024 + ((i + t) & 3) * 256);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc1, 0, 0, 0);
    }
    if (MODE & 2) __syncthreads();
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* out, const char* name) {
  for (int bpc : {2, 3, 4, 5}) {
    const int blocks = 256 * bpc, iters = 24000 / bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256, 16384>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<blocks, 256, 16384>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 24 * 32 * 32 * 2 * 2;
    printf("%-28s waves/SIMD %d: %7.2f ms  %6.1f TFLOP/s\n", name, bpc, ms, flops / ms / 1e9);
  }
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
  run<0>(out, "mfma only");
  run<1>(out, "+3 ds_read_b128 / 8 mfma");
  run<2>(out, "+barrier / 24 mfma");
  run<3>(out, "+both");
  return 0;
}
