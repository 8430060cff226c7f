// Dev probe: sustained v_mfma_f32_32x32x2_f32 rate vs waves/SIMD (no memory traffic) + in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 64) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)); hipMalloc(&clk, 128 * 8);
  for (int bpc : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * bpc, iters = 20000 / bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<2><<<blocks, 256>>>(out, 100, clk); hipDeviceSynchronize();
    hipEventRecord(a); k<2><<<blocks, 256>>>(out, iters, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[128]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
    printf("blocks/CU %d (waves/SIMD %d): %.2f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", bpc, bpc, ms, flops / ms / 1e9,
           (double)h[0] / (double)h[1] * 100.0);
  }
  return 0;
}
