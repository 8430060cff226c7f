// Dev probe: bf16 MFMA (32x32x16) rate for the split-bf16 inner loop shape: per tap 8 ds_read_b128 (A hi/lo x2, B hi/lo x2)
// feeding 12 MFMAs (2x2 tiles x {hi*hi, hi*lo, lo*hi}); and the pure-bf16 shape (4 reads, 4 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int SPLIT, int READS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 255) * 1e-3f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* pa = sm + (lane & 31) * 8 + (lane >> 5) * 4;
  bf16x8 ah[2], al[2], bh[2], bl[2];
  for (int m = 0; m < 2; ++m) for (int e = 0; e < 8; ++e) { ah[m][e] = 0x3f80; al[m][e] = 0x3c00; bh[m][e] = 0x3f80; bl[m][e] = 0x3c00; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (READS) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(pa + ((i + t + m) & 7) * 256);
          bh[m] = *reinterpret_cast<const bf16x8*>(pa + 2048 + ((i + t + m) & 7) * 256);
          if (SPLIT) {
            al[m] = *reinterpret_cast<const bf16x8*>(pa + 4096 + ((i + t + m) & 7) * 256);
            bl[m] = *reinterpret_cast<const bf16x8*>(pa + 6144 + ((i + t + m) & 7) * 256);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          if (SPLIT) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }
  float s = 0; for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SPLIT, int READS> void run(float* out, const char* name) {
  for (int bpc : {1, 2, 3}) {
    const int blocks = 256 * bpc, iters = 60000 / bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<SPLIT, READS><<<blocks, 256, 32768>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(a); k<SPLIT, READS><<<blocks, 256, 32768>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * 4 * iters * 3 * 4 * (SPLIT ? 3 : 1);
    const double raw = mfmas * 32 * 32 * 16 * 2;                 // bf16 MFMA flops actually issued
    const double eff = raw / (SPLIT ? 3 : 1);                    // algorithmic (fp32-equivalent) flops
    printf("%-34s waves/SIMD %d: %7.2f ms  raw %7.1f TF  algorithmic %7.1f TF\n", name, bpc, ms, raw / ms / 1e9, eff / ms / 1e9);
  }
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
  run<0, 0>(out, "bf16 mfma only");
  run<0, 1>(out, "bf16 + 4 reads / 4 mfma");
  run<1, 0>(out, "bf16x3 mfma only");
  run<1, 1>(out, "bf16x3 + 8 reads / 12 mfma");
  return 0;
}
