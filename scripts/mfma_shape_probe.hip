// Dev probe (round 3): does the chip hold a higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16
// in a loop shaped like the wide 3x3 kernel's stage (MI355X_MICROARCH.md, DVFS give-back item 7)?
// Same output tile per wave (64 x 64 accumulators = 64 registers), operands re-read from LDS by ds_read_b128 (1 KiB per
// 32 KFLOP either way), random bf16 data, 2 or 3 waves per SIMD, one barrier per `BAR` k-steps.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_probe.hip -o scripts/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// LDS: 32 KiB of random bf16 per operand (A rows / B rows, 16-B pieces).  A k-step of 32 (two 32x32x16 steps or one 16x16x32 step).
template <int SHAPE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const unsigned short* __restrict__ rnd, float* out, int iters, int bar) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  for (int i = threadIdx.x; i < 65536 / 16; i += NW * 64) reinterpret_cast<uint4*>(sm)[i] = reinterpret_cast<const uint4*>(rnd)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // fragment: row li, k-half lh -> image [k-half][row][8]: lanes 0-31 read 512 contiguous bytes
    const int fo = (lane & 31) * 16 + (lane >> 5) * 512;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {       // 8 k-steps of 16
        bf16x8 fa[2], fb[2];
        const int base = ((i * 8 + ks) & 15) * 2048;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          fa[m] = *reinterpret_cast<const bf16x8*>(sm + base + m * 1024 + fo);
          fb[m] = *reinterpret_cast<const bf16x8*>(sm + 32768 + base + m * 1024 + fo);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
      }
      if (bar && (i % bar) == bar - 1) __syncthreads();
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  } else {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    // fragment: row l16, k-quarter q -> image [row][4 pieces]: 64 lanes read 1 KiB contiguous (piece swizzled by row quad)
    const int l16 = lane & 15, q = lane >> 4;
    const int perm = (0x1320 >> (4 * ((l16 >> 2) & 3))) & 3;      // row quads 0,1,2,3 -> 0,2,3,1: conflict-free for the b128 lane groups
    const int fo = l16 * 64 + ((q ^ perm) * 16);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {       // 4 k-steps of 32
        bf16x8 fa[4], fb[4];
        const int base = ((i * 4 + ks) & 7) * 4096;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          fa[m] = *reinterpret_cast<const bf16x8*>(sm + base + m * 1024 + fo);
          fb[m] = *reinterpret_cast<const bf16x8*>(sm + 32768 + base + m * 1024 + fo);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
      }
      if (bar && (i % bar) == bar - 1) __syncthreads();
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  }
  out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int SHAPE, int NW> double run(const unsigned short* rnd, float* out, int iters, int bar) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<SHAPE, NW><<<256, NW * 64, 65536>>>(rnd, out, 200, bar); hipDeviceSynchronize();
  hipEventRecord(a); k<SHAPE, NW><<<256, NW * 64, 65536>>>(rnd, out, iters, bar); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = 256.0 * NW * iters * 8 * 4 * 32768.0;
  return flops / ms / 1e9;
}

int main() {
  unsigned short* h = (unsigned short*)malloc(65536);
  srand(1);
  for (int i = 0; i < 32768; ++i) {       // random bf16 in [-1, 1): full-range mantissa, random sign
    float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
  }
  unsigned short* rnd; float* out;
  if (hipMalloc(&rnd, 65536) != hipSuccess || hipMalloc(&out, 256 * 1024 * sizeof(float)) != hipSuccess) return 1;
  hipMemcpy(rnd, h, 65536, hipMemcpyHostToDevice);
  for (int round = 0; round < 3; ++round) {
    for (int bar : {0, 9}) {
      printf("round %d bar %d | 8 waves:  32x32x16 %7.1f TF   16x16x32 %7.1f TF | 12 waves: 32x32x16 %7.1f TF   16x16x32 %7.1f TF\n", round, bar,
             run<32, 8>(rnd, out, 20000, bar), run<16, 8>(rnd, out, 20000, bar), run<32, 12>(rnd, out, 14000, bar), run<16, 12>(rnd, out, 14000, bar));
      fflush(stdout);
    }
  }
  return 0;
}
