// Shared device helpers of the bf16 MFMA kernels (conv_bf16.hip, vgg_mask.hip): LDS-DMA issue, counted waits, bf16 packing.
#pragma once
#include "common.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void xglds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

// make a wave-uniform value provably uniform for the "s" operands of the DMA statement (guide 5.7 / T20)
__device__ __forceinline__ const void* xuni_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned xuni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void xwait_vmcnt(int n) {
  switch (n) {
#define GP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GP_W(0) GP_W(1) GP_W(2) GP_W(3) GP_W(4) GP_W(5) GP_W(6) GP_W(7) GP_W(8) GP_W(9) GP_W(10) GP_W(11) GP_W(12) GP_W(13) GP_W(14) GP_W(15)
    GP_W(16) GP_W(17) GP_W(18) GP_W(19) GP_W(20) GP_W(21) GP_W(22) GP_W(23) GP_W(24) GP_W(25) GP_W(26) GP_W(27) GP_W(28) GP_W(29) GP_W(30) GP_W(31)
#undef GP_W
    default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
  }
}

__device__ __forceinline__ unsigned xcvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float xbf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float xbf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }



}  // namespace gpemsr
