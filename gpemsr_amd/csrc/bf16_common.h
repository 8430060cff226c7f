// Shared device helpers of the bf16 MFMA kernels (conv_bf16.hip, vgg_mask.hip): LDS-DMA issue, counted waits, bf16 packing.
#pragma once
#include "common.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void xglds16(unsigned voff, const void* base_, unsigned lds_addr_) {
  // the "s" operands must be provably wave-uniform at THIS statement: values that reach it through loop-carried state can be
  // classed divergent and land in VGPRs ("invalid operand"); a readfirstlane of an SGPR value folds away
  const unsigned long long bv = (unsigned long long)base_;
  const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bv), bhi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
  const void* base = (const void*)(((unsigned long long)bhi << 32) | (unsigned long long)blo);     // (the builtin returns a SIGNED int)
  const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane(lds_addr_);
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

// counted wait: vmcnt is a 6-bit counter on gfx950 (0..63); n is wave-uniform, so this is a scalar jump table
__device__ __forceinline__ void xwait_vmcnt(int n) {
  switch (n) {
#define GP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GP_W(0) GP_W(1) GP_W(2) GP_W(3) GP_W(4) GP_W(5) GP_W(6) GP_W(7) GP_W(8) GP_W(9) GP_W(10) GP_W(11) GP_W(12) GP_W(13) GP_W(14) GP_W(15)
    GP_W(16) GP_W(17) GP_W(18) GP_W(19) GP_W(20) GP_W(21) GP_W(22) GP_W(23) GP_W(24) GP_W(25) GP_W(26) GP_W(27) GP_W(28) GP_W(29) GP_W(30) GP_W(31)
    GP_W(32) GP_W(33) GP_W(34) GP_W(35) GP_W(36) GP_W(37) GP_W(38) GP_W(39) GP_W(40) GP_W(41) GP_W(42) GP_W(43) GP_W(44) GP_W(45) GP_W(46) GP_W(47)
    GP_W(48) GP_W(49) GP_W(50) GP_W(51) GP_W(52) GP_W(53) GP_W(54) GP_W(55) GP_W(56) GP_W(57) GP_W(58) GP_W(59) GP_W(60) GP_W(61) GP_W(62)
#undef GP_W
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
}

// 16-byte LDS read through an explicit LDS (address space 3) pointer: a generic `char*` that the compiler cannot trace back to
// the shared array through loop-carried state becomes flat_load (seen in one 7x7 instantiation: 4x slower)
typedef __attribute__((address_space(3))) const bf16x8 xlds_bf16x8_t;
__device__ __forceinline__ unsigned xlds_addr(const void* shared_ptr) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)shared_ptr;
}
__device__ __forceinline__ bf16x8 xlds_read16(unsigned lds_addr) {
  return *reinterpret_cast<xlds_bf16x8_t*>((unsigned long long)lds_addr);
}

__device__ __forceinline__ unsigned xcvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// t / d and t % d for a run-time divisor with its host-side reciprocal m = floor((2^32 - 1) / d): q = mulhi(t, m) is floor(t / d) or one less
// (t * (2^32 - m d) / (d 2^32) < 1 for every 32-bit t).  The compiler's own run-time division is ~35 instructions; a persistent workgroup
// decodes a tile index (three divisions) per tile -- stamps: ~1,000 cycles per tile and wave with the matrix pipe idle.
__device__ __forceinline__ void xdivmod(int t, int d, unsigned m, int& q, int& r) {
  unsigned qq = __umulhi((unsigned)t, m);
  unsigned rr = (unsigned)t - qq * (unsigned)d;
  if (rr >= (unsigned)d) { ++qq; rr -= (unsigned)d; }
  q = (int)qq; r = (int)rr;
}

__device__ __forceinline__ float xbf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float xbf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }



}  // namespace gpemsr
