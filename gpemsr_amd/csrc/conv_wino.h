// Shared declarations of the fp32 Winograd F(2x2, 3x3) kernels (conv_wino.hip, conv_wino_p.hip): launch parameters, LDS geometry, the
// LDS-DMA and counted-wait helpers.  (Moved out of conv_wino.hip unchanged.)
#pragma once
#include "common.h"
#include <stdlib.h>

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// timing hooks of scripts/attic/wino_probe.py exist only in a probe build (results are wrong on purpose with any of them set)
#ifdef GPEMSR_WINO_PROBE
#define WINO_DBG(P) ((P).dbg)
#else
#define WINO_DBG(P) 0
#endif
struct WinoParams {
  const float* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w;
  int cin_pad, cout;
  const float* weight;            // U [cin_pad / 8][16][cout][8]  (packing.pack_winograd)
  const float* bias; int act;
  const float* residual; int res_ld;
  const float* pixmul;
  float* out; int out_ld;
  int tiles_x, tiles_y, tiles_n;
  int nblocks;
  int tn_group;                   // wide kernel: cout blocks of one pixel tile that are neighbours in the launch order (divides tiles_n)
  unsigned mg_g, mg_x, mg_y, mg_n; // floor((2^32 - 1) / d) of tn_group, tiles_x, tiles_y, n: the persistent kernel decodes a tile per loop trip (wn_divmod)
  float* gn_ws; int gn_parts;     // wide kernel: GroupNorm partial sums of (conv + bias) per (tile, channel): [n][gn_parts][cout][2] (conv_mfma.hip XEPI = 1)
  float* cos_ws;                  // wide kernel, cout == 64: patch-cosine partial sums against `residual` INSTEAD of storing (conv_mfma.hip XEPI = 2)
  int pixshuf, cq;                // wide kernel: store as PixelShuffle(2) (cout index = (2i + j) * cq + c, out is [n][2h][2w][cq])
  int dbg;                        // timing experiments only, compiled in with -DGPEMSR_WINO_PROBE (scripts/attic/wino_probe.py; GPEMSR_WINO_DBG): 1 no fragment reads, 2 no barriers, 4 no DMA after the prologue
};

constexpr int WN_HH = 18, WN_HW2 = 17;                       // halo rows, halo columns per parity
constexpr int WN_ASLOTS = 2 * 2 * WN_HH * WN_HW2;            // 1224 16-byte slots: [quad][parity][row][col / 2]
constexpr int WN_ABYTES = WN_ASLOTS * 16;                    // 19,584
constexpr int WN_BSLOTS = 16 * 32 * 2;                       // [pos][cout][quad]
constexpr int WN_BBYTES = WN_BSLOTS * 16;                    // 16,384
constexpr int WN_STAGE = WN_ABYTES + WN_BBYTES;              // 35,968
constexpr int WN_RING = 4;
constexpr int WN_NA = (WN_ASLOTS + 511) / 512;               // 3 slots per thread
constexpr int WN_NB = WN_BSLOTS / 512;                       // 2
constexpr int WN_EPIX = 36;                                  // floats per (tile) row of the exchange buffer: 32 couts + 4 (bank spread)
constexpr int WN_EBYTES = 4 * 2 * 128 * WN_EPIX * 4;         // 147,456

__device__ __forceinline__ void wn_glds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// t / d and t % d for a run-time divisor with its host-side reciprocal m = floor((2^32 - 1) / d): q = mulhi(t, m) is floor(t / d) or one less
__device__ __forceinline__ void wn_divmod(int t, int d, unsigned m, int& q, int& r) {
  unsigned qq = __umulhi((unsigned)t, m);
  unsigned rr = (unsigned)t - qq * (unsigned)d;
  if (rr >= (unsigned)d) { ++qq; rr -= (unsigned)d; }
  q = (int)qq; r = (int)rr;
}
__device__ __forceinline__ void wn_wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
  }
}


// ---- the 64-cout form (conv_wino2_f32_kernel and its persistent twin) ----
constexpr int W2_HH = 10;
constexpr int W2_ASLOTS = 2 * 2 * W2_HH * WN_HW2;            // 680
constexpr int W2_ABYTES = W2_ASLOTS * 16;                    // 10,880
constexpr int W2_BSLOTS = 16 * 64 * 2;                       // 2048
constexpr int W2_BBYTES = W2_BSLOTS * 16;                    // 32,768
constexpr int W2_STAGE = W2_ABYTES + W2_BBYTES;              // 43,648
constexpr int W2_RING = 3;
constexpr int W2_EPIX = 68;
constexpr int W2_EBYTES = 4 * 2 * 64 * W2_EPIX * 4;          // 139,264

// conv_wino_p.hip: the persistent form of the 64-cout kernel (chunk count % 3 == 2)
int launch_wino2_persistent(const WinoParams& P, hipStream_t st);

}  // namespace gpemsr
