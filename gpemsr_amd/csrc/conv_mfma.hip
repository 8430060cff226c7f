// Implicit-GEMM convolution on the CDNA4 f32 matrix pipe (v_mfma_f32_32x32x2_f32).   [v2]
//
// One workgroup (4 waves) produces a 4x32-pixel x BN-channel output tile:
//   D[pixel][cout] = sum_{tap} sum_{cin} In[pixel*stride + tap][cin] * W[tap][cout][cin]
// K loop = (cin chunk of CK channels) x (tap group).  Per chunk the input halo tile
// ((4-1)*s+kh) x ((32-1)*s+kw) x CK is staged ONCE in LDS and every tap reads it at a shifted
// offset (k*k-fold reuse out of LDS, no im2col in HBM); the weight slice [taps][BN][CK] of the
// stage sits beside it.  Both are double-buffered: stage s+1 is written to the other buffer
// before the MFMAs of stage s and the global loads of stage s+2 are in flight in registers
// during them -- ONE barrier per stage.  Inside a stage the tap loop is software-pipelined
// (fragments of tap t+1 are read from LDS while the MFMAs of tap t issue) and tap offsets are
// scalar arithmetic.
//
// MFMA operand maps (guide section 3): A[i = lane&31][k = lane>>5], B[k][j = lane&31],
// D col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel).  A lane reads one
// float4 of A and of B per tap (ds_read_b128): element e is k-step e, in which lane half h
// supplies channel 4h+e of the chunk -- A and B agree, so the k order is free.
//
// Epilogue: accumulators go through LDS ([pixel][BN+4]) and leave as coalesced float4 rows
// with bias / activation / residual / per-pixel multiplier applied; PixelShuffle(2) and the
// transposed convolution only change the (pixel, channel) -> address map of that store.
//
// ConvTranspose2d(k3,s2,p1,op1) is evaluated as a 2x2-tap convolution with 4*Cout
// "phase-stacked" output channels (n' = (co/32)*128 + q*32 + co%32, q = 2*py+px): tap (dy,dx)
// only feeds the phases with py>=dy, px>=dx, so each wave owns all four phases of 32 channels
// and skips the zero (tap, phase) blocks through a per-tap N-tile mask: 9 MFMA blocks per 4
// output pixels = the true FLOP count, balanced across waves.
//
// Staging comes in two flavours.  DMA (the fast one, used whenever every source has
// c % CK == 0 and 16-B aligned rows): global_load_lds_dwordx4 writes the LDS images directly
// (1 KiB per wave-instruction, no VGPR round trip, no ds_write; out-of-image halo pixels and
// channels past cout read a 16-B zero constant).  The LDS images are therefore lane-linear;
// the 128-B rows of the 1x1/GEMM case are XOR-swizzled on the SOURCE address and on the read
// (guide rule 21) so ds_read_b128 stays conflict-free.  The register-staged flavour (global ->
// VGPR -> ds_write, zero-padding by predication) serves 1/2/34-channel sources.
//
// Replaces the ATen conv / conv_transpose / bmm / linear calls under model/GPEMSR.py:323-456.
#include "common.h"
#include <stdlib.h>

namespace gpemsr {


typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef GP_STAMP
__device__ unsigned long long g_stamps[8 * 65536];
#define GP_ST(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) g_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GP_ST(i)
#endif

constexpr int TILE_W = 32;                // a 32-pixel MFMA tile is one contiguous tile row: conflict-free A fragment reads
constexpr int A_LOADS = 5;   // float4 per thread per A stage (17*33 halo px * 2 / 256 -> 5)

enum { STORE_PLAIN = 0, STORE_PIXSHUF = 1, STORE_CONVT = 2, STORE_ROWPAIR = 3 };

struct ConvParams {
  const float* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int vec[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w;            // input geometry
  int oh, ow;             // conv grid
  int OH, OW;             // stored output geometry
  int cin_pad, cout;      // cout = GEMM N (4*Cout for the transposed form)
  int kh, kw, stride, pad;
  int stride_x;           // horizontal stride (= stride except in the row-pair form: rows step 2, columns 1)
  const float* weight; long long w_img_stride;
  const float* bias; int act;
  const float* residual; int res_ld;
  const float* pixmul;
  int store_mode, cq;     // cq: channels per sub-pixel group (PIXSHUF)
  float* out; int out_ld;
  int out_vec, res_vec;
  int tiles_x, tiles_y, tiles_n;
  int halo_h, halo_w;
  int tpg_h, tpg_w, ngroups;      // taps of one stage: tpg_h x tpg_w (rows x cols of the filter)
  int a_buf_floats, b_buf_floats;
  int na, nb;                     // DMA slots (float4 per thread) per A / B stage
  int ring;                       // DMA: weight images in a 3-deep ring, counted vmcnt (needs ngroups >= 2 or not; see kernel)
  int nblocks;
  int tw_lg;                      // log2 of the tile width in pixels: 5 (TH x 32 tiles) or 4 (2*TH x 16 tiles, for maps <= 16 wide)
  int epi_fast;                   // lean epilogue: float4 rows, cout % 4 == 0, 32-bit byte offsets, 16-B aligned bias
  float* gn_ws; int gn_parts;     // optional GroupNorm partial sums of (conv + bias): [n][gn_parts][cout][2] (fast epilogue only)
  float* cos_ws;                  // optional: patch-cosine partial sums against the `residual` operand INSTEAD of storing the result
};

// One LDS-DMA piece: lane l's 16 bytes at (base + voff) land at LDS byte (lds_addr + 16*l); base and lds_addr are
// wave-uniform (SGPRs), voff is the lane's 32-bit byte offset; lanes masked off by EXEC write nothing.
// Issued through inline asm ON PURPOSE: hipcc treats a builtin LDS-DMA as a pending LDS store and drains it with
// s_waitcnt vmcnt(0) before the next ds_read, which serialises the pipeline; an asm DMA is outside its wait-count
// bookkeeping (guide 5.7 item 1), so the counted waits below (wait_vmcnt + s_barrier) are the ONLY ordering.
__device__ __forceinline__ void glds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}

// s_waitcnt vmcnt(n) with a wave-uniform runtime n (the instruction needs an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
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
    default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
  }
}

// XEPI = 1 / 2: the instantiations whose lean epilogue also forms GroupNorm (1) / patch-cosine (2) partial sums (1: no residual or
// multiplier operands exist; 2: nothing is stored).  A separate instantiation ON PURPOSE: with
// that code in the common kernels the 168-register budget overflowed (164-232 bytes of scratch per lane in EVERY launch, +150 GB of
// HBM traffic per step); the common kernels compile exactly as before and only the launches that ask for the sums pay for them.
template <int CK, int BN, int WM, int WN, int TH, bool MASKED, bool DMA, int XEPI = 0>
__global__ __launch_bounds__(256, (MASKED && TH == 8) ? 2 : 3) void conv_mfma_kernel(ConvParams P) {
  constexpr int NPIX = TH * TILE_W;   // output pixels per workgroup (TH x 32)
  constexpr int PM = NPIX / WM;       // pixels per wave
  constexpr int MT = PM / 32;
  constexpr int WNT = BN / WN;        // couts per wave
  constexpr int NT = WNT / 32;
  constexpr int APIX = (CK == 8 || DMA) ? CK : CK + 4;   // floats per halo pixel in LDS (DMA images are unpadded)
  constexpr bool SWZ = DMA && CK == 32;                  // XOR-swizzled 128-B rows
  constexpr int BPIX = APIX;
  constexpr int V4 = CK / 4;          // float4 per pixel per chunk
  constexpr int KJ = CK / 8;          // 8-channel groups per chunk
  constexpr int MAXTPG = (CK == 32) ? 1 : (BN == 128 ? 4 : 7);
  constexpr int B_LOADS = (MAXTPG * BN * V4 + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * P.a_buf_floats;

  GP_ST(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- XCD-aware block id remap (bijective; consecutive logical tiles share an XCD/L2) ----
  int bid = blockIdx.x;
  {
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = bid;
  const int tn = t % P.tiles_n; t /= P.tiles_n;
  const int tx = t % P.tiles_x; t /= P.tiles_x;
  const int ty = t % P.tiles_y; t /= P.tiles_y;
  const int img = t;
  const int tw_lg = P.tw_lg, tw_mask = (1 << tw_lg) - 1;
  const int oy0 = ty * (NPIX >> tw_lg), ox0 = tx << tw_lg, n0 = tn * BN;
  const int S = P.stride, SX = P.stride_x;
  const int iy0 = oy0 * S - P.pad, ix0 = ox0 * SX - P.pad;
  const int halo_px = P.halo_h * P.halo_w;
  const int tpg = P.tpg_h * P.tpg_w;

  // ---- per-thread staging slots (independent of the stage) ----
  // register path: a_pix/a_lds/a_ch + b_goff/b_lds ; DMA path: byte offsets a_boff/b_boff (INVALID -> zero source)
  int a_pix[A_LOADS];      // pixel index inside the image, or -1
  int a_lds[A_LOADS];      // LDS float offset, or -1
  int a_ch[A_LOADS];       // channel offset of the float4 inside the chunk
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int e = tid + i * 256;
    a_pix[i] = -1; a_lds[i] = -1; a_ch[i] = 0;
    if (e < halo_px * V4) {
      const int hp = e / V4, pc = e % V4;
      const int j = SWZ ? (pc ^ (hp & 7)) : pc;           // logical 16-B chunk held at physical chunk pc
      const int hy = hp / P.halo_w, hx = hp % P.halo_w;
      const int iy = iy0 + hy, ix = ix0 + hx;
      a_lds[i] = hp * APIX + 4 * pc; a_ch[i] = 4 * j;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pix[i] = iy * P.w + ix;
    }
  }
  int b_goff[B_LOADS];     // float offset of the float4 inside the weight tensor (stage offset excluded), or -1
  int b_lds[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int e = tid + i * 256;
    b_goff[i] = -1; b_lds[i] = -1;
    if (e < tpg * BN * V4) {
      const int row = e / V4, pc = e % V4;
      const int j = SWZ ? (pc ^ (row & 7)) : pc;
      const int tt = row / BN, nn = row % BN;
      b_lds[i] = row * BPIX + 4 * pc;
      if (n0 + nn < P.cout) b_goff[i] = (tt * P.cout + n0 + nn) * P.cin_pad + 4 * j;
    }
  }
  // per-lane fragment offsets (floats)
  int a_frag[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = wm * PM + mt * 32 + li;
    a_frag[mt] = (((p >> tw_lg) * S) * P.halo_w + (p & tw_mask) * SX) * APIX + (SWZ ? 0 : 4 * lh);
  }
  const int b_frag = (wn * WNT + li) * BPIX + (SWZ ? 0 : 4 * lh);
  const int swz_x = li & 7;            // SWZ: row & 7 of this lane's A and B rows (tile bases are multiples of 8)

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- stage enumeration (chunk-major, tap-group-minor), advanced incrementally ----
  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += (P.c[s] + CK - 1) / CK;
  const int ngroups = P.ngroups;
  const int nstages = nchunks * ngroups;
  const float* wbase = P.weight + (long long)img * P.w_img_stride;
  const int grp_stride = tpg * P.cout * P.cin_pad;     // weight floats between tap groups

  // state of the NEXT stage to be fetched
  int f_src = 0, f_c0 = 0, f_cpad = 0, f_grp = 0, f_chunk = 0, f_stage = 0;
  float4 ra[A_LOADS];
  float4 rb[B_LOADS];

  auto advance = [&]() {
    ++f_stage;
    if (++f_grp == ngroups) {
      f_grp = 0; ++f_chunk; f_c0 += CK;
      const int cp = ((P.c[f_src] + CK - 1) / CK) * CK;
      if (f_c0 >= cp && f_src + 1 < P.nsrc) { f_c0 = 0; f_cpad += cp; ++f_src; }
    }
  };
  auto fetch = [&]() {      // register path: global -> registers for the next stage
    if (f_grp == 0) {
      const float* sp = P.src[f_src] + (long long)img * P.img_stride[f_src];
      const int ld = P.ld[f_src], cs = P.c[f_src];
      const bool vec = P.vec[f_src] != 0;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_pix[i] >= 0) {
          const int cc = f_c0 + a_ch[i];
          const float* gp = sp + (long long)a_pix[i] * ld + cc;
          if (vec && cc + 3 < cs) {
            v = *reinterpret_cast<const float4*>(gp);
          } else {
            if (cc + 0 < cs) v.x = gp[0];
            if (cc + 1 < cs) v.y = gp[1];
            if (cc + 2 < cs) v.z = gp[2];
            if (cc + 3 < cs) v.w = gp[3];
          }
        }
        ra[i] = v;
      }
    }
    const float* wp = wbase + (long long)f_grp * grp_stride + f_cpad + f_c0;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_goff[i] >= 0) v = *reinterpret_cast<const float4*>(wp + b_goff[i]);
      rb[i] = v;
    }
    advance();
  };
  auto commit = [&](int chunk, int grp, int stage) {   // register path: registers -> the LDS buffers of that stage
    if (grp == 0) {
      float* A = As0 + (chunk & 1) * P.a_buf_floats;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        if (a_lds[i] >= 0) *reinterpret_cast<float4*>(A + a_lds[i]) = ra[i];
    }
    float* B = Bs0 + (stage & 1) * P.b_buf_floats;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      if (b_lds[i] >= 0) *reinterpret_cast<float4*>(B + b_lds[i]) = rb[i];
  };
  // DMA path: the next stage's images go global -> LDS directly.  Lane l of wave w, slot i, lands at
  // image byte 16 * (i*256 + 64*w + l): the images are linear in the slot index by construction.
  // DMA path.  Lane l of wave w, slot i lands at image byte 16*(i*256 + 64*w + l): the images are linear in the slot
  // index by construction.  Slots that never receive data (out-of-image halo pixels, rows past cout, the padding of
  // the last 1-KiB piece) are zeroed ONCE below and their lanes stay masked off in every DMA.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem) + (unsigned)wave * 1024u);   // wave's 1-KiB piece of slot 0
  const int nbuf_b = DMA ? 3 : 2;
  if (DMA) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      if (i < P.na && a_pix[i] < 0)
        for (int bsel = 0; bsel < 2; ++bsel) *reinterpret_cast<float4*>(As0 + bsel * P.a_buf_floats + (tid + i * 256) * 4) = z;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      if (i < P.nb && b_goff[i] < 0)
        for (int bsel = 0; bsel < 3; ++bsel) *reinterpret_cast<float4*>(Bs0 + bsel * P.b_buf_floats + (tid + i * 256) * 4) = z;
  }
  // DMA instructions THIS WAVE really issues per image: a slot whose 64 lanes are all masked off is branched over
  // (s_cbranch_execz), so the counted waits must not include it.
  int na_w = 0, nb_w = 0;
  if (DMA) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) na_w += (i < P.na && __ballot(a_pix[i] >= 0) != 0ull) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) nb_w += (i < P.nb && __ballot(b_goff[i] >= 0) != 0ull) ? 1 : 0;
  }
  auto dma_b = [&]() -> int {                   // weight image of the fetch state's stage -> ring slot f_stage % 3
    const float* wp = wbase + (long long)f_grp * grp_stride + f_cpad + f_c0;
    const unsigned lb = lds0 + (unsigned)(2 * P.a_buf_floats + (f_stage % nbuf_b) * P.b_buf_floats) * 4u;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      if (i < P.nb && b_goff[i] >= 0) glds16((unsigned)b_goff[i] * 4u, wp, lb + i * 4096u);
    return nb_w;
  };

  // ---- prologue ----
  // ring mode (DMA only): the fetch state runs TWO stages ahead for the weight images; the activation image of
  // chunk c+1 is issued during the FIRST stage of chunk c.  a_next_* is a second fetch cursor for the A images.
  int an_src = 0, an_c0 = 0, an_chunk = 0;      // chunk whose A image is issued next (ring mode)
  auto a_cursor_advance = [&]() {
    ++an_chunk; an_c0 += CK;
    const int cp = ((P.c[an_src] + CK - 1) / CK) * CK;
    if (an_c0 >= cp && an_src + 1 < P.nsrc) { an_c0 = 0; ++an_src; }
  };
  auto dma_a_cursor = [&]() -> int {            // activation image of chunk an_chunk -> A buffer an_chunk & 1
    const float* sp = P.src[an_src] + (long long)img * P.img_stride[an_src] + an_c0;
    const unsigned pixb = (unsigned)P.ld[an_src] * 4u;
    const unsigned la = lds0 + (unsigned)((an_chunk & 1) * P.a_buf_floats) * 4u;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      if (i < P.na && a_pix[i] >= 0) glds16((unsigned)a_pix[i] * pixb + 4u * (unsigned)a_ch[i], sp, la + i * 4096u);
    a_cursor_advance();
    return na_w;
  };
  if (DMA) {
    dma_a_cursor();                              // A(0)
    dma_b(); advance();                          // B(0)
    int infl = 0;
    if (nstages > 1) { infl = dma_b(); advance(); }   // B(1) may stay in flight
    wait_vmcnt(infl);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero-fill ds_writes above
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    fetch();
    commit(0, 0, 0);
    if (nstages > 1) fetch();
    __syncthreads();
  }

  GP_ST(1);
  int chunk = 0, grp = 0;
  for (int stage = 0; stage < nstages; ++stage) {
    int nchunk = chunk, ngrp = grp + 1;
    if (ngrp == ngroups) { ngrp = 0; ++nchunk; }
    int issued = 0;                                       // ring mode: DMA instructions issued during this stage
    if (DMA) {
      // order: A(chunk+1) first, then B(stage+2).  At the end of the stage we wait until only THIS stage's issues
      // (minus the A image when the next stage already needs it, i.e. ngroups == 1) may still be in flight.
      int a_issued = 0;
      if (grp == 0 && an_chunk < nchunks) a_issued = dma_a_cursor();
      if (stage + 2 < nstages) { issued = dma_b(); advance(); }
      if (ngroups > 1) issued += a_issued;
    } else {
      // stage+1 -> the other LDS buffers (its loads were issued a whole stage ago)
#ifndef GP_EXP_NO_COMMIT
      if (stage + 1 < nstages) commit(nchunk, ngrp, stage + 1);
#endif
#ifndef GP_EXP_NO_FETCH
      if (stage + 2 < nstages) fetch();                   // stage+2 in flight during the MFMAs below
#endif
    }

    const float* A = As0 + (chunk & 1) * P.a_buf_floats;
    const float* B = Bs0 + (stage % nbuf_b) * P.b_buf_floats + b_frag;
    // tap loop, software pipelined by two
    const int ky0 = grp * P.tpg_h;
    const int nsteps = tpg * KJ;
    float4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    int s_ky = 0, s_kx = 0, s_kj = 0;                   // coordinates of the NEXT step to load
    auto load_step = [&](float4 (&fa)[MT], float4 (&fb)[NT], unsigned& mask) {
      const int tt = s_ky * P.tpg_w + s_kx;
      // SWZ: this lane's 16-B chunk (2*kj + lh) of a 128-B row sits at physical chunk (2*kj + lh) ^ (row & 7)
      const int koff = SWZ ? 4 * (((2 * s_kj + lh) ^ swz_x)) : 8 * s_kj;
      const int aoff = ((ky0 + s_ky) * P.halo_w + s_kx) * APIX + koff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const float4*>(A + a_frag[mt] + aoff);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const float4*>(B + (tt * BN + nt * 32) * BPIX + koff);
      if (MASKED) {     // transposed form: tap (dy,dx) feeds phases q = 2py+px with py >= dy, px >= dx
        mask = s_ky ? (s_kx ? 0x8u : 0xCu) : (s_kx ? 0xAu : 0xFu);
      } else {
        mask = 0xFu;
      }
      if (++s_kj == KJ) { s_kj = 0; if (++s_kx == P.tpg_w) { s_kx = 0; ++s_ky; } }
    };
    auto mma_step = [&](const float4 (&fa)[MT], const float4 (&fb)[NT], unsigned mask) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (MASKED && !((mask >> nt) & 1u)) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].x, fb[nt].x, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].y, fb[nt].y, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].z, fb[nt].z, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].w, fb[nt].w, acc[mt][nt], 0, 0, 0);
        }
      }
    };
    if (MASKED && CK == 8) {
      // transposed form: the stage is always the 4 taps (dy,dx) of the 2x2 stencil, so the tap loop is unrolled with
      // COMPILE-TIME phase masks -- no branches between the MFMAs, fragments of masked N tiles are never read
      float4 ta[2][MT], tb[2][NT];
      auto tload = [&](int set, int t) {
        const unsigned mk = (t == 0) ? 0xFu : (t == 1) ? 0xAu : (t == 2) ? 0xCu : 0x8u;
        const int aoff = ((t >> 1) * P.halo_w + (t & 1)) * APIX;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ta[set][mt] = *reinterpret_cast<const float4*>(A + a_frag[mt] + aoff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if ((mk >> nt) & 1u) tb[set][nt] = *reinterpret_cast<const float4*>(B + (t * BN + nt * 32) * BPIX);
      };
      auto tmma = [&](int set, int t) {
        const unsigned mk = (t == 0) ? 0xFu : (t == 1) ? 0xAu : (t == 2) ? 0xCu : 0x8u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (!((mk >> nt) & 1u)) continue;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[set][mt].x, tb[set][nt].x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[set][mt].y, tb[set][nt].y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[set][mt].z, tb[set][nt].z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[set][mt].w, tb[set][nt].w, acc[mt][nt], 0, 0, 0);
          }
        }
      };
      tload(0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) tload((t + 1) & 1, t + 1);
        tmma(t & 1, t);
      }
    } else {
    unsigned m0 = 0xF, m1 = 0xF;
    load_step(fa0, fb0, m0);
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {
      load_step(fa1, fb1, m1);
      mma_step(fa0, fb0, m0);
      if (st + 2 < nsteps) load_step(fa0, fb0, m0);
      mma_step(fa1, fb1, m1);
    }
    if (st < nsteps) mma_step(fa0, fb0, m0);
    }
    if (DMA) {
      // everything issued BEFORE this stage has landed (this wave's part); the barrier makes all waves' parts visible
      // and guarantees every wave is done reading the buffers the next stage's DMA will overwrite.
      wait_vmcnt(issued);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");        // no LDS read of the next stage may be hoisted above the barrier
    } else {
      __syncthreads();   // (a) every wave finished reading this stage's buffers, (b) stage+1's writes/DMA are visible
    }
    chunk = nchunk; grp = ngrp;
  }

  GP_ST(2);
  // ---- epilogue: accumulators -> LDS [pixel][EW+4] (64 output columns per pass) -> coalesced rows ----
  // Everything the epilogue derives from the thread index is tile-invariant; hoisted above the main loop by the compiler it cost the
  // 168-register instantiations five spilled registers (20 bytes per lane of scratch in the two instantiations that run 40 % of the fp32
  // step: profiles/r03_scratch.txt).  An opaque copy of the thread index keeps that arithmetic here.
  int tid_epi = threadIdx.x;
  asm volatile("" : "+v"(tid_epi));
  {
  const int tid = tid_epi;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int EW = BN < 64 ? BN : 64;
  constexpr int EPIX = EW + 4;
  constexpr int NCP = BN / EW;         // column passes
  constexpr int NPP = NPIX / 128;      // pixel passes (128 pixels per pass)
  constexpr int NV = EW / 4;
  float* E = smem;
  if (DMA) __syncthreads();
  constexpr int ITER = (128 * NV) / 256;      // readout float4 per thread per pass (8 for 64 columns, 4 for 32)
  constexpr int PSTEP = 256 / NV;             // pixels between a thread's consecutive items
  const int ej = tid % NV, ep0 = tid / NV;    // this thread's float4 column and first pixel of every pass
  // per-image bases (wave-uniform) + 32-bit in-image pixel indices keep the item state small
  const long long img_pix0 = (long long)img * P.OH * P.OW;
  const float* res_img = P.residual ? P.residual + img_pix0 * P.res_ld : nullptr;
  const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
  float* out_img = P.out + img_pix0 * P.out_ld;
  // The f32 MFMA executes on the vector ALUs, so every VALU instruction of this epilogue (and of the prologue) is paid in
  // matrix-pipe time of the co-resident workgroups: in-kernel s_memtime stamps showed the general epilogue below taking
  // 13 % (64->64 convs) to 59 % (transposed convs) of a workgroup's lifetime.  The fast path is the same data flow with
  // the per-element work stripped to the bone: whole float4 columns only (cout % 4 == 0), 32-bit byte offsets from
  // wave-uniform bases, one activation form per launch, no code for absent residual / multiplier operands.
  if (P.epi_fast) {
    const unsigned out_ldb = (unsigned)P.out_ld * 4u, res_ldb = (unsigned)P.res_ld * 4u;
    const char* res_b = reinterpret_cast<const char*>(res_img);
    char* out_b = reinterpret_cast<char*>(out_img);
    const bool has_res = XEPI == 1 ? false : (XEPI == 2 ? true : P.residual != nullptr), has_mul = XEPI ? false : P.pixmul != nullptr;
    const int act = P.act;
#pragma unroll 1
    for (int pass = 0; pass < NCP * NPP; ++pass) {
      const int cpass = pass % NCP, ppass = pass / NCP;
      const int nidx = n0 + cpass * EW + 4 * ej;
      int ch = nidx, bidx = nidx, sy = 0, sx = 0;
      if (P.store_mode == STORE_PIXSHUF) {
        const int q = nidx / P.cq; ch = nidx - q * P.cq; sy = q >> 1; sx = q & 1;
      } else if (P.store_mode == STORE_CONVT) {
        const int blk = nidx >> 7, q = (nidx & 127) >> 5; ch = blk * 32 + (nidx & 31); bidx = ch; sy = q >> 1; sx = q & 1;
      } else if (P.store_mode == STORE_ROWPAIR) {        // GEMM rows cq .. 2cq-1 are the same couts of the output row below
        sy = nidx >= P.cq ? 1 : 0; ch = nidx - sy * P.cq; bidx = ch;
      }
      const bool colok = nidx < P.cout;
      const bool up = P.store_mode != STORE_PLAIN;
      const bool upx = up && P.store_mode != STORE_ROWPAIR;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias && colok) b4 = *reinterpret_cast<const float4*>(P.bias + bidx);
      const unsigned chb = (unsigned)ch * 4u;
      unsigned opx[ITER];                       // pixel index inside the image, ~0u = nothing to store
      float4 rres[ITER];
      float rmul[ITER];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int p = ppass * 128 + ep0 + it * PSTEP;
        const int oy = oy0 + (p >> tw_lg), ox = ox0 + (p & tw_mask);
        const int Y = up ? 2 * oy + sy : oy, X = upx ? 2 * ox + sx : ox;
        const bool ok = colok && oy < P.oh && ox < P.ow && Y < P.OH;
        opx[it] = ok ? (unsigned)(Y * P.OW + X) : 0xFFFFFFFFu;
        if (has_res && ok) rres[it] = *reinterpret_cast<const float4*>(res_b + opx[it] * res_ldb + chb);
        if (has_mul && ok) rmul[it] = mul_img[opx[it]];
      }
      if (pass > 0) __syncthreads();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col0 = wn * WNT + nt * 32;
        if (col0 / EW != cpass) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int prow0 = wm * PM + mt * 32;
          if (prow0 / 128 != ppass) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            E[(prow0 - ppass * 128 + row) * EPIX + (col0 - cpass * EW) + li] = acc[mt][nt][r];
          }
        }
      }
      __syncthreads();
      const float* erow = E + ep0 * EPIX + 4 * ej;
      float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};       // GroupNorm partial sums of this thread's 4 channels
      float cab[2] = {0.f, 0.f}, caa[2] = {0.f, 0.f}, cbb[2] = {0.f, 0.f};    // patch-cosine partial sums, two patch columns
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        if (opx[it] == 0xFFFFFFFFu) continue;
        float4 v = *reinterpret_cast<const float4*>(erow + it * PSTEP * EPIX);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        if (XEPI == 1) {
          gs[0] += v.x; gs[1] += v.y; gs[2] += v.z; gs[3] += v.w;
          gq[0] = fmaf(v.x, v.x, gq[0]); gq[1] = fmaf(v.y, v.y, gq[1]); gq[2] = fmaf(v.z, v.z, gq[2]); gq[3] = fmaf(v.w, v.w, gq[3]);
        }
        if (act == GPEMSR_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (act == GPEMSR_ACT_LRELU) {          // max(v, 0.1 v) == (v > 0 ? v : 0.1 v)
          v.x = fmaxf(v.x, 0.1f * v.x); v.y = fmaxf(v.y, 0.1f * v.y); v.z = fmaxf(v.z, 0.1f * v.z); v.w = fmaxf(v.w, 0.1f * v.w);
        } else if (act != GPEMSR_ACT_NONE) {
          v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
        }
        if (XEPI == 2) {          // 64-column passes: this thread's pixel of item `it` lies in patch column it & 1 of the 32-pixel tile row
          const float4 r = rres[it];
          const int pc = it & 1;
          cab[pc] += (v.x * r.x + v.y * r.y) + (v.z * r.z + v.w * r.w);
          caa[pc] += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
          cbb[pc] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          continue;              // the feature map itself is not stored
        }
        if (has_res) { v.x += rres[it].x; v.y += rres[it].y; v.z += rres[it].z; v.w += rres[it].w; }
        if (has_mul) { v.x *= rmul[it]; v.y *= rmul[it]; v.z *= rmul[it]; v.w *= rmul[it]; }
        *reinterpret_cast<float4*>(out_b + opx[it] * out_ldb + chb) = v;
      }
      if (XEPI == 2) {
        // model/GPEMSR.py:387-395 without the second feature map in memory: sums of a.b, a.a, b.b over this 4-row strip of the two 16-pixel
        // patch columns the tile row crosses (b = this convolution's result, a = the `residual` operand); four strips make a patch
        // (gpemsr_patch_cosine_finish).  Fixed reduction tree; records numbered by strip and patch column, independent of the tile height.
        float* red = E + 128 * EPIX;
        float c6[6] = {cab[0], caa[0], cbb[0], cab[1], caa[1], cbb[1]};
#pragma unroll
        for (int k = 0; k < 6; ++k)
          for (int o = 1; o < 64; o <<= 1) c6[k] += __shfl_xor(c6[k], o);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) red[wave * 8 + k] = c6[k];
        }
        __syncthreads();
        if (tid < 6) {
          const float tot = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
          const int strip = ty * NPP + ppass, pcol = tx * 2 + tid / 3;
          P.cos_ws[(((long long)img * (P.tiles_y * NPP) + strip) * (P.tiles_x * 2) + pcol) * 4 + tid % 3] = tot;
        }
      }
      if (XEPI == 1) {
        // first pass of GroupNorm (model/blocks.py:5-6) from the accumulators: per (tile, pixel pass, channel) sum and sum of squares of
        // conv + bias.  Lanes with the same float4 column (lane % NV) are added by shuffles, the 4 waves through LDS behind the E tile;
        // fixed order -> bit-stable.  Readers of pass p and writers of pass p + 1 are separated by the barriers at the top of p + 1.
        float* red = E + 128 * EPIX;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          for (int o = NV; o < 64; o <<= 1) { gs[k] += __shfl_xor(gs[k], o); gq[k] += __shfl_xor(gq[k], o); }
        if (lane < NV) {
          float* r = red + (wave * NV + lane) * 8;
          *reinterpret_cast<float4*>(r) = make_float4(gs[0], gq[0], gs[1], gq[1]);
          *reinterpret_cast<float4*>(r + 4) = make_float4(gs[2], gq[2], gs[3], gq[3]);
        }
        __syncthreads();
        if (tid < NV) {
          float4 a = *reinterpret_cast<const float4*>(red + tid * 8), b = *reinterpret_cast<const float4*>(red + tid * 8 + 4);
#pragma unroll
          for (int wv = 1; wv < 4; ++wv) {
            const float4 a2 = *reinterpret_cast<const float4*>(red + (wv * NV + tid) * 8), b2 = *reinterpret_cast<const float4*>(red + (wv * NV + tid) * 8 + 4);
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
          }
          const int c0 = n0 + cpass * EW + 4 * tid;
          if (c0 < P.cout) {
            // canonical numbering by 128-pixel strip (4 tile rows), whatever the tile height: the records and the order gpemsr_groupnorm_finish
            // adds them in do not depend on the batch-size-dependent tile choice -> results are bit-identical across batch sizes
            const int part = (ty * NPP + ppass) * P.tiles_x + tx;
            float* wsp = P.gn_ws + (((long long)img * P.gn_parts + part) * P.cout + c0) * 2;
            *reinterpret_cast<float4*>(wsp) = a;
            *reinterpret_cast<float4*>(wsp + 4) = b;
          }
        }
      }
    }
    GP_ST(3);
    return;
  }
#pragma unroll 1
  for (int pass = 0; pass < NCP * NPP; ++pass) {
    const int cpass = pass % NCP, ppass = pass / NCP;
    // ---- (1) addresses of this pass's items; residual / multiplier loads are issued BEFORE the LDS round trip ----
    const int nidx = n0 + cpass * EW + 4 * ej;
    int ch = nidx, bidx = nidx, sy = 0, sx = 0;
    if (P.store_mode == STORE_PIXSHUF) {          // bias was permuted with the weight rows
      const int q = nidx / P.cq; ch = nidx - q * P.cq; sy = q >> 1; sx = q & 1;
    } else if (P.store_mode == STORE_CONVT) {     // bias is per true output channel
      const int blk = nidx >> 7, q = (nidx & 127) >> 5; ch = blk * 32 + (nidx & 31); bidx = ch; sy = q >> 1; sx = q & 1;
    }
    const int nvalid = (P.cout - nidx) < 4 ? (P.cout - nidx) : 4;      // <= 0: this thread's columns are past cout
    const bool full = nvalid == 4;
    // (the GENERAL path is the rare one -- ragged channel counts, unaligned rows, sigmoid: its residual / multiplier operands are loaded
    // where they are used; the eight float4 it used to prefetch per thread were what pushed the 168-register instantiations into scratch)
    int opix[ITER];                                                     // pixel index inside the image, or -1
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int p = ppass * 128 + ep0 + it * PSTEP;
      const int oy = oy0 + (p >> tw_lg), ox = ox0 + (p & tw_mask);
      opix[it] = -1;
      if (oy < P.oh && ox < P.ow && nvalid > 0) {
        const int Y = P.store_mode == STORE_PLAIN ? oy : 2 * oy + sy, X = P.store_mode == STORE_PLAIN ? ox : 2 * ox + sx;
        opix[it] = Y * P.OW + X;
      }
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < nvalid) bv[k] = P.bias[bidx + k];
    }
    if (pass == 0) GP_ST(4);
    // ---- (2) this pass's accumulator columns / pixel rows -> LDS ----
    if (pass > 0) __syncthreads();                       // previous pass fully read out
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col0 = wn * WNT + nt * 32;               // first column of this wave's N tile inside the block tile
      if (col0 / EW != cpass) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int prow0 = wm * PM + mt * 32;             // first pixel of this M tile inside the block tile
        if (prow0 / 128 != ppass) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          E[(prow0 - ppass * 128 + row) * EPIX + (col0 - cpass * EW) + li] = acc[mt][nt][r];
        }
      }
    }
    __syncthreads();
    if (pass == 0) GP_ST(5);
    // ---- (3) coalesced rows: bias, activation, residual, multiplier, store ----
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
      if (opix[it] < 0) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(E + (ep0 + it * PSTEP) * EPIX + 4 * ej);
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k] + bv[k], P.act);
      if (P.residual) {
        const float* rp = res_img + (long long)opix[it] * P.res_ld + ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nvalid) v[k] += rp[k];
      }
      if (P.pixmul) { const float m = mul_img[opix[it]]; v[0] *= m; v[1] *= m; v[2] *= m; v[3] *= m; }
      float* op = out_img + (long long)opix[it] * P.out_ld + ch;
      if (P.out_vec && full) {
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nvalid) op[k] = v[k];
      }
    }
    if (pass == 0) GP_ST(6);
  }
  GP_ST(3);
  }
}

template <int CK, int BN, int WM, int WN, int TH, bool MASKED, bool DMA, int XEPI = 0>
static int launch(const ConvParams& P, size_t lds_bytes, hipStream_t st) {
  auto kfn = conv_mfma_kernel<CK, BN, WM, WN, TH, MASKED, DMA, XEPI>;
  if (lds_bytes > 64 * 1024) {
    static dev_once_t attr_done{0};
    if (dev_once_begin(attr_done)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "conv2d: cannot raise the dynamic LDS limit");
      dev_once_done(attr_done);
    }
  }
  hipLaunchKernelGGL(kfn, dim3(P.nblocks), dim3(256), lds_bytes, st, P);
  return check_launch("conv_mfma_kernel");
}

}  // namespace gpemsr

using namespace gpemsr;
// tuning switches are read from the environment ONCE per process (this entry point runs 225 times per forward)
template <int SLOT>
static bool env_flag_once(const char* name) {
  static const bool set = getenv(name) != nullptr;
  return set;
}


struct ConvPlanName { char* buf; int cap; };
static int conv2d_impl(const gpemsr_conv_desc* d, void* stream, int* parts_only, ConvPlanName* name_only = nullptr);

extern "C" int gpemsr_conv2d(const gpemsr_conv_desc* d, void* stream) { return conv2d_impl(d, stream, nullptr); }

// the kernel instantiation gpemsr_conv2d would launch for `d`, as text (bench.py's per-kernel table); nothing is launched
extern "C" int gpemsr_conv2d_kernel_name(const gpemsr_conv_desc* d, char* buf, int cap) {
  GP_REQUIRE(buf != nullptr && cap > 0, "conv2d_kernel_name: no buffer");
  ConvPlanName nm{buf, cap};
  buf[0] = 0;
  return conv2d_impl(d, nullptr, nullptr, &nm);
}

// number of partial-sum records per image (`parts`) a launch of this descriptor leaves in d->gn_partials: [n][parts][cout][2]
extern "C" int gpemsr_conv2d_gn_parts(const gpemsr_conv_desc* d) {
  int parts = 0;
  const int rc = conv2d_impl(d, nullptr, &parts);
  return rc < 0 ? rc : parts;
}

static int conv2d_impl(const gpemsr_conv_desc* d, void* stream, int* parts_only, ConvPlanName* name_only) {
  GP_REQUIRE(d != nullptr, "conv2d: null descriptor");
  GP_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "conv2d: bad geometry n=%d h=%d w=%d cout=%d", d->n, d->h, d->w, d->cout);
  GP_REQUIRE(d->nsrc >= 1 && d->nsrc <= GPEMSR_MAX_SRC, "conv2d: nsrc=%d", d->nsrc);
  GP_REQUIRE(d->ksize == 1 || d->ksize == 3 || d->ksize == 7, "conv2d: ksize=%d unsupported", d->ksize);
  GP_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
  if (d->transposed == 3) {          // Winograd F(2x2, 3x3) form of a 3x3 stride-1 convolution (conv_wino.hip)
    return conv2d_winograd(d, stream, name_only ? name_only->buf : nullptr, name_only ? name_only->cap : 0, parts_only);
  }
  GP_REQUIRE(d->a_scale == nullptr || d->transposed == 5, "conv2d: a_scale / a_shift (folded source GroupNorm) exist in the F(4x4,3x3) form only");
  if (d->transposed == 5) {          // Winograd F(4x4, 3x3) form of a 3x3 stride-1 convolution with many input channels (conv_wino4.hip)
    return conv2d_winograd4(d, stream, name_only ? name_only->buf : nullptr, name_only ? name_only->cap : 0, parts_only);
  }
  if (d->transposed == 4) {          // 1-D Winograd F(2, 7) form of a 7x7 stride-1 convolution (conv7_wino.hip)
    GP_REQUIRE(parts_only == nullptr, "conv2d (F(2,7) form): no GroupNorm partial sums");
    return conv2d_winograd7(d, stream, name_only ? name_only->buf : nullptr, name_only ? name_only->cap : 0);
  }
  if (d->transposed == 6) {          // 2-D Winograd F(2x2, 7x7) form of a 7x7 stride-1 convolution (conv7_wino2d.hip)
    GP_REQUIRE(parts_only == nullptr, "conv2d (F(2x2,7x7) form): no GroupNorm partial sums");
    return conv2d_winograd77(d, stream, name_only ? name_only->buf : nullptr, name_only ? name_only->cap : 0);
  }
  ConvParams P{};
  const bool tr = d->transposed == 1;
  const bool rowpair = d->transposed == 2;
  GP_REQUIRE(d->transposed >= 0 && d->transposed <= 2, "conv2d: transposed=%d", d->transposed);
  if (rowpair) GP_REQUIRE(d->ksize == 7 && d->stride == 1 && d->cout == 16 && !d->pixel_shuffle,
                          "conv2d: the row-pair form is a 7x7 stride-1 convolution with 16 output channels");
  if (tr) GP_REQUIRE(d->ksize == 3 && !d->pixel_shuffle && d->cout % 32 == 0 && !d->pixmul,
                     "conv2d: transposed needs k=3, cout%%32==0, no pixel_shuffle/pixmul");
  else GP_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride=%d unsupported (use conv2d_direct)", d->stride);
  if (d->pixel_shuffle) GP_REQUIRE(d->cout % 16 == 0 && d->stride == 1, "conv2d: pixel_shuffle needs cout%%16==0, stride 1");
  const int CK = (d->ksize == 1) ? 32 : 8;
  int cin_pad = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].ld >= d->src[s].c, "conv2d: bad source %d", s);
    P.src[s] = d->src[s].ptr; P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    P.vec[s] = (d->src[s].ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0) && (P.img_stride[s] % 4 == 0);
    cin_pad += ((d->src[s].c + CK - 1) / CK) * CK;
  }
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0 && d->weight_image_stride % 4 == 0, "conv2d: weight must be 16B aligned");
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w;
  P.cin_pad = cin_pad;
  P.weight = d->weight; P.w_img_stride = d->weight_image_stride; P.bias = d->bias; P.act = d->act;
  P.residual = d->residual; P.res_ld = d->res_ld; P.pixmul = d->pixmul;
  P.out = d->out; P.out_ld = d->out_ld;
  P.out_vec = (d->out_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->out) & 15) == 0);
  P.res_vec = d->residual && (d->res_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->residual) & 15) == 0);
  int BN;
  if (tr) {
    // phase-stacked 2x2-tap form; weight = [4 taps (dy,dx)][4*Cout][cin_pad] (gpemsr_amd/packing.py::pack_convT)
    P.kh = P.kw = 2; P.stride = 1; P.stride_x = 1; P.pad = 0; P.cout = 4 * d->cout;
    P.oh = d->h; P.ow = d->w; P.OH = 2 * d->h; P.OW = 2 * d->w;
    P.store_mode = STORE_CONVT; P.cq = d->cout;
    P.tpg_h = 2; P.tpg_w = 2; P.ngroups = 1; BN = 128;
  } else if (rowpair) {
    // 16 couts fill half of a 32-row MFMA tile: rows 16..31 compute the SAME couts for the output row below -- an 8 x 7-tap
    // filter on every second row (weights shifted down one tap row for the second half: packing.pack_rowpair7), 56 / 49 of the
    // taps for half the pixel tiles
    P.kh = 8; P.kw = 7; P.stride = 2; P.stride_x = 1; P.pad = 3; P.cout = 32;
    P.oh = (d->h + 1) / 2; P.ow = d->w; P.OH = d->h; P.OW = d->w;
    P.store_mode = STORE_ROWPAIR; P.cq = 16;
    P.tpg_h = 1; P.tpg_w = 7; P.ngroups = 8; BN = 32;
  } else {
    P.kh = P.kw = d->ksize; P.stride = d->stride; P.stride_x = d->stride; P.pad = d->ksize / 2; P.cout = d->cout;
    P.oh = (d->h + 2 * P.pad - d->ksize) / P.stride + 1;
    P.ow = (d->w + 2 * P.pad - d->ksize) / P.stride + 1;
    P.store_mode = d->pixel_shuffle ? STORE_PIXSHUF : STORE_PLAIN; P.cq = d->cout / 4;
    P.OH = d->pixel_shuffle ? 2 * P.oh : P.oh; P.OW = d->pixel_shuffle ? 2 * P.ow : P.ow;
    BN = d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128);
    GP_REQUIRE(!(d->ksize == 7 && BN == 128), "conv2d: 7x7 with cout > 64 unsupported");
    if (d->ksize == 1) { P.tpg_h = 1; P.tpg_w = 1; P.ngroups = 1; }
    else { P.tpg_h = 1; P.tpg_w = d->ksize; P.ngroups = d->ksize; }   // one filter row per stage (small LDS images -> more blocks per CU)
  }
  // 8x32-pixel blocks (every wave owns 64 pixels x all couts) for stride-1 k>=3 convs with cout <= 64; 4x32 otherwise
  // transposed: 8x32 input pixels (two A tiles per wave reuse every weight fragment; 72 MFMAs per stage and barrier)
  int TH = (tr || (d->ksize == 3 && BN <= 64 && P.stride == 1)) ? 8 : 4;   // (7x7: the LDS images would allow 1 block/CU)
  // small launches (training crops): 8-row blocks would leave CUs idle, the 4x32 flavour doubles the block count
  if (!tr && TH == 8 && (long long)d->n * cdiv(P.oh, 8) * cdiv(P.ow, TILE_W) * cdiv(P.cout, BN) < 1024) TH = 4;
  // still fewer blocks than the chip has slots (3 per CU): split the output channels over narrower blocks (128 -> 64 -> 32)
  if (!tr && BN == 128 && d->ksize != 7 && !env_flag_once<0>("GPEMSR_CONV_NO_BNSPLIT") &&
      (long long)d->n * cdiv(P.oh, TH) * cdiv(P.ow, TILE_W) * cdiv(P.cout, BN) < 512) BN = 64;
  const bool want_gn = d->gn_partials != nullptr || parts_only != nullptr || d->cos_partials != nullptr;     // (64-column epilogue passes keep the summation order fixed)
  if (!tr && BN == 64 && !want_gn && !env_flag_once<1>("GPEMSR_CONV_NO_BN32") &&
      (long long)d->n * cdiv(P.oh, TH) * cdiv(P.ow, TILE_W) * cdiv(P.cout, BN) < 512) BN = 32;
  // narrow maps (training crops: 16x16 latents): 2*TH x 16-pixel tiles waste no columns; a 32-pixel MFMA tile is then two rows
  const int pad32 = cdiv(P.ow, 32) * 32, pad16 = cdiv(P.ow, 16) * 16;
  const int TW = (!tr && (pad32 - pad16) * 4 >= pad32 && !env_flag_once<2>("GPEMSR_CONV_NO_TW16")) ? 16 : TILE_W;   // >= 25 % fewer dead columns
  P.tw_lg = TW == 16 ? 4 : 5;
  const int TROWS = TH * TILE_W / TW;
  P.halo_h = (TROWS - 1) * P.stride + P.kh; P.halo_w = (TW - 1) * P.stride_x + P.kw;
  P.tiles_x = cdiv(P.ow, TW); P.tiles_y = cdiv(P.oh, TROWS); P.tiles_n = cdiv(P.cout, BN);
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d: grid too large");
  P.nblocks = (int)nb;
  P.epi_fast = P.out_vec && (P.cout % 4 == 0) && (!d->residual || P.res_vec) && (!d->bias || (reinterpret_cast<uintptr_t>(d->bias) & 15) == 0) &&
               ((long long)P.OH * P.OW * d->out_ld * 4 < (1ll << 32)) && (!d->residual || (long long)P.OH * P.OW * d->res_ld * 4 < (1ll << 32)) &&
               (P.store_mode != STORE_PIXSHUF || P.cq % 4 == 0);
  GP_REQUIRE(P.halo_h * P.halo_w * (CK / 4) <= A_LOADS * 256, "conv2d: halo too large");
  if (d->cos_partials) {
    GP_REQUIRE(!parts_only && !d->gn_partials && P.epi_fast && P.store_mode == STORE_PLAIN && !d->pixmul && d->residual && BN == 64 && P.tiles_n == 1 &&
               P.tw_lg == 5 && P.oh % 16 == 0 && P.ow % 32 == 0 && (reinterpret_cast<uintptr_t>(d->cos_partials) & 15) == 0,
               "conv2d: patch-cosine sums need 33..64 output channels, the operand in `residual`, height %% 16 == 0, width %% 32 == 0");
    P.cos_ws = d->cos_partials;
  }
  if (d->gn_partials || parts_only) {
    GP_REQUIRE(P.epi_fast && P.store_mode == STORE_PLAIN && d->act == GPEMSR_ACT_NONE && !d->residual && !d->pixmul && P.cout % 4 == 0,
               "conv2d: GroupNorm partial sums need the plain store, no activation / residual / multiplier, cout %% 4 == 0, aligned rows");
    P.gn_parts = P.tiles_y * P.tiles_x * (TH / 4);
    if (parts_only) { *parts_only = P.gn_parts; return 0; }
    GP_REQUIRE((reinterpret_cast<uintptr_t>(d->gn_partials) & 15) == 0, "conv2d: gn_partials must be 16-byte aligned");
    P.gn_ws = d->gn_partials;
  }
  if (rowpair) GP_REQUIRE(P.epi_fast, "conv2d: the row-pair form needs 16-byte aligned out / bias / residual rows");
  // DMA staging needs: every source 16-B aligned rows with c % CK == 0 (no partial chunks), 32-bit byte offsets
  bool dma = true;
  for (int s = 0; s < d->nsrc; ++s)
    dma = dma && P.vec[s] && (P.c[s] % CK == 0) && ((long long)d->h * d->w * P.ld[s] * 4 < (1ll << 32));
  dma = dma && ((long long)P.kh * P.kw * P.cout * P.cin_pad * 4 < (1ll << 32));
  const int tpg = P.tpg_h * P.tpg_w;
  P.na = cdiv((long long)P.halo_h * P.halo_w * (CK / 4), 256);
  P.nb = cdiv((long long)tpg * BN * (CK / 4), 256);
  if (dma) {
    P.a_buf_floats = P.na * 1024;           // linear images, padded to whole 1-KiB wave pieces
    P.b_buf_floats = P.nb * 1024;
  } else {
    const int APIX = CK == 8 ? 8 : CK + 4;
    P.a_buf_floats = (P.halo_h * P.halo_w * APIX + 3) & ~3;
    P.b_buf_floats = tpg * BN * APIX;
  }
  P.ring = dma ? 1 : 0;
  size_t lds_floats = 2 * (size_t)P.a_buf_floats + (size_t)(P.ring ? 3 : 2) * P.b_buf_floats;
  const size_t epi_floats = 128 * (size_t)((BN < 64 ? BN : 64) + 4) + ((P.gn_ws || P.cos_ws) ? 512 : 0);   // + the partial sums' 4-wave exchange
  if (epi_floats > lds_floats) lds_floats = epi_floats;
  const size_t lds = lds_floats * sizeof(float);
  GP_REQUIRE(lds <= 160 * 1024, "conv2d: LDS %zu too large", lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // (name_only: the same dispatch below, but the "launch" writes the instantiation's name)
#define GP_NAME(CKv, BNv, WMv, WNv, THv, MK, DMAv, XE) \
  (snprintf(name_only->buf, (size_t)name_only->cap, "conv_mfma_kernel<%d,%d,%d,%d,%d,%s,%s,%d>", CKv, BNv, WMv, WNv, THv, (MK) ? "true" : "false", (DMAv) ? "true" : "false", XE), 0)
#define GP_LAUNCH(CKv, BNv, WMv, WNv, THv, MK) \
  (name_only ? GP_NAME(CKv, BNv, WMv, WNv, THv, MK, dma, 0) : \
   (dma ? launch<CKv, BNv, WMv, WNv, THv, MK, true>(P, lds, st) : launch<CKv, BNv, WMv, WNv, THv, MK, false>(P, lds, st)))
#define GP_LAUNCH_X(CKv, BNv, WMv, WNv, THv) (name_only ? GP_NAME(CKv, BNv, WMv, WNv, THv, false, true, 1) : launch<CKv, BNv, WMv, WNv, THv, false, true, 1>(P, lds, st))
  if (P.gn_ws || P.cos_ws) {        // the partial-sum epilogue lives in its own instantiations (DMA staging only)
    GP_REQUIRE(dma && BN >= 32, "conv2d: partial sums need 16-byte aligned sources with channel counts that are multiples of the chunk");
    if (P.cos_ws) {                 // (3x3, 64-column blocks: checked above)
      GP_REQUIRE(CK == 8 && BN == 64, "conv2d: patch-cosine sums are a 3x3 form");
      if (name_only) return TH == 8 ? GP_NAME(8, 64, 4, 1, 8, false, true, 2) : GP_NAME(8, 64, 2, 2, 4, false, true, 2);
      return TH == 8 ? launch<8, 64, 4, 1, 8, false, true, 2>(P, lds, st) : launch<8, 64, 2, 2, 4, false, true, 2>(P, lds, st);
    }
    if (CK == 8) {
      if (BN == 32) return TH == 8 ? GP_LAUNCH_X(8, 32, 4, 1, 8) : GP_LAUNCH_X(8, 32, 4, 1, 4);
      if (BN == 64) return TH == 8 ? GP_LAUNCH_X(8, 64, 4, 1, 8) : GP_LAUNCH_X(8, 64, 2, 2, 4);
      return GP_LAUNCH_X(8, 128, 2, 2, 4);
    }
    if (BN == 32) return GP_LAUNCH_X(32, 32, 4, 1, 4);
    if (BN == 64) return GP_LAUNCH_X(32, 64, 2, 2, 4);
    return GP_LAUNCH_X(32, 128, 2, 2, 4);
  }
  if (tr) return GP_LAUNCH(8, 128, 4, 1, 8, true);
  if (CK == 8) {
    if (BN == 32) return TH == 8 ? GP_LAUNCH(8, 32, 4, 1, 8, false) : GP_LAUNCH(8, 32, 4, 1, 4, false);
    if (BN == 64) return TH == 8 ? GP_LAUNCH(8, 64, 4, 1, 8, false) : GP_LAUNCH(8, 64, 2, 2, 4, false);
    return GP_LAUNCH(8, 128, 2, 2, 4, false);
  }
  if (BN == 32) return GP_LAUNCH(32, 32, 4, 1, 4, false);
  if (BN == 64) return GP_LAUNCH(32, 64, 2, 2, 4, false);
  return GP_LAUNCH(32, 128, 2, 2, 4, false);
#undef GP_LAUNCH
#undef GP_LAUNCH_X
#undef GP_NAME
}

namespace gpemsr {
// four 4-row strips of partial sums -> cosine of one 16x16 patch (model/GPEMSR.py:392-395: F.normalize's eps 1e-12 on each norm)
__global__ __launch_bounds__(256) void patch_cosine_finish_kernel(const float* ws, int n, int ph, int pw, float* out) {
  const int total = n * ph * pw;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int px = e % pw, py = (e / pw) % ph, img = e / (pw * ph);
    float ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (((long long)img * (ph * 4) + py * 4 + s) * pw + px) * 4);
      ab += v.x; aa += v.y; bb += v.z;
    }
    out[e] = ab / (fmaxf(sqrtf(aa), 1e-12f) * fmaxf(sqrtf(bb), 1e-12f));
  }
}
}  // namespace gpemsr

extern "C" int gpemsr_patch_cosine_finish(const float* ws, int n, int ph, int pw, float* out, void* stream) {
  GP_REQUIRE(ws && out && n > 0 && ph > 0 && pw > 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "patch_cosine_finish: bad args");
  const int total = n * ph * pw;
  hipLaunchKernelGGL(gpemsr::patch_cosine_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ws, n, ph, pw, out);
  return check_launch("patch_cosine_finish");
}

#ifdef GP_STAMP
extern "C" int gpemsr_debug_read_stamps(unsigned long long* host, int nblocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gpemsr::g_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks);
}
#endif

// The Python binding (gpemsr_amd/_abi.py) mirrors this struct field by field.
static_assert(sizeof(gpemsr_conv_desc) == 248, "gpemsr_conv_desc layout changed: update gpemsr_amd/_abi.py");
