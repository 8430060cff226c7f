// Implicit-GEMM convolution with bf16 activations in HBM (precision = "bf16": BASELINE.json configs[2], "bf16 MFMA").
//
//   D[pixel][cout] = sum_tap sum_cin In[pixel*s + tap][cin] * W[tap][cout][cin]      on v_mfma_f32_32x32x16_bf16
//
// What differs from conv_split.hip (fp32 activations, converted in LDS on every read): the NHWC bf16 halo tile goes
// global -> LDS by LDS-DMA and IS the MFMA A image -- no raw buffer, no split pass, half the staged bytes -- and the
// epilogue stores bf16 (optionally fp32, optionally both), so a layer costs 2 B/element of HBM traffic each way.
//
// One 256-thread workgroup (4 waves, two workgroups per CU) = TH x 32 output pixels x BN couts.  K loop = (chunk of CK
// input channels) x (stage of TPS filter taps):
//   * A image of a chunk: [halo pixel][CK bf16] rows of R = CK/8 16-byte pieces, double-buffered by chunk.  A lane's
//     fragment (8 consecutive k of one pixel) is one ds_read_b128.  Rows are 64 B (CK = 32), so 16 consecutive pixels
//     would hit every 16-B bank slot four times: piece q of halo pixel hp is stored at physical piece
//     q ^ ((hp >> 2) & 3)  (CK = 16: q ^ ((hp >> 3) & 1)) -- conflict-free for any tap shift.  LDS-DMA writes lane-linear,
//     so the permutation is applied to the per-lane SOURCE address (guide rule 21); R consecutive lanes still fetch one
//     pixel's contiguous 16*R bytes;
//   * B image of a stage: [tap][piece][BN couts][8 bf16] -- weights are packed on the host in exactly the staged order
//     [chunk][tap][piece][cout][8] (gpemsr_amd/packing.py::pack_conv_bf16), so a DMA instruction reads 1 KiB of
//     consecutive global memory and lanes 0-31 of a fragment read 512 contiguous LDS bytes.  RING stages deep,
//     stage s + RING - 1 is issued at the top of stage s; every stage ends with a COUNTED s_waitcnt vmcnt(N) (N = DMA
//     instructions this wave issued after the ones stage s + 1 needs) + raw s_barrier, so RING - 2 stages of weights
//     (and the next chunk's A image) stay in flight across the barrier;
//   * GEMM form (1x1 / Linear / batched matrix products): no halo, the A tile [NPIX pixels][CK] rides in the same ring;
//     per-image B (attention) through w_img_stride;
//   * ConvTranspose2d(k3,s2,p1,op1): 2x2 taps, 4*Cout phase-stacked columns, per-tap N-tile masks (see conv_mfma.hip);
//   * stride 2: the A fragment walks every other halo pixel (2 rows x 32 px tiles keep the halo small);
//   * epilogue: accumulators -> LDS -> rows of 8 channels per thread (16-B bf16 stores): bias, activation, residual (bf16
//     or fp32), per-pixel multiplier, PixelShuffle / transposed-phase / B-operand ("kpack") store maps, optional fp32
//     copy (`out32`, the master copy of residual trunks) and optional GroupNorm partial sums per (tile, channel), which
//     replace the separate statistics pass over the tensor (R:model/blocks.py:5-6).
//
// Replaces, for bf16 tensors, the same ATen calls as gpemsr_conv2d (R:model/GPEMSR.py:323-456 and the modules it calls).
#include "common.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#ifdef GP16_STAMP
__device__ unsigned long long g_xstamps[8 * 65536];
#define XST(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) g_xstamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XST(i)
#endif

constexpr int XA_LOADS = 6;       // 16-B A slots per thread (halo_px * R / 256)
constexpr int XB_LOADS = 7;       // 16-B B slots per thread per stage (TPS * R * BN / 256; 7x7 row stage of 64 couts: 7)

enum { XS_PLAIN = 0, XS_PIXSHUF = 1, XS_CONVT = 2, XS_KPACK = 3 };

struct XParams {
  const unsigned short* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];      // elements
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w, oh, ow, OH, OW;
  int cout;                                  // GEMM N (4*Cout for the transposed form)
  int kw, kk, stride, pad;                   // filter width, taps per chunk, stride, padding
  const unsigned short* weight; long long w_img_stride;
  const float* bias; int act;
  const void* residual; int res_ld, res_f32;
  const float* pixmul;
  int store_mode, cq;
  void* out; int out_ld, out_f32;
  float* out32; int out32_ld;
  float* gn_ws; int gn_parts;                // [n][gn_parts = tiles per image][cout][2]
  long long kpack_img_stride;                // XS_KPACK: elements between images of the packed output
  int tiles_x, tiles_y, tiles_n;
  int halo_h, halo_w, halo_px;
  int tw_lg;
  int na, nb;                                // DMA slots per thread: A image, B stage image
  int a_bytes, b_bytes;                      // LDS bytes of one A image / one B stage image
  int ring;                                  // B (and, GEMM form, A) ring depth
  int n_abuf;                                // A images in LDS: GEMM: ring; conv: 2 (1 when there is a single chunk)
  int spc;                                   // stages per chunk = kk / TPS
  int nblocks;
};

__device__ __forceinline__ void xglds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

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

// CK: channels per chunk (32; 16 for sources that are odd multiples of 16).  TPS: taps per stage.  CONVT / GEMM: see above.
template <int CK, int BN, int WM, int WN, int TH, int TPS, bool CONVT, bool GEMM>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(XParams P) {
  constexpr int R = CK / 8;            // 16-byte pieces per pixel row
  constexpr int KS = CK / 16;          // MFMA k-steps per tap
  constexpr int NPIX = TH * 32;
  constexpr int PM = NPIX / WM;
  constexpr int MT = PM / 32;
  constexpr int WNT = BN / WN;
  constexpr int NT = WNT / 32;
  constexpr int ROWB = R * 16;         // bytes per pixel row of the A image
  constexpr int SWZ_SH = (R == 4) ? 2 : 3, SWZ_MK = R - 1;

  extern __shared__ __attribute__((aligned(16))) char xsm[];
  XST(0);
  const int n_abuf = P.n_abuf;
  char* const a_base = xsm;
  char* const b_base = xsm + n_abuf * P.a_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

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
  const int n0 = tn * BN;
  const int S = P.stride;
  // conv: tile rows oy0.., cols ox0..; GEMM: NPIX consecutive pixels of the flattened image starting at ox0 (oy0 = 0)
  const int oy0 = GEMM ? 0 : ty * (NPIX >> tw_lg), ox0 = GEMM ? tx * NPIX : (tx << tw_lg);
  const int iy0 = oy0 * S - P.pad, ix0 = ox0 * S - P.pad;
  const int hw_in = P.h * P.w;

  // ---- per-thread DMA slots ----
  int a_goff[XA_LOADS];        // element offset of the slot's 8 channels inside the source image, chunk base excluded; -1: none
#pragma unroll
  for (int i = 0; i < XA_LOADS; ++i) {
    const int e = tid + i * 256;
    a_goff[i] = -1;
    if (i < P.na && e < P.halo_px * R) {
      const int hp = e / R, pp = e % R;
      const int q = pp ^ ((hp >> SWZ_SH) & SWZ_MK);               // logical piece held at physical piece pp
      int pix = -1;
      if (GEMM) { const int p = ox0 + hp; if (p < hw_in) pix = p; }
      else {
        const int iy = iy0 + hp / P.halo_w, ix = ix0 + hp % P.halo_w;
        if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) pix = iy * P.w + ix;
      }
      if (pix >= 0) a_goff[i] = (pix << 8) | q;                   // pixel index (<= 2^23) and piece; the source ld is per chunk
    }
  }
  int b_goff[XB_LOADS];        // element offset inside the weight tensor relative to (chunk, first tap of the stage); -1: none
#pragma unroll
  for (int i = 0; i < XB_LOADS; ++i) {
    const int e = tid + i * 256;
    b_goff[i] = -1;
    if (i < P.nb && e < TPS * R * BN) {
      const int nn = e % BN, tq = e / BN;                         // LDS image [tap][piece][BN][8]
      if (n0 + nn < P.cout) b_goff[i] = (tq * P.cout + n0 + nn) * 8;
    }
  }
  int na_w = 0, nb_w = 0;
#pragma unroll
  for (int i = 0; i < XA_LOADS; ++i) na_w += (__ballot(a_goff[i] >= 0) != 0ull) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < XB_LOADS; ++i) nb_w += (__ballot(b_goff[i] >= 0) != 0ull) ? 1 : 0;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)xsm + (unsigned)wave * 1024u);

  // zero the slots no DMA ever writes (out-of-image halo pixels, rows past cout): once, in every buffer
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < XA_LOADS; ++i)
      if (i < P.na && (tid + i * 256) < P.halo_px * R && a_goff[i] < 0)
        for (int bsel = 0; bsel < n_abuf; ++bsel) *reinterpret_cast<float4*>(a_base + bsel * P.a_bytes + (tid + i * 256) * 16) = z;
#pragma unroll
    for (int i = 0; i < XB_LOADS; ++i)
      if (i < P.nb && (tid + i * 256) < TPS * R * BN && b_goff[i] < 0)
        for (int bsel = 0; bsel < P.ring; ++bsel) *reinterpret_cast<float4*>(b_base + bsel * P.b_bytes + (tid + i * 256) * 16) = z;
  }

  // ---- chunk / stage bookkeeping ----
  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / CK;
  const int spc = P.spc;
  const int nstages = nchunks * spc;
  const int RING = P.ring;

  int an_src = 0, an_c0 = 0, an_chunk = 0;                       // A cursor: next chunk image to issue
  int bn_chunk = 0, bn_grp = 0, bn_stage = 0;                    // B cursor: next stage to issue
  const unsigned short* wimg = P.weight + (long long)img * P.w_img_stride;

  auto issue_a = [&]() -> int {
    const unsigned short* sp = P.src[an_src] + (long long)img * P.img_stride[an_src] + an_c0;
    const unsigned pixb = (unsigned)P.ld[an_src] * 2u;
    const unsigned la = lds0 + (unsigned)((an_chunk % n_abuf) * P.a_bytes);
#pragma unroll
    for (int i = 0; i < XA_LOADS; ++i)
      if (a_goff[i] >= 0) xglds16((unsigned)(a_goff[i] >> 8) * pixb + 16u * (unsigned)(a_goff[i] & 255), sp, la + i * 4096u);
    ++an_chunk; an_c0 += CK;
    if (an_c0 >= P.c[an_src] && an_src + 1 < P.nsrc) { an_c0 = 0; ++an_src; }
    return na_w;
  };
  auto issue_b = [&]() -> int {
    const unsigned short* wp = wimg + ((long long)bn_chunk * P.kk + bn_grp * TPS) * (R * 8) * P.cout;
    const unsigned lb = lds0 + (unsigned)(n_abuf * P.a_bytes + (bn_stage % RING) * P.b_bytes);
#pragma unroll
    for (int i = 0; i < XB_LOADS; ++i)
      if (b_goff[i] >= 0) xglds16((unsigned)b_goff[i] * 2u, wp, lb + i * 4096u);
    ++bn_stage;
    if (++bn_grp == spc) { bn_grp = 0; ++bn_chunk; }
    return nb_w;
  };

  // ---- fragment addressing ----
  int hp0[MT];                       // halo pixel of this lane's A row at tap (0,0)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = wm * PM + mt * 32 + li;
    hp0[mt] = GEMM ? p : ((p >> tw_lg) * S) * P.halo_w + (p & tw_mask) * S;
  }
  const int b_frag = (wn * WNT + li) * 16 + lh * (BN * 16);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- prologue: A(0) [+ GEMM: A(1..RING-2)], B(0..RING-2) ----
  // issue log: iss[j] = DMA instructions this wave has issued in total once stage j's images are on their way.
  // A stage may be read after  issued_total - iss[stage]  instructions at most are still in flight (vmcnt is in order).
  int issued_total = 0;
  int iss_ring[8];                   // iss of stages s+1 .. (static indexing below: RING <= 8)
#pragma unroll
  for (int j = 0; j < 8; ++j) iss_ring[j] = 0;
  auto log_stage = [&](int stage) {  // called right after the issues that complete `stage`
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (stage & 7)) iss_ring[j] = issued_total;
  };
  auto iss_of = [&](int stage) -> int {
    int v = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (stage & 7)) v = iss_ring[j];
    return v;
  };
  if (GEMM) {
    for (int j = 0; j < RING - 1 && j < nstages; ++j) { issued_total += issue_a(); issued_total += issue_b(); log_stage(j); }
  } else {
    issued_total += issue_a();
    for (int j = 0; j < RING - 1 && j < nstages; ++j) { issued_total += issue_b(); log_stage(j); }
  }
  XST(1);
  xwait_vmcnt(issued_total - iss_of(0));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  XST(2);

  int chunk = 0, grp = 0;
  for (int stage = 0; stage < nstages; ++stage) {
    // top of the stage: everybody is past the barrier that ended stage-1, so ring slot (stage-1) % RING and the A buffer of
    // chunk-1 are free.  Issue order: next chunk's A image first, then the weights of stage + RING - 1.
    if (!GEMM && grp == 0 && an_chunk < nchunks && an_chunk == chunk + 1) issued_total += issue_a();
    if (stage + RING - 1 < nstages) {
      if (GEMM) issued_total += issue_a();
      issued_total += issue_b();
      log_stage(stage + RING - 1);
    }

    const char* A = a_base + (chunk % n_abuf) * P.a_bytes;
    const char* B = b_base + (stage % RING) * P.b_bytes + b_frag;
    const int tap0 = grp * TPS;
    bf16x8 fa[2][MT], fb[2][NT];
    auto tap_mask = [&](int tap) -> unsigned {       // CONVT: N tiles (phases q = 2py+px) fed by tap (dy,dx) = (tap>>1, tap&1)
      if (!CONVT) return 0xFu;
      return (tap >> 1) ? ((tap & 1) ? 0x8u : 0xCu) : ((tap & 1) ? 0xAu : 0xFu);
    };
    // step = (tap tt of the stage, k-step ks)
    auto load_step = [&](int set, int tt, int ks) {
      const int tap = tap0 + tt;
      const int ky = GEMM ? 0 : tap / P.kw, kx = GEMM ? 0 : tap - ky * P.kw;
      const int toff = ky * P.halo_w + kx;
      const unsigned mask = tap_mask(tap);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int hp = hp0[mt] + toff;
        const int pp = (2 * ks + lh) ^ ((hp >> SWZ_SH) & SWZ_MK);
        fa[set][mt] = *reinterpret_cast<const bf16x8*>(A + hp * ROWB + pp * 16);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (CONVT && !((mask >> nt) & 1u)) continue;
        fb[set][nt] = *reinterpret_cast<const bf16x8*>(B + (tt * R + 2 * ks) * (BN * 16) + nt * 512);
      }
    };
    auto mma_step = [&](int set, int tt) {
      const unsigned mask = tap_mask(tap0 + tt);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (CONVT && !((mask >> nt) & 1u)) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mt], fb[set][nt], acc[mt][nt], 0, 0, 0);
      }
    };
    constexpr int NSTEP = TPS * KS;
    load_step(0, 0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {            // software pipelined by two (static register sets)
      if (st + 1 < NSTEP) load_step((st + 1) & 1, (st + 1) / KS, (st + 1) % KS);
      mma_step(st & 1, st / KS);
    }

    // end of the stage: stage+1's images must have landed (this wave's pieces; the barrier covers the other waves') and
    // every wave must be done reading this stage's slot before the next top-of-stage overwrites it.
    if (stage + 1 < nstages) xwait_vmcnt(issued_total - iss_of(stage + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (++grp == spc) { grp = 0; ++chunk; }
  }

  XST(3);
  // ---- epilogue: accumulators -> LDS [128 px][EW + 4] fp32 -> rows of 8 channels per thread ----
  constexpr int EW = BN < 64 ? BN : 64;
  constexpr int EPIX = EW + 4;
  constexpr int NCP = BN / EW;
  constexpr int EP = NPIX < 128 ? NPIX : 128; // pixels per pass
  constexpr int NPP = NPIX / EP;
  constexpr int NV = EW / 8;                  // 8-channel groups per pixel per pass
  constexpr int ITER = (EP * NV) / 256;       // items per thread per pass (4 for 128 px x 64 columns)
  constexpr int PSTEP = 256 / NV;
  static_assert(ITER >= 1, "epilogue pass smaller than the workgroup");
  float* E = reinterpret_cast<float*>(xsm);
  const int ej = tid % NV, ep0 = tid / NV;
  const long long img_pix0 = (long long)img * P.OH * P.OW;
  const char* res_img = P.residual ? reinterpret_cast<const char*>(P.residual) + img_pix0 * P.res_ld * (P.res_f32 ? 4 : 2) : nullptr;
  const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
  char* out_img = reinterpret_cast<char*>(P.out) + (P.store_mode == XS_KPACK ? (long long)img * P.kpack_img_stride * 2
                                                                             : img_pix0 * P.out_ld * (P.out_f32 ? 4 : 2));
  float* out32_img = P.out32 ? P.out32 + img_pix0 * P.out32_ld : nullptr;
  const int tile_in_img = GEMM ? tx : ty * P.tiles_x + tx;
#pragma unroll 1
  for (int pass = 0; pass < NCP * NPP; ++pass) {
    const int cpass = pass % NCP, ppass = pass / NCP;
    const int nidx = n0 + cpass * EW + 8 * ej;
    int ch = nidx, bidx = nidx, sy = 0, sx = 0;
    if (P.store_mode == XS_PIXSHUF) { const int q = nidx / P.cq; ch = nidx - q * P.cq; sy = q >> 1; sx = q & 1; }
    else if (P.store_mode == XS_CONVT) { const int blk = nidx >> 7, q = (nidx & 127) >> 5; ch = blk * 32 + (nidx & 31); bidx = ch; sy = q >> 1; sx = q & 1; }
    const int nvalid = (P.cout - nidx) < 8 ? (P.cout - nidx) : 8;
    const bool full = nvalid == 8;
    const bool up = P.store_mode == XS_PIXSHUF || P.store_mode == XS_CONVT;
    int opix[ITER];
    float rres[ITER][8];
    float rmul[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int p = ppass * EP + ep0 + it * PSTEP;
      int oy, ox; bool ok;
      if (GEMM) { oy = 0; ox = ox0 + p; ok = ox < P.oh * P.ow; }
      else { oy = oy0 + (p >> tw_lg); ox = ox0 + (p & tw_mask); ok = oy < P.oh && ox < P.ow; }
      opix[it] = -1; rmul[it] = 1.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) rres[it][k] = 0.f;
      if (ok && nvalid > 0) {
        const int Y = up ? 2 * oy + sy : oy, X = up ? 2 * ox + sx : ox;
        opix[it] = GEMM ? ox : Y * P.OW + X;
        if (P.residual) {
          if (P.res_f32) {
            const float* rp = reinterpret_cast<const float*>(res_img) + (long long)opix[it] * P.res_ld + ch;
            if (full) {
              const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
              rres[it][0] = r0.x; rres[it][1] = r0.y; rres[it][2] = r0.z; rres[it][3] = r0.w;
              rres[it][4] = r1.x; rres[it][5] = r1.y; rres[it][6] = r1.z; rres[it][7] = r1.w;
            } else {
              for (int k = 0; k < nvalid; ++k) rres[it][k] = rp[k];
            }
          } else {
            const unsigned short* rp = reinterpret_cast<const unsigned short*>(res_img) + (long long)opix[it] * P.res_ld + ch;
            if (full) {
              const uint4 u = *reinterpret_cast<const uint4*>(rp);
              rres[it][0] = xbf_lo(u.x); rres[it][1] = xbf_hi(u.x); rres[it][2] = xbf_lo(u.y); rres[it][3] = xbf_hi(u.y);
              rres[it][4] = xbf_lo(u.z); rres[it][5] = xbf_hi(u.z); rres[it][6] = xbf_lo(u.w); rres[it][7] = xbf_hi(u.w);
            } else {
              for (int k = 0; k < nvalid; ++k) rres[it][k] = __uint_as_float((unsigned)rp[k] << 16);
            }
          }
        }
        if (P.pixmul) rmul[it] = mul_img[opix[it]];
      }
    }
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = (P.bias && k < nvalid) ? P.bias[bidx + k] : 0.f;
    __syncthreads();                            // main loop / previous pass done with the LDS
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col0 = wn * WNT + nt * 32;
      if (col0 / EW != cpass) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int prow0 = wm * PM + mt * 32;
        if (prow0 / EP != ppass) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          E[(prow0 - ppass * EP + row) * EPIX + (col0 - cpass * EW) + li] = acc[mt][nt][r];
        }
      }
    }
    __syncthreads();
    float gs[8], gq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gs[k] = 0.f; gq[k] = 0.f; }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (opix[it] < 0) continue;
      const float* er = E + (ep0 + it * PSTEP) * EPIX + 8 * ej;
      const float4 a0 = *reinterpret_cast<const float4*>(er), a1 = *reinterpret_cast<const float4*>(er + 4);
      float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] += bv[k];
        gs[k] += v[k]; gq[k] = fmaf(v[k], v[k], gq[k]);
        v[k] = apply_act(v[k], P.act);
        v[k] = (v[k] + rres[it][k]) * rmul[it];
      }
      if (out32_img) {
        float* o32 = out32_img + (long long)opix[it] * P.out32_ld + ch;
        if (full) {
          *reinterpret_cast<float4*>(o32) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          for (int k = 0; k < nvalid; ++k) o32[k] = v[k];
        }
      }
      if (P.out_f32) {
        float* op = reinterpret_cast<float*>(out_img) + (long long)opix[it] * P.out_ld + ch;
        if (full) {
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          for (int k = 0; k < nvalid; ++k) op[k] = v[k];
        }
      } else {
        const uint4 pk = make_uint4(xcvt_pk_bf16(v[0], v[1]), xcvt_pk_bf16(v[2], v[3]), xcvt_pk_bf16(v[4], v[5]), xcvt_pk_bf16(v[6], v[7]));
        unsigned short* op;
        if (P.store_mode == XS_KPACK) op = reinterpret_cast<unsigned short*>(out_img) + ((long long)(ch >> 3) * (P.OH * P.OW) + opix[it]) * 8;
        else op = reinterpret_cast<unsigned short*>(out_img) + (long long)opix[it] * P.out_ld + ch;
        if (full) {
          *reinterpret_cast<uint4*>(op) = pk;
        } else {
          const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
          for (int k = 0; k < nvalid; ++k) op[k] = (unsigned short)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu));
        }
      }
    }
    if (P.gn_ws) {
      // per-(tile, channel) sum / sum of squares of (conv + bias): threads with the same ej hold the same 8 channels
      __syncthreads();                          // everybody is done reading E
      float* G = E;                             // [PSTEP rows][EW channels][2]
#pragma unroll
      for (int k = 0; k < 8; ++k) { G[((ep0 * EW) + 8 * ej + k) * 2] = gs[k]; G[((ep0 * EW) + 8 * ej + k) * 2 + 1] = gq[k]; }
      __syncthreads();
      if (tid < EW * 2) {
        const int chn = tid >> 1, st = tid & 1;
        float s = 0.f;
        for (int r = 0; r < PSTEP; ++r) s += G[((r * EW) + chn) * 2 + st];         // fixed order
        const int co = n0 + cpass * EW + chn;
        if (co < P.cout) {
          float* wsp = P.gn_ws + (((long long)img * P.gn_parts + tile_in_img * NPP + ppass) * P.cout + co) * 2 + st;
          *wsp = s;
        }
      }
    }
  }
  XST(4);
}

template <int CK, int BN, int WM, int WN, int TH, int TPS, bool CONVT = false, bool GEMM = false>
static int launch_x(const XParams& P, size_t lds, hipStream_t st) {
  auto kfn = conv_bf16_kernel<CK, BN, WM, WN, TH, TPS, CONVT, GEMM>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "conv2d_bf16: cannot raise the dynamic LDS limit");
      done = true;
    }
  }
  hipLaunchKernelGGL(kfn, dim3(P.nblocks), dim3(256), lds, st, P);
  return check_launch("conv_bf16_kernel");
}

}  // namespace gpemsr

using namespace gpemsr;

namespace {
struct XPlan { int CK, BN, TH, TPS, WM, WN; bool tr, gemm; size_t lds; };

// geometry + tile choice of one launch (shared by the launcher and by gpemsr_conv2d_bf16_gn_parts)
int plan_x(const gpemsr_conv16_desc* d, XParams& P, XPlan& L) {
  GP_REQUIRE(d && d->weight && d->out, "conv2d_bf16: null pointer");
  const bool tr = d->transposed != 0;
  const bool gemm = !tr && d->ksize == 1;
  L.tr = tr; L.gemm = gemm;
  GP_REQUIRE(d->nsrc >= 1 && d->nsrc <= GPEMSR_MAX_SRC && d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "conv2d_bf16: bad geometry");
  if (tr) GP_REQUIRE(d->ksize == 3 && !d->pixel_shuffle && !d->pixmul && d->cout % 32 == 0 && !d->kpack, "conv2d_bf16: transposed needs k=3, cout%%32==0");
  else GP_REQUIRE((d->ksize == 1 || d->ksize == 3 || d->ksize == 7) && (d->stride == 1 || (d->stride == 2 && d->ksize == 3 && d->cout > 32)),
                  "conv2d_bf16: k in {1,3,7}, stride 1 (3x3 with cout > 32: 1 or 2)");
  GP_REQUIRE(d->weight_image_stride == 0 || gemm, "conv2d_bf16: per-image weights only for 1x1");
  if (d->pixel_shuffle) GP_REQUIRE(d->cout % 32 == 0 && d->ksize == 3 && d->stride == 1 && !d->kpack, "conv2d_bf16: pixel_shuffle needs k=3, cout%%32==0");
  if (d->kpack) GP_REQUIRE(!d->out_f32 && d->cout % 8 == 0, "conv2d_bf16: kpack output is bf16, cout%%8==0");
  bool ck32 = true;
  int nchunk_total = 0;
  for (int s = 0; s < d->nsrc; ++s) ck32 = ck32 && (d->src[s].c % 32 == 0);
  const int CK = ck32 ? 32 : 16;
  L.CK = CK;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].c % CK == 0 && d->src[s].ld % 8 == 0 && d->src[s].ld >= d->src[s].c &&
               ((reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0), "conv2d_bf16: source %d needs c%%16==0, ld%%8==0, 16-B alignment", s);
    P.src[s] = reinterpret_cast<const unsigned short*>(d->src[s].ptr); P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    GP_REQUIRE(P.img_stride[s] % 8 == 0 && (long long)d->h * d->w * P.ld[s] * 2 < (1ll << 32) && (long long)d->h * d->w < (1 << 23),
               "conv2d_bf16: source %d too large / misaligned", s);
    nchunk_total += d->src[s].c / CK;
  }
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0 && d->weight_image_stride % 8 == 0, "conv2d_bf16: weight alignment");
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w;
  P.weight = reinterpret_cast<const unsigned short*>(d->weight); P.w_img_stride = d->weight_image_stride;
  P.bias = d->bias; P.act = d->act; P.residual = d->residual; P.res_ld = d->res_ld; P.res_f32 = d->res_f32; P.pixmul = d->pixmul;
  P.out = d->out; P.out_ld = d->out_ld; P.out_f32 = d->out_f32; P.out32 = d->out32; P.out32_ld = d->out32_ld;
  P.gn_ws = d->gn_partials;
  int BN, TH, TPS, WM, WN;
  const int var = d->variant;                // 0 = default tile choice; > 0: alternatives (A/B tuning, scripts/conv16_microbench.py)
  if (tr) {
    P.kw = 2; P.kk = 4; P.stride = 1; P.pad = 0; P.cout = 4 * d->cout;
    P.oh = d->h; P.ow = d->w; P.OH = 2 * d->h; P.OW = 2 * d->w;
    P.store_mode = XS_CONVT; P.cq = d->cout; BN = 128; TH = 8; TPS = 2; WM = 4; WN = 1;
  } else {
    P.kw = d->ksize; P.kk = d->ksize * d->ksize; P.stride = d->stride; P.pad = d->ksize / 2; P.cout = d->cout;
    P.oh = (d->h + 2 * P.pad - d->ksize) / P.stride + 1; P.ow = (d->w + 2 * P.pad - d->ksize) / P.stride + 1;
    P.store_mode = d->pixel_shuffle ? XS_PIXSHUF : (d->kpack ? XS_KPACK : XS_PLAIN); P.cq = d->cout / 4;
    P.OH = d->pixel_shuffle ? 2 * P.oh : P.oh; P.OW = d->pixel_shuffle ? 2 * P.ow : P.ow;
    BN = d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128);
    if (gemm) { TH = 4; TPS = 1; WM = BN == 32 ? 4 : 2; WN = BN == 32 ? 1 : 2; }
    else if (d->ksize == 7) {
      // single-chunk layers (cin <= 32) keep ONE A image and a 2-deep ring of 7-tap row stages; wider inputs use 32-cout blocks
      TH = 4; TPS = 7;
      if (nchunk_total > 1 || BN == 32) { BN = 32; WM = 4; WN = 1; } else { BN = 64; WM = 2; WN = 2; }
    }
    else if (d->stride == 2) { TH = 2; WM = 2; WN = 2; TPS = (BN == 128) ? 1 : 3; }
    else if (BN == 128) {
      if (var == 1) { TH = 4; TPS = 3; WM = 2; WN = 2; }          // 4x32 px, row stages, 2-deep ring
      else if (var == 2) { TH = 8; TPS = 1; WM = 2; WN = 2; }     // 8x32 px, wave = 128 px x 64 couts
      else { TH = 8; TPS = 1; WM = 4; WN = 1; }                   // 8x32 px, wave = 64 px x 128 couts, tap stages
    }
    else { TH = 8; TPS = (var == 1) ? 1 : 3; WM = 4; WN = 1; }
  }
  L.BN = BN; L.TH = TH; L.TPS = TPS; L.WM = WM; L.WN = WN;
  GP_REQUIRE(!(P.store_mode == XS_PIXSHUF) || P.cq % 8 == 0, "conv2d_bf16: pixel_shuffle needs cout%%32==0");
  // 16-byte vector accesses are used by threads that own 8 valid channels; with cout < 8 every access is scalar
  const bool vec = d->cout >= 8;
  if (d->out_f32) GP_REQUIRE(!vec || (d->out_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 15) == 0), "conv2d_bf16: fp32 out alignment");
  else if (P.store_mode != XS_KPACK) GP_REQUIRE(!vec || (d->out_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 15) == 0), "conv2d_bf16: bf16 out needs ld%%8==0, 16-B alignment");
  else GP_REQUIRE((reinterpret_cast<uintptr_t>(d->out) & 15) == 0, "conv2d_bf16: kpack out alignment");
  if (d->residual) GP_REQUIRE(!vec || (d->res_ld % (d->res_f32 ? 4 : 8) == 0 && (reinterpret_cast<uintptr_t>(d->residual) & 15) == 0), "conv2d_bf16: residual alignment");
  if (d->out32) GP_REQUIRE(!vec || (d->out32_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->out32) & 15) == 0), "conv2d_bf16: out32 alignment");
  P.kpack_img_stride = (long long)P.oh * P.ow * d->cout;
  P.tw_lg = 5;
  const int NPIX = TH * 32;
  if (gemm) {
    P.halo_h = 1; P.halo_w = NPIX; P.halo_px = NPIX;
    P.tiles_x = cdiv((long long)P.oh * P.ow, NPIX); P.tiles_y = 1;
  } else {
    P.halo_h = (TH - 1) * P.stride + P.kw; P.halo_w = 31 * P.stride + P.kw; P.halo_px = P.halo_h * P.halo_w;
    P.tiles_x = cdiv(P.ow, 32); P.tiles_y = cdiv(P.oh, TH);
  }
  P.tiles_n = cdiv(P.cout, BN);
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d_bf16: grid too large");
  P.nblocks = (int)nb;
  P.gn_parts = P.tiles_y * P.tiles_x * (NPIX > 128 ? NPIX / 128 : 1);
  if (d->gn_partials) GP_REQUIRE(P.store_mode == XS_PLAIN && !tr, "conv2d_bf16: gn partial sums need the plain store");
  const int R = CK / 8;
  P.na = cdiv((long long)P.halo_px * R, 256);
  P.nb = cdiv((long long)TPS * R * BN, 256);
  GP_REQUIRE(P.na <= XA_LOADS && P.nb <= XB_LOADS, "conv2d_bf16: tile too large (na=%d nb=%d)", P.na, P.nb);
  P.a_bytes = P.halo_px * R * 16;                          // lanes past the image are masked: no padding to whole DMA pieces
  P.b_bytes = TPS * R * BN * 16;
  GP_REQUIRE(P.kk % TPS == 0, "conv2d_bf16: taps per stage must divide the tap count");
  P.spc = P.kk / TPS;
  // ring depth: as deep as 80 KiB per workgroup (two workgroups per CU) allows, at most spc + 1 (then the next chunk's A image,
  // issued at the chunk's first stage, is always older than the weights the end-of-stage wait retires) and at most 8
  const int budget = 80 * 1024;
  const int n_abuf_conv = nchunk_total > 1 ? 2 : 1;
  int ring = gemm ? 4 : 8;
  if (!gemm && ring > P.spc + 1) ring = P.spc + 1;
  if (ring > nchunk_total * P.spc) ring = nchunk_total * P.spc > 2 ? nchunk_total * P.spc : 2;
  if (ring < 2) ring = 2;
  while (ring > 2 && (gemm ? ring * (P.a_bytes + P.b_bytes) : n_abuf_conv * P.a_bytes + ring * P.b_bytes) > budget) --ring;
  P.ring = ring;
  P.n_abuf = gemm ? ring : n_abuf_conv;
  size_t lds = (size_t)P.n_abuf * P.a_bytes + (size_t)ring * P.b_bytes;
  const size_t epi = (size_t)(NPIX < 128 ? NPIX : 128) * (size_t)((BN < 64 ? BN : 64) + 4) * 4;
  if (epi > lds) lds = epi;
  GP_REQUIRE(lds <= 160 * 1024, "conv2d_bf16: LDS %zu too large", lds);
  L.lds = lds;
  return GPEMSR_OK;
}
}  // namespace

static_assert(sizeof(gpemsr_conv16_desc) == 240, "gpemsr_conv16_desc layout changed: update gpemsr_amd/_abi.py");

extern "C" int gpemsr_conv2d_bf16_gn_parts(const gpemsr_conv16_desc* d) {
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  return rc == GPEMSR_OK ? P.gn_parts : rc;
}

extern "C" int gpemsr_conv2d_bf16(const gpemsr_conv16_desc* d, void* stream) {
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  if (rc != GPEMSR_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = L.lds;
#define GP_X(BNv, WMv, WNv, THv, TPSv, TRv, GEMMv) \
  (L.CK == 32 ? launch_x<32, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv>(P, lds, st) : launch_x<16, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv>(P, lds, st))
#define GP_IS(BNv, WMv, WNv, THv, TPSv) (L.BN == BNv && L.WM == WMv && L.WN == WNv && L.TH == THv && L.TPS == TPSv)
  if (L.tr) return GP_X(128, 4, 1, 8, 2, true, false);
  if (L.gemm) {
    if (GP_IS(32, 4, 1, 4, 1)) return GP_X(32, 4, 1, 4, 1, false, true);
    if (GP_IS(64, 2, 2, 4, 1)) return GP_X(64, 2, 2, 4, 1, false, true);
    return GP_X(128, 2, 2, 4, 1, false, true);
  }
  if (GP_IS(32, 4, 1, 4, 7)) return GP_X(32, 4, 1, 4, 7, false, false);
  if (GP_IS(64, 2, 2, 4, 7)) return GP_X(64, 2, 2, 4, 7, false, false);
  if (GP_IS(64, 2, 2, 2, 3)) return GP_X(64, 2, 2, 2, 3, false, false);
  if (GP_IS(128, 2, 2, 2, 1)) return GP_X(128, 2, 2, 2, 1, false, false);
  if (GP_IS(32, 4, 1, 8, 3)) return GP_X(32, 4, 1, 8, 3, false, false);
  if (GP_IS(32, 4, 1, 8, 1)) return GP_X(32, 4, 1, 8, 1, false, false);
  if (GP_IS(64, 4, 1, 8, 3)) return GP_X(64, 4, 1, 8, 3, false, false);
  if (GP_IS(64, 4, 1, 8, 1)) return GP_X(64, 4, 1, 8, 1, false, false);
  if (GP_IS(128, 2, 2, 4, 3)) return GP_X(128, 2, 2, 4, 3, false, false);
  if (GP_IS(128, 2, 2, 8, 1)) return GP_X(128, 2, 2, 8, 1, false, false);
  if (GP_IS(128, 4, 1, 8, 1)) return GP_X(128, 4, 1, 8, 1, false, false);
  return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: no kernel for BN=%d WM=%d WN=%d TH=%d TPS=%d", L.BN, L.WM, L.WN, L.TH, L.TPS);
#undef GP_X
#undef GP_IS
}

#ifdef GP16_STAMP
extern "C" int gpemsr_debug_read_xstamps(unsigned long long* host, int nblocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gpemsr::g_xstamps), sizeof(unsigned long long) * 8 * (size_t)nblocks);
}
#endif
