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
#include "conv_bf16.h"

namespace gpemsr {

#ifdef GP16_STAMP
__device__ unsigned long long g_xstamps[8 * 65536];
#define XST(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) g_xstamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// segment accounting (wave 0 of every workgroup): XSEG(k) adds the cycles since the previous mark to bucket k
#define XSEG_DECL unsigned long long xs_t = __builtin_amdgcn_s_memtime(), xs_acc[6] = {0, 0, 0, 0, 0, 0}
#define XSEG(k) do { const unsigned long long xs_n = __builtin_amdgcn_s_memtime(); xs_acc[k] += xs_n - xs_t; xs_t = xs_n; } while (0)
#define XSEG_FLUSH do { if (threadIdx.x == 0 && blockIdx.x < 65536) { for (int k = 0; k < 6; ++k) g_xstamps[8 * blockIdx.x + k] = xs_acc[k]; g_xstamps[8 * blockIdx.x + 6] = T_me; } } while (0)
// resident2's loader wave 8 (thread 512): 0 issue | 1 wait for the DMA to land | 2 barrier; rows 1024 + blockIdx.x
#define R2L_DECL unsigned long long r2_t = __builtin_amdgcn_s_memtime(), r2_acc[3] = {0, 0, 0}
#define R2L(k) do { const unsigned long long r2_n = __builtin_amdgcn_s_memtime(); r2_acc[k] += r2_n - r2_t; r2_t = r2_n; } while (0)
#define R2L_FLUSH do { if (threadIdx.x == 512 && blockIdx.x < 1024) { for (int k = 0; k < 3; ++k) g_xstamps[8 * (1024 + blockIdx.x) + k] = r2_acc[k]; g_xstamps[8 * (1024 + blockIdx.x) + 6] = NI; } } while (0)
#else
#define XST(i)
#define XSEG_DECL
#define XSEG(k)
#define XSEG_FLUSH
#define R2L_DECL
#define R2L(k)
#define R2L_FLUSH
#endif

}  // namespace gpemsr
#include "conv_bf16_epi.h"
namespace gpemsr {


// conv64_resident2_kernel's k loop, software-pipelined BY HAND.  Only one wave of a SIMD multiplies at a time there, so nothing but the
// wave's own reads-ahead can cover the LDS latency; hipcc folds a source-level double buffer back into ONE fragment set and every
// MFMA then waits for a read issued one or two MFMAs earlier (stamps: 46 instead of ~32 clocks per MFMA, with or without
// sched_group_barrier).  All reads and MFMAs of a chunk are asm volatile (program order = issue order): the fragments of step st + 2
// are requested before the MFMAs of step st, three register sets rotate, and the waits count what may still be in flight.
template <int ST, typename TA>
__device__ __forceinline__ void r2_reads(bf16x8 (&fa)[2], bf16x8 (&fb)[2], const TA (&arow)[4][3], unsigned abase, unsigned bbase) {
  constexpr int tap = ST >> 1, ks = ST & 1;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const unsigned o = (unsigned)arow[mt + tap / 3][tap % 3];
    const unsigned ad = abase + (ks ? (o ^ 32u) : o);
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[mt]) : "v"(ad));
  }
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0]) : "v"(bbase), "n"((tap * 4 + 2 * ks) * 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1]) : "v"(bbase), "n"((tap * 4 + 2 * ks) * 1024 + 512));
}
template <int ST, typename TA>
__device__ __forceinline__ void r2_steps(f32x16 (&acc)[2][2], bf16x8 (&fa)[3][2], bf16x8 (&fb)[3][2], const TA (&arow)[4][3], unsigned abase, unsigned bbase) {
  if constexpr (ST < 18) {
    if constexpr (ST + 2 < 18) {
      r2_reads<ST + 2>(fa[(ST + 2) % 3], fb[(ST + 2) % 3], arow, abase, bbase);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else if constexpr (ST + 1 < 18) {
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    constexpr int c = ST % 3;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0][0]) : "v"(fb[c][0]), "v"(fa[c][0]));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[1][0]) : "v"(fb[c][0]), "v"(fa[c][1]));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0][1]) : "v"(fb[c][1]), "v"(fa[c][0]));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[1][1]) : "v"(fb[c][1]), "v"(fa[c][1]));
    r2_steps<ST + 1>(acc, fa, fb, arow, abase, bbase);
  }
}

// CK: channels per chunk (32; 16 for sources that are odd multiples of 16).  TPS: taps per stage.  CONVT / GEMM: see above.
// Filter width and stride follow from the instantiation: GEMM -> 1x1, CONVT -> 2x2 taps, TPS == 7 -> 7x7, TH == 2 -> 3x3 stride 2.
//
// PERSISTENT workgroups: a workgroup walks tiles  blockIdx.x, +gridDim.x, ...  and treats their chunks / stages as ONE stream:
// the weights of the next tile's first stages and its first halo image are already in flight while the current tile's last
// stages are multiplied and while its results are stored, so the per-tile prologue (DMA round trip) and epilogue overlap
// with matrix work instead of adding to it (in-kernel stamps of the first version: 8k cycles set-up + 20k cycles epilogue
// around 15k cycles of main loop on 64-channel layers).  The accumulators are kept TRANSPOSED (D^T = W . In^T: couts on
// the register axis, pixels on the lanes), so after bias / activation a pair of v_permlane32_swap gives every lane 8
// consecutive output channels of one pixel: the epilogue stores 16-byte pieces straight from registers -- no LDS staging,
// no barrier, each wave on its own.
//
// NL > 0: WAVE SPECIALISATION.  In-kernel stamps of the 8-wave big tile on a 256 -> 256 layer: 26 % of a tile in the MFMA
// stages, 30 % issuing refill DMA (the CU's global -> LDS path takes ~25-30 B/clk, so a 1-KiB instruction holds its wave for
// hundreds of cycles), 30 % waiting + barrier -- serialised, because every wave did every job and all waves are in the same
// phase.  With NL loader waves beside the WM x WN multiplying waves the DMA issue stalls and the counted waits sit on waves
// that have nothing else to do: per stage the multiplying waves run  MFMAs -> barrier , the loader waves  wait for stage
// s + 1 -> barrier -> refill the slot the barrier freed , and both meet at the one barrier per stage.
//
// AXF: the source gets a per-(image, channel) affine map + ReLU on its way into LDS -- GroupNorm's apply pass (R:model/blocks.py:5-29)
// folded into the convolution that consumes it, so the normalised tensor never exists in HBM (it cost one read + one write of every
// VQGAN block's first intermediate).  LDS-DMA cannot transform, so in this form the loader waves stage the halo image through
// REGISTERS: 16-byte global loads one chunk ahead (a whole stage of latency cover), 8 FMAs + max + 4 packs per piece, ds_write_b128
// into the same swizzled image the DMA form builds.  A loader thread keeps ONE logical 8-channel piece (dtid % R), so its 8 scales and
// shifts are loaded once per chunk; padding pixels are written as zeros (zero padding applies to the NORMALISED tensor).  Weights
// still arrive by LDS-DMA; the in-order vmcnt sees the register loads and the DMA instructions in one queue (all counted below).
// TCOMP (CONVT, TPS == 4, CK == 32): the stage image holds only the NINE non-zero (tap, phase) weight blocks of a chunk, tap by tap
// [tap][piece][rows_t][8] with rows_t = 128 / 64 / 64 / 32 (tap (0,0) feeds all four phases, (0,1) phases 1 and 3, (1,0) phases 2 and 3,
// (1,1) phase 3): 18 KB instead of the 32 KB of the phase-stacked form whose zero blocks were staged like the others (a stage of the
// transposed layers moved 51 KB by LDS-DMA for 18 MFMAs of a wave).  Weights: packing.pack_convT_bf16's compact form.
// XEPI == 1 (GEMM): the result is NOT stored; every wave leaves, per GEMM row, the maximum of (acc + bias) over its columns and the
// column where it is (lowest on ties) -- the arg-max of the codebook logits without the logits in memory (R:model/codebook.py:34-43).
template <int CK, int BN, int WM, int WN, int TH, int TPS, bool CONVT, bool GEMM, int NL = 0, bool LEAN = false, int SS = 0, bool AXF = false, bool TCOMP = false,
          int XEPI = 0>
__global__ __launch_bounds__((WM * WN + NL) * 64, (WM * WN + NL) > 12 ? 4 : ((WM * WN + NL) > 8 ? 3 : 2)) void conv_bf16_kernel(XParams P) {
  constexpr int NC = WM * WN;          // multiplying waves
  constexpr bool SPEC = NL > 0;
  constexpr int NTH = (NC + NL) * 64;  // 256 threads: two workgroups per CU; 512 threads ("big tile" forms): one
  constexpr int DTH = SPEC ? NL * 64 : NTH;   // threads that issue DMA
  constexpr int KW = GEMM ? 1 : (CONVT ? 2 : (TPS == 7 ? 7 : 3));
  constexpr int S = SS ? SS : ((!GEMM && !CONVT && TH == 2) ? 2 : 1);     // SS: explicit stride (the 4-row stride-2 loader-wave tile)
  constexpr int KK = KW * KW;
  constexpr int SPC = KK / TPS;        // stages per chunk
  static_assert(KK % TPS == 0, "taps per stage must divide the tap count");
  constexpr int R = CK / 8;            // 16-byte pieces per pixel row
  constexpr int KS = CK / 16;          // MFMA k-steps per tap
  constexpr int NPIX = TH * 32;
  constexpr int PM = NPIX / WM;
  constexpr int MT = PM / 32;
  constexpr int WNT = BN / WN;
  constexpr int NT = WNT / 32;
  constexpr int ROWB = R * 16;         // bytes per pixel row of the A image
  constexpr int SWZ_SH = (R == 8) ? 1 : ((R == 4) ? 2 : 3), SWZ_MK = R - 1;   // 256 B of rows share one bank sweep
  constexpr int HALO_W = GEMM ? NPIX : 31 * S + KW;
  constexpr int HALO_H = GEMM ? 1 : (TH - 1) * S + KW;
  constexpr int HALO_PX = HALO_W * HALO_H;
  constexpr int NA = (HALO_PX * R + DTH - 1) / DTH;   // A slots (16 B) per DMA thread
  static_assert(!TCOMP || (CONVT && TPS == 4 && CK == 32 && BN == 128 && WN == 1), "compact transposed weights: the 8x32 loader-wave tile");
  constexpr int B_SLOTS = TCOMP ? R * 288 : TPS * R * BN;   // 16-byte pieces of a stage's weight image
  constexpr int NB = (B_SLOTS + DTH - 1) / DTH;       // B slots per DMA thread per stage
  constexpr int A_BYTES = HALO_PX * R * 16;
  constexpr int B_BYTES = B_SLOTS * 16;
  static_assert(NA <= XA_LOADS && NB <= XB_LOADS, "tile too large");

  extern __shared__ __attribute__((aligned(16))) char xsm[];
  XSEG_DECL;
  const int n_abuf = P.n_abuf;
  const int RING = P.ring;             // <= 4
  char* const a_base = xsm;
  char* const b_base = xsm + n_abuf * A_BYTES;
  float* const bias_lds = reinterpret_cast<float*>(b_base + RING * B_BYTES);
  x_stage_bias(P, bias_lds, P.nbias, NTH);            // (published by the prologue barrier)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = !SPEC || wave >= NC, is_mul = !SPEC || wave < NC;     // wave-uniform roles
  const int dtid = SPEC ? tid - NC * 64 : tid, dwave = SPEC ? wave - NC : wave;  // DMA thread / wave index (loaders)
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int hw_in = P.h * P.w;

  // ---- sources (static copies: no dependent scalar loads in the loop) ----
  const unsigned short* srcp[GPEMSR_MAX_SRC];
  long long src_istride[GPEMSR_MAX_SRC];
  unsigned src_pixb[GPEMSR_MAX_SRC];
  int src_c[GPEMSR_MAX_SRC];
  int nchunks = 0;
#pragma unroll
  for (int s = 0; s < GPEMSR_MAX_SRC; ++s) {
    srcp[s] = P.src[s]; src_istride[s] = P.img_stride[s]; src_pixb[s] = (unsigned)P.ld[s] * 2u; src_c[s] = s < P.nsrc ? P.c[s] : 0;
    nchunks += src_c[s] / CK;
  }
  const int G = nchunks * SPC;                        // stages per tile

  // ---- slot geometry (recomputed where needed: registers are better spent on accumulators) ----
  // A slot i of this thread = 16-B piece e = tid + 256 i of the halo image: halo pixel hp = e / R holds, at physical piece
  // e % R, the logical piece (e % R) ^ swizzle(hp).  B slot i = piece e of the stage image [tap][piece][BN][8].
  auto a_slot_exists = [&](int i) -> bool { return dtid + i * DTH < HALO_PX * R; };
  auto a_slot_piece = [&](int i) -> unsigned { const int e = dtid + i * DTH, hp = e / R; return (unsigned)((e % R) ^ ((hp >> SWZ_SH) & SWZ_MK)); };
  // DMA instructions per image THIS WAVE issues for the weights (rows past cout are clamped, not masked: they only feed
  // accumulator rows that are never stored) -- wave-uniform and constant
  int nb_w = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) nb_w += (i * DTH + dwave * 64 < B_SLOTS) ? 1 : 0;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)xsm + (unsigned)dwave * 1024u);

  // ---- tiles of this workgroup ----
  const int ntiles = P.nblocks;                        // total tiles of the launch; gridDim.x <= ntiles
  const int grid = gridDim.x;
  const int T_me = (ntiles - (int)blockIdx.x + grid - 1) / grid;
  const int TS = T_me * G, TC = T_me * nchunks;        // stages / chunks of this workgroup's whole stream

  typedef XGeo Geo;
  auto tile_geo = [&](int ti) -> Geo {
    int t = (int)blockIdx.x + ti * grid;
    {   // XCD-aware remap (bijective): consecutive logical tiles of concurrently running workgroups share an XCD / L2
      const int q = ntiles / 8, r = ntiles % 8, xcd = t % 8;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t / 8;
    }
    int tn, tx, ty;
    xdivmod(t, P.tiles_n, P.mg_n, t, tn);
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    xdivmod(t, P.tiles_y, P.mg_y, t, ty);
    Geo g;
    g.img = t; g.n0 = tn * BN;
    g.oy0 = GEMM ? 0 : ty * TH; g.ox0 = GEMM ? tx * NPIX : tx * 32;
    g.tile_in_img = GEMM ? tx : ty * P.tiles_x + tx;
    return g;
  };
  // `cur` = the tile being multiplied, `nxt` = the one after it.  The DMA cursors run at most one tile ahead (host: ring <=
  // stages per tile, images <= chunks per tile) and copy what they need from `nxt` when they cross into it.
  Geo cur = tile_geo(0);
  Geo nxt = T_me > 1 ? tile_geo(1) : cur;
  int a_pix[NA];                                       // A cursor's tile: pixel index inside the source image per slot, -1: padding
  int na_w = 0, a_img = 0;                             // ... DMA instructions this wave issues per chunk image, image number
  bool a_pad = false;                                  // ... tile has padding slots (zero-fill needed)
  auto a_enter_tile = [&](const Geo& g) {
    const int iy0 = g.oy0 * S - P.pad, ix0 = g.ox0 * S - P.pad;
    int cnt = 0; bool pad = false;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int pix = -1;
      if (a_slot_exists(i)) {
        const int hp = (dtid + i * DTH) / R;
        if (GEMM) { const int pp = g.ox0 + hp; if (pp < hw_in) pix = pp; }
        else {
          const int iy = iy0 + hp / HALO_W, ix = ix0 + hp % HALO_W;
          if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) pix = iy * P.w + ix;
        }
        pad = pad || pix < 0;
      }
      a_pix[i] = pix;
      cnt += (__ballot(pix >= 0) != 0ull) ? 1 : 0;
    }
    na_w = cnt; a_img = g.img;
    a_pad = __ballot(pad) != 0ull;
  };
  int b_img = cur.img, b_n0 = cur.n0;                  // B cursor's tile
  // A cursor that has issued the last item of its tile crosses into the next one LAZILY, when it issues that tile's first
  // item: that moment always lies inside the main loop's current tile (ring <= stages per tile, images <= chunks per tile),
  // where `nxt` is the tile in question; the moment of the wrap itself may still lie in the tile before.
  bool a_cross = false, b_cross = false;

  // ---- DMA cursors over the whole stream ----
  int issued_total = 0;
  int markB[4] = {0, 0, 0, 0}, markA[4] = {0, 0, 0, 0};
  auto set4 = [&](int (&m)[4], int idx, int v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j == (idx & 3)) m[j] = v;
  };
  auto get4 = [&](const int (&m)[4], int idx) -> int {
    int v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j == (idx & 3)) v = m[j];
    return v;
  };
  int a_next = 0, a_ti = 0, a_chunk = 0, a_src = 0, a_c0 = 0;           // next chunk image to issue: stream index, tile, position
  int b_next = 0, b_ti = 0, b_chunk = 0, b_grp = 0;                      // next weight stage to issue
  int a_dst = 0, b_dst = 0;                                              // their destinations: a_next % n_abuf, b_next % RING
  // Issue of one image = begin (operands), one DMA instruction per slot, end (book-keeping).  In the main loop the slot
  // instructions are spread over the MFMA steps of the NEXT stage instead of being fired in one burst after the barrier:
  // an LDS-DMA instruction holds the issuing wave for 100-200 cycles while the CU's address path accepts it (in-kernel
  // stamps: a third of a wide layer's time was spent in that burst, with the matrix pipe idle), so it must go between MFMAs.
  const unsigned short* ia_sp = nullptr; unsigned ia_pixb = 0, ia_la = 0; int ia_cs = 0;
  auto issue_a_begin = [&]() {
    if (a_cross) { a_enter_tile(tile_geo(a_ti)); a_cross = false; }     // (the cursor may be TWO tiles ahead: single-chunk layers with two images)
    const unsigned short* sp = nullptr; unsigned pixb = 0; int cs = 0;
#pragma unroll
    for (int s = 0; s < GPEMSR_MAX_SRC; ++s)
      if (s == a_src) { sp = srcp[s] + (long long)a_img * src_istride[s] + a_c0; pixb = src_pixb[s]; cs = src_c[s]; }
    ia_la = xuni(lds0 + (unsigned)(a_dst * A_BYTES));
    ia_sp = reinterpret_cast<const unsigned short*>(xuni_ptr(sp));
    ia_pixb = xuni(pixb); ia_cs = cs;
    if (a_pad && a_chunk < n_abuf) {                    // padding slots of this tile: zero once per LDS image
      char* ab = a_base + a_dst * A_BYTES;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (a_slot_exists(i) && a_pix[i] < 0) *reinterpret_cast<float4*>(ab + (dtid + i * DTH) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto issue_a_slot = [&](int i) {
    if (a_pix[i] >= 0) xglds16((unsigned)a_pix[i] * ia_pixb + 16u * a_slot_piece(i), ia_sp, ia_la + i * (DTH * 16u));
  };
  auto issue_a_end = [&]() {
    issued_total += na_w;
    set4(markA, a_next, issued_total);
    if (++a_dst == n_abuf) a_dst = 0;
    ++a_next; a_c0 += CK;
    if (a_c0 >= ia_cs) { a_c0 = 0; ++a_src; }
    if (++a_chunk == nchunks) { a_chunk = 0; a_src = 0; a_c0 = 0; ++a_ti; a_cross = true; }
  };
  auto issue_a = [&]() {
    issue_a_begin();
#pragma unroll
    for (int i = 0; i < NA; ++i) issue_a_slot(i);
    issue_a_end();
  };
  const unsigned short* ib_wp = nullptr; unsigned ib_lb = 0;
  auto issue_b_begin = [&]() {
    if (b_cross) { const Geo gb = tile_geo(b_ti); b_img = gb.img; b_n0 = gb.n0; b_cross = false; }
    const unsigned short* wp = P.weight + (long long)b_img * P.w_img_stride + ((long long)b_chunk * KK + b_grp * TPS) * (R * 8) * P.cout;
    if (TCOMP) wp = P.weight + (long long)b_chunk * (B_SLOTS * 8) * (P.cout >> 7);        // [chunk][tap][piece][cout/32][rows_t][8]
    ib_wp = reinterpret_cast<const unsigned short*>(xuni_ptr(wp));
    ib_lb = xuni(lds0 + (unsigned)(n_abuf * A_BYTES + b_dst * B_BYTES));
  };
  auto issue_b_slot = [&](int i) {
    const int e = dtid + i * DTH;
    if (TCOMP) {
      if (e < B_SLOTS) {
        // tap t owns slots [512 t', ...): 4 pieces x rows_t rows; source = the same slot inside this cout block's share of the tap region
        const int t = e < 512 ? 0 : (e < 768 ? 1 : (e < 1024 ? 2 : 3));
        const int base = t == 0 ? 0 : (t == 1 ? 512 : (t == 2 ? 768 : 1024)), lg = t == 0 ? 7 : (t == 3 ? 5 : 6);
        const int piece = (e - base) >> lg, row = (e - base) & ((1 << lg) - 1);
        const int nblk = P.cout >> 7, blk = b_n0 >> 7;
        xglds16((unsigned)(base * nblk + ((piece * nblk + blk) << lg) + row) * 16u, ib_wp, ib_lb + i * (DTH * 16u));
      }
    } else if (e < TPS * R * BN) {
      int row = b_n0 + e % BN;
      row = row < P.cout ? row : P.cout - 1;
      xglds16((unsigned)((e / BN) * P.cout + row) * 16u, ib_wp, ib_lb + i * (DTH * 16u));
    }
  };
  auto issue_b_end = [&]() {
    issued_total += nb_w;
    set4(markB, b_next, issued_total);
    if (++b_dst == RING) b_dst = 0;
    ++b_next;
    if (++b_grp == SPC) {
      b_grp = 0;
      if (++b_chunk == nchunks) { b_chunk = 0; ++b_ti; b_cross = true; }
    }
  };
  auto issue_b = [&]() {
    issue_b_begin();
#pragma unroll
    for (int i = 0; i < NB; ++i) issue_b_slot(i);
    issue_b_end();
  };

  // ---- fragment addressing ----
  // PMC of the first versions: 8 VALU + 10 SALU instructions per MFMA, nearly all of it fragment address arithmetic redone for
  // every read -- the vector ALU was as busy as the matrix pipe.  The byte offsets are tile-invariant, so they are formed
  // ONCE: for filters up to 3x3 a table a_off[mt][tap] (k-step 0; k-step s > 0 flips piece bits: offset ^ (32 s)); the 7x7
  // filter (49 taps) keeps one filter row (= one stage) of offsets, rebuilt per stage.  Taps are compile-time (the stage loop
  // is unrolled over a chunk's stages), weight fragment offsets are immediates on one per-stage base.
  int hp0[MT];                       // halo pixel of this lane's A row at tap (0,0)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = wm * PM + mt * 32 + li;
    hp0[mt] = GEMM ? p : ((p >> 5) * S) * HALO_W + (p & 31) * S;
  }
  auto a_offset2 = [&](int mt, int ky, int kx) -> int {     // byte offset of k-step 0 of tap (ky,kx) inside a halo image
    const int hp = hp0[mt] + ky * HALO_W + kx;
    return hp * ROWB + ((lh ^ ((hp >> SWZ_SH) & SWZ_MK)) * 16);
  };
  // A wave's 32-pixel tiles are consecutive tile rows, so tap (ky, kx) of tile mt reads halo row  mt * S + ky : the table holds
  // one entry per (halo row, kx) -- (MT - 1) S + KW rows instead of MT * KW.
  constexpr bool TABLE = KK <= 9;
  constexpr int AR = !TABLE || GEMM ? MT : (MT - 1) * S + KW, AC = !TABLE ? TPS : (GEMM ? 1 : KW);
  int a_off[AR][AC];                 // TABLE: (re)built at the top of every tile, so it is not live across the epilogue
  const int b_frag = TCOMP ? li * 16 : (wn * WNT + li) * 16 + lh * (BN * 16);     // TCOMP: the piece stride depends on the tap (load_step)
  const unsigned xsm_lds = xlds_addr(xsm);

  if constexpr (AXF) {
    static_assert(SPEC && !GEMM && !CONVT && SPC == 1 && DTH % R == 0 && R == 4, "AXF: loader-wave 3x3 form with whole-chunk stages");
    if (!is_mul) {
      // ---- loader waves, register-staged + transformed halo images (see AXF above); host: one source, n_abuf == 2, RING <= stages ----
      const int q = dtid % R;                            // this thread's logical 16-byte piece (8 channels) of every halo pixel
      uint4 raw[NA];
      float sc[8], sh[8];
      unsigned padmask = 0u;                              // bit i: slot i of the image held in `raw` is padding (or does not exist)
      const float lo = P.ax_relu ? 0.f : -3.0e38f;
      auto axf_load = [&]() {                             // halo image of the A cursor's chunk -> registers; NA + 4 vector-memory operations
        if (a_cross) { a_enter_tile(nxt); a_cross = false; }
        const unsigned short* sp = srcp[0] + (long long)a_img * src_istride[0] + a_c0 + 8 * q;
        const float* scp = P.axs + (long long)a_img * src_c[0] + a_c0 + 8 * q;
        const float* shp = P.axh + (long long)a_img * src_c[0] + a_c0 + 8 * q;
        unsigned pm = 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int pix = a_pix[i];
          pm |= (pix < 0 ? 1u : 0u) << i;
          raw[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sp) + (size_t)(unsigned)(pix < 0 ? 0 : pix) * src_pixb[0]);
        }
        padmask = pm;
        const float4 s0 = *reinterpret_cast<const float4*>(scp), s1 = *reinterpret_cast<const float4*>(scp + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(shp), h1 = *reinterpret_cast<const float4*>(shp + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        issued_total += NA + 4;
        ++a_next; a_c0 += CK;
        if (++a_chunk == nchunks) { a_chunk = 0; a_c0 = 0; ++a_ti; a_cross = true; }
      };
      auto axf_write = [&](int buf) {                     // registers -> transformed image `buf` (every existing slot is written)
        char* ab = a_base + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int e = dtid + i * DTH;
          if (e < HALO_PX * R) {
            const int hp = e / R;
            const unsigned in[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float a = fmaxf(fmaf(__uint_as_float(in[k] << 16), sc[2 * k], sh[2 * k]), lo);
              const float b = fmaxf(fmaf(__uint_as_float(in[k] & 0xFFFF0000u), sc[2 * k + 1], sh[2 * k + 1]), lo);
              o[k] = xcvt_pk_bf16(a, b);
            }
            const bool pad = (padmask >> i) & 1u;
            *reinterpret_cast<uint4*>(ab + hp * ROWB + ((q ^ ((hp >> SWZ_SH) & SWZ_MK)) * 16)) =
                pad ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(o[0], o[1], o[2], o[3]);
          }
        }
      };
      // prologue: A(0) -> image 0, A(1) -> registers, weights of the first RING stages
      a_enter_tile(cur);
      axf_load();
      axf_write(0);
      if (a_next < TC) axf_load();
      for (int j = 0; j < RING && b_next < TS; ++j) issue_b();
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int wbuf = 1;                                       // image that receives the chunk held in registers
      int s_glob = 0;
      for (int ti = 0; ti < T_me; ++ti) {
        for (int cc = 0; cc < nchunks; ++cc, ++s_glob) {
          // concurrent with the multiplying waves' stage s_glob: image (s+1) % 2 and ring slot (s-1) % RING were freed by the last barrier
          if (s_glob + 1 < TC) axf_write(wbuf);           // A(s+1), loaded a stage ago (the compiler's wait covers exactly those loads:
          wbuf ^= 1;                                      //  everything older -- the DMA below is issued BEFORE the next loads -- has landed)
          if (s_glob > 0 && b_next < TS) issue_b();       // B(s-1+RING)
          if (a_next < TC) axf_load();                    // A(s+2) -> registers
          if (s_glob + 1 < TS) xwait_vmcnt(issued_total - get4(markB, s_glob + 1));    // B(s+1) has landed
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        cur = nxt;
        if (ti + 2 < T_me) nxt = tile_geo(ti + 2);
      }
      return;
    }
    // the loader waves of this form issue ~400 vector instructions per stage: the multiplying waves that share their SIMDs get the
    // issue slots first (MI355X_MICROARCH.md, two waves per SIMD, item 2: arbitration by priority, then age)
    __builtin_amdgcn_s_setprio(2);
  }
  // ---- prologue: tile 0 (and 1) geometry, A(0) [A(1)], B(0 .. RING-1) ----
  if (is_loader) {
    a_enter_tile(cur);
    issue_a();
    issue_b();
    for (int j = 1; j < n_abuf && a_next < TC; ++j) issue_a();
    for (int j = 1; j < RING && b_next < TS; ++j) issue_b();
  }
  XSEG(0);
  if (is_loader) {
    const int ma = get4(markA, 0), mb = get4(markB, 0);
    xwait_vmcnt(issued_total - (ma > mb ? ma : mb));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  XSEG(1);

  int gs = 0, gc = 0;               // stream position: stage, chunk
  int a_slot = 0, b_slot = 0;       // LDS image / ring slot of the current chunk / stage (gc % n_abuf, gs % RING)

  // ---- end of a stage: DMA == true on the waves that load (all of them without specialisation) ----
  auto stage_end = [&](const bool chunk_end, const bool DMA) {
    if (DMA && gs + 1 < TS) {      // the next stage's weights (and, at a chunk boundary, the next chunk's halo image) must have landed
      int need = get4(markB, gs + 1);
      if (chunk_end && n_abuf >= 2) { const int ma = get4(markA, gc + 1); need = ma > need ? ma : need; }
      xwait_vmcnt(issued_total - need);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // every wave is done with this stage's ring slot (and, at a chunk end, with the A image)
    asm volatile("" ::: "memory");
    XSEG(3);
    // the freed slot / image are refilled at once: weights of stage gs + RING, halo image of chunk gc + n_abuf.  (Spreading
    // these DMA instructions over the next stage's MFMA steps was tried and lost 2-10 %: a DMA instruction stalls the wave
    // in front of its own MFMAs -- hence the loader waves of the specialised form.)
    if (DMA && b_next < TS) issue_b();
    if (++b_slot == RING) b_slot = 0;
    if (chunk_end) {
      if (DMA && n_abuf >= 2 && a_next < TC) issue_a();
      ++gc;
      if (++a_slot == n_abuf) a_slot = 0;
      if (n_abuf < 2 && gc < TC) {       // single A image: needed by the very next stage -- issue now and wait for it
        if (DMA) {
          issue_a();
          xwait_vmcnt(issued_total - get4(markA, a_next - 1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    XSEG(4);
    ++gs;
  };

  if (SPEC && !is_mul) {
    // ---- loader waves: the same stage / barrier sequence, no arithmetic ----
    for (int ti = 0; ti < T_me; ++ti) {
      for (int cc = 0; cc < nchunks; ++cc) {
#pragma unroll 1
        for (int grp = 0; grp < SPC; ++grp) stage_end(grp == SPC - 1, true);
      }
      cur = nxt;
      if (ti + 2 < T_me) nxt = tile_geo(ti + 2);
    }
    return;
  }

  for (int ti = 0; ti < T_me; ++ti) {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    if (TABLE) {
      int li3 = li, lh3 = lh;           // opaque: the table is tile-invariant, but kept across the epilogue it costs 12 registers of the 168
      asm volatile("" : "+v"(li3), "+v"(lh3));
      const int p0 = wm * PM + li3;
#pragma unroll
      for (int rr = 0; rr < AR; ++rr)
#pragma unroll
        for (int kx = 0; kx < AC; ++kx) {
          const int hp = GEMM ? p0 + rr * 32 : ((p0 >> 5) * S + rr) * HALO_W + (p0 & 31) * S + kx;
          a_off[rr][kx] = hp * ROWB + ((lh3 ^ ((hp >> SWZ_SH) & SWZ_MK)) * 16) + a_slot * A_BYTES;
        }
    }

    auto stage = [&](const int grp) {                  // grp is a compile-time constant in the unrolled (TABLE) form
      // TABLE: a_off already points into the current halo image (moved once per chunk, below) -- adding the image base per
      // read gave the compiler 2 x 18 more loop invariants to keep (and spill) on the 168-register budget of the 12-wave form
      const unsigned A = xsm_lds + (TABLE ? 0u : (unsigned)(a_slot * A_BYTES));
      const unsigned B = xsm_lds + (unsigned)(n_abuf * A_BYTES + b_slot * B_BYTES + b_frag);
      const int tap0 = grp * TPS;
      // (The hand-pipelined asm loop of the weights-resident kernel -- r2_steps, same fragment geometry -- was tried here for the wide
      // 3x3 tile: the step went from 109.3 to 115.9 ms.  With two multiplying waves per SIMD the other wave already covers the LDS
      // latency, and three fragment sets do not fit this kernel's 168-register budget beside its address tables.)
      bf16x8 fa[2][MT], fb[2][NT];
      if (!TABLE) {                                    // 7x7: a stage is filter row ky = grp
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int tt = 0; tt < TPS; ++tt) a_off[mt][tt] = a_offset2(mt, grp, tt);
      }
      auto tap_mask = [&](int tap) -> unsigned {       // CONVT: N tiles (phases q = 2py+px) fed by tap (dy,dx) = (tap>>1, tap&1)
        if (!CONVT) return 0xFu;
        return (tap >> 1) ? ((tap & 1) ? 0x8u : 0xCu) : ((tap & 1) ? 0xAu : 0xFu);
      };
      auto load_step = [&](int set, int tt, int ks) {
        const unsigned mask = tap_mask(tap0 + tt);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int tap = tap0 + tt;
          const int o = !TABLE ? a_off[mt][tt] : (GEMM ? a_off[mt][0] : a_off[mt * S + tap / KW][tap % KW]);
          fa[set][mt] = xlds_read16(A + (unsigned)(ks ? (o ^ (32 * ks)) : o));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (CONVT && !((mask >> nt) & 1u)) continue;
          if (TCOMP) {
            // tap tt: rows_t rows per piece, the fed phases in increasing order -> slot of phase nt inside the tap's image
            const int tap = tap0 + tt;
            const int rt = tap == 0 ? 128 : (tap == 3 ? 32 : 64), tb = tap == 0 ? 0 : (tap == 1 ? 512 : (tap == 2 ? 768 : 1024));
            const int slot = tap == 0 ? nt : (tap == 1 ? (nt - 1) / 2 : (tap == 2 ? nt - 2 : 0));
            fb[set][nt] = xlds_read16(B + (unsigned)(tb * 16 + 2 * ks * rt * 16 + slot * 512) + (unsigned)(lh * (rt * 16)));
          } else {
            fb[set][nt] = xlds_read16(B + (unsigned)((tt * R + 2 * ks) * (BN * 16) + nt * 512));
          }
        }
      };
      auto mma_step = [&](int set, int tt) {
        const unsigned mask = tap_mask(tap0 + tt);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (CONVT && !((mask >> nt) & 1u)) continue;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)       // D^T: weights are the row operand, pixels the column operand
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][nt], fa[set][mt], acc[mt][nt], 0, 0, 0);
        }
      };
      constexpr int NSTEP = TPS * KS;
      load_step(0, 0, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {            // software pipelined by two (static register sets)
        if (st + 1 < NSTEP) load_step((st + 1) & 1, (st + 1) / KS, (st + 1) % KS);
        mma_step(st & 1, st / KS);
      }
      XSEG(2);
      const int a_before = a_slot;
      stage_end(grp == SPC - 1, !SPEC);               // (compile-time after unrolling)
      if (TABLE && grp == SPC - 1 && a_slot != a_before) {
        const int delta = (a_slot - a_before) * A_BYTES;
#pragma unroll
        for (int rr = 0; rr < AR; ++rr)
#pragma unroll
          for (int kx = 0; kx < AC; ++kx) a_off[rr][kx] += delta;
      }
    };
    for (int cc = 0; cc < nchunks; ++cc) {
      if (TABLE) {
#pragma unroll
        for (int grp = 0; grp < SPC; ++grp) stage(grp);      // unrolled: taps, tap masks, table indices are compile-time
      } else {
#pragma unroll 1
        for (int grp = 0; grp < SPC; ++grp) stage(grp);
      }
    }

    // both DMA cursors have crossed into the next tile by now: advance the tile window
    const Geo g = cur;
    cur = nxt;
    if (ti + 2 < T_me) nxt = tile_geo(ti + 2);

    // ---- epilogue, straight from the accumulators (D^T: lane = pixel, registers = couts) ----
    {
      int li2 = li, lh2 = lh;           // opaque per-tile copies: keeps the epilogue's tile-invariant addressing out of the MFMA loop's live set
      asm volatile("" : "+v"(li2), "+v"(lh2));
      if constexpr (XEPI == 1) x_epilogue_rowmax<MT, NT>(P, g, acc, wm * PM, wn * WNT, wn, WN, bias_lds, li2, lh2);
      else x_epilogue<MT, NT, GEMM, CONVT, LEAN>(P, g, acc, wm * PM, wn * WNT, g.tile_in_img * WM + wm, bias_lds, li2, lh2);
    }
    XSEG(5);
  }
  XSEG_FLUSH;
}


// ---------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with 64 input channels -- the most common layer of the network (feature extraction, fusion
// blocks, reconstruction trunk, up-convs, HRconv, the 64-channel VQGAN blocks) -- with the WEIGHTS RESIDENT IN LDS.
// In-kernel stamps of the ring kernel above show those layers bound by global -> LDS staging, not by the matrix pipe: a CU
// stages ~25-30 B/clk, and with K = 576 the weights (73.7 KB per 64 couts) were re-staged for every 256-pixel tile.  Here a
// 512-thread workgroup (8 waves, one per CU) keeps the [2 chunks][9 taps][4 pieces][64 couts][8] weight slab of ITS 64 couts
// in LDS for its whole life and streams only halo images: tile = 16 x 32 pixels, every wave owns two pixel rows x all 64
// couts (64 accumulator registers), the halo image of a 32-channel chunk (18 x 34 pixels, 39 KB) is double buffered, so
// chunk c + 1 (or the next tile's chunk 0) lands while chunk c is multiplied: 72 MFMAs per wave between barriers, two
// barriers per tile, no DMA accounting beyond vmcnt(0) (each image was issued a whole phase earlier).
// Staged bytes per FLOP drop 2.7x against the ring kernel (78 KB per 37.7 MFLOP instead of 117 KB per 18.9).
// ---------------------------------------------------------------------------------------------------------------------
template <bool LEAN>
__global__ __launch_bounds__(512, 2) void conv64_resident_kernel(XParams P) {
  constexpr int TH = 16, HALO_W = 34, HALO_H = 18, HALO_PX = HALO_W * HALO_H, R = 4;
  constexpr int A_BYTES = HALO_PX * R * 16;            // 39,168
  constexpr int W_BYTES = 2 * 9 * 4 * 64 * 16;         // 73,728
  constexpr int NA = (HALO_PX * R + 511) / 512;        // 5 slots per thread per chunk image
  constexpr int NW = W_BYTES / 16 / 512;               // 9 slots per thread for the weight slab
  constexpr int MT = 2, NT = 2;
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  char* const w_base = xsm;
  char* const a_base = xsm + W_BYTES;
  float* const bias_lds = reinterpret_cast<float*>(xsm + W_BYTES + 2 * A_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tn = (int)blockIdx.x % P.tiles_n, sg = (int)blockIdx.x / P.tiles_n;
  const int n0 = tn * 64;
  const int gpt = P.gpt, NS = P.ns;
  const int T_me = (NS - sg + gpt - 1) / gpt;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)xsm + (unsigned)wave * 1024u);

  x_stage_bias(P, bias_lds, P.nbias, 512);
  // weight slab of this workgroup: rows n0 .. n0+63 of every (chunk, tap, piece) -- 72 runs of 1 KiB (rows past cout clamped)
  {
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.weight));
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 512;                      // 16-B piece: (chunk*9 + tap)*4 + piece = e / 64, cout row = e % 64
      int row = n0 + (e & 63);
      row = row < P.cout ? row : P.cout - 1;
      xglds16((unsigned)((e >> 6) * P.cout + row) * 16u, wp, lds0 + i * 8192u);
    }
  }

  XGeo g;
  int a_pix[NA];
  bool a_pad = false;
  auto enter_tile = [&](int ti) {
    int t = sg + ti * gpt;
    const int tx = t % P.tiles_x; t /= P.tiles_x;
    const int ty = t % P.tiles_y; t /= P.tiles_y;
    g.img = t; g.n0 = n0; g.oy0 = ty * TH; g.ox0 = tx * 32; g.tile_in_img = ty * P.tiles_x + tx;
    bool pad = false;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 512;
      int pix = -1;
      if (e < HALO_PX * R) {
        const int hp = e / R;
        const int iy = g.oy0 - 1 + hp / HALO_W, ix = g.ox0 - 1 + hp % HALO_W;
        if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) pix = iy * P.w + ix;
        pad = pad || pix < 0;
      }
      a_pix[i] = pix;
    }
    a_pad = __ballot(pad) != 0ull;
  };
  const unsigned pixb = (unsigned)P.ld[0] * 2u;
  // halo image of `chunk` of the tile described by g / a_pix -> buffer `chunk`: begin (operands, padding), one DMA instruction
  // per slot (spread over the MFMA steps of the following phase: an LDS-DMA instruction holds the wave for 100-200 cycles)
  const unsigned short* ia_sp = nullptr; unsigned ia_la = 0;
  auto issue_a_begin = [&](int chunk) {
    ia_sp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.src[0] + (long long)g.img * P.img_stride[0] + chunk * 32));
    ia_la = xuni(lds0 + (unsigned)(W_BYTES + chunk * A_BYTES));
    if (a_pad) {
      char* ab = a_base + chunk * A_BYTES;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (tid + i * 512 < HALO_PX * R && a_pix[i] < 0) *reinterpret_cast<float4*>(ab + (tid + i * 512) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto issue_a_slot = [&](int i) {
    if (a_pix[i] >= 0) {
      const int e = tid + i * 512, hp = e / R;
      const unsigned q = (unsigned)((e % R) ^ ((hp >> 2) & 3));
      xglds16((unsigned)a_pix[i] * pixb + 16u * q, ia_sp, ia_la + i * 8192u);
    }
  };
  auto issue_a = [&](int chunk) {
    issue_a_begin(chunk);
#pragma unroll
    for (int i = 0; i < NA; ++i) issue_a_slot(i);
  };

  // Fragment byte offsets are tile-invariant: computed ONCE (PMC of the first version: 8 VALU + 4 SALU instructions per MFMA,
  // most of them this address arithmetic -- the vector ALU was as busy as the matrix pipe).  aoff[mt][tap] = k-step 0 of
  // tap (ky,kx) for pixel row mt; k-step 1 is the same address with bit 5 flipped (piece ^ 2).
  int aoff[MT][9];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hp = (2 * wave + mt) * HALO_W + li + (tap / 3) * HALO_W + tap % 3;
      aoff[mt][tap] = hp * 64 + ((lh ^ ((hp >> 2) & 3)) * 16);
    }
  const int b_frag = li * 16 + lh * 1024;

  XSEG_DECL;
  enter_tile(0);
  issue_a(0);
  issue_a(1);
  XSEG(0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  XSEG(1);

  int n_st = 0;                       // result stores this wave issued in the previous tile's epilogue (lower bound)
  for (int ti = 0; ti < T_me; ++ti) {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const XGeo gcur = g;
    auto compute = [&](int chunk, bool pend) {       // pend: the halo image begun by issue_a_begin goes out between the MFMAs
      const char* A = a_base + chunk * A_BYTES;
      const char* B = w_base + chunk * (9 * 4 * 1024) + b_frag;
      bf16x8 fa[2][MT], fb[2][NT];
      auto load_step = [&](int set, int st) {
        const int tap = st >> 1, ks = st & 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const bf16x8*>(A + (ks ? (aoff[mt][tap] ^ 32) : aoff[mt][tap]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const bf16x8*>(B + (tap * 4 + 2 * ks) * 1024 + nt * 512);
      };
      load_step(0, 0);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        if (st + 1 < 18) load_step((st + 1) & 1, st + 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st & 1][nt], fa[st & 1][mt], acc[mt][nt], 0, 0, 0);
      }
    };
    const bool more = ti + 1 < T_me;
    bool pend = false;
#pragma unroll 1
    for (int chunk = 0; chunk < 2; ++chunk) {
      compute(chunk, pend);
      pend = false;
      XSEG(2);
      // The image needed next (chunk 1 of this tile / chunk 0 of the next) was issued a whole phase ago; after the barrier
      // everybody is done with this chunk's buffer, which is refilled at once.  Chunk 1's image is OLDER than the previous
      // tile's result stores (vmcnt retires in issue order), so that wait leaves `n_st` operations in flight and the stores
      // drain under this tile's matrix work instead of in front of it; n_st is a lower bound of the stores really issued
      // (an instruction counts when at least one lane is active), which keeps the wait on the safe side.
      if (chunk == 0) xwait_vmcnt(n_st); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      XSEG(3);
      if (more) {
        // buffer 0 is refilled between the MFMAs of this tile's second phase; buffer 1 at once (the epilogue follows)
        if (chunk == 0) enter_tile(ti + 1);
        issue_a(chunk);
      }
      XSEG(4);
    }
    x_epilogue<MT, NT, false, false, LEAN>(P, gcur, acc, wave * 64, 0, gcur.tile_in_img * 8 + wave, bias_lds, li, lh);
    n_st = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int p = wave * 64 + mt * 32 + li;
      const bool pok = (gcur.oy0 + (p >> 5)) < P.oh && (gcur.ox0 + (p & 31)) < P.ow;
#pragma unroll
      for (int k = 0; k < 2 * NT; ++k) n_st += (__ballot(pok && (n0 + 16 * k + 8 * lh) < P.cout) != 0ull) ? 1 : 0;
    }
    XSEG(5);
  }
  XSEG_FLUSH;
}

// conv64_resident2_kernel's LEAN epilogue for one pixel row of a wave (32 pixels x 64 couts, the bias already inside the
// accumulators): stamps showed the general x_epilogue costing as many issue cycles as the tile's MFMAs (~700 vector instructions
// per wave and tile), and on one SIMD the other group's pending MFMA and these instructions take turns.  Here: pack to bf16 FIRST,
// ReLU as a packed signed-integer max, half-wave swaps on packed pairs, one address per pixel with immediate offsets
// (PACKED: 8 cvt + 8 max + 4 swaps + 2 stores per 32 x 32 accumulator tile instead of ~125).  Residual / per-pixel multiplier /
// LeakyReLU layers keep fp32 arithmetic (PACKED = false); their loads go out before the first store (in-order vmcnt).
template <int NT, bool PACKED>
__device__ __forceinline__ void r2_store_row(const XParams& P, const XGeo& g, f32x16 (&acc)[NT], int pix_base, int li, int lh) {
  const int p = pix_base + li;
  const int oy = g.oy0 + (p >> 5), ox = g.ox0 + (p & 31);
  const bool pok = oy < P.oh && ox < P.ow;
  const long long opl = (long long)g.img * P.OH * P.OW + (long long)oy * P.OW + ox;
  unsigned short* const op = reinterpret_cast<unsigned short*>(P.out) + opl * P.out_ld + g.n0 + 8 * lh;
  if (PACKED) {
    const bool relu = P.act == GPEMSR_ACT_RELU;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      unsigned pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[q][0] = xcvt_pk_bf16(acc[nt][4 * q], acc[nt][4 * q + 1]);
        pk[q][1] = xcvt_pk_bf16(acc[nt][4 * q + 2], acc[nt][4 * q + 3]);
      }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[q][j]) : "v"(pk[q][j]));      // bf16 sign bit = int16 sign bit
      }
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 1][1], false, false);
        if (pok) *reinterpret_cast<uint4*>(op + nt * 32 + 8 * gp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    }
  } else {
    const unsigned short* const rp = reinterpret_cast<const unsigned short*>(P.residual) + opl * P.res_ld + g.n0 + 8 * lh;
    uint4 rpre[NT][2];
    float mpre = 1.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const bool use = P.residual != nullptr && pok;                       // branch-free: see x_epilogue_stores
        const unsigned short* ap = use ? rp + nt * 32 + 16 * h2 : P.weight;
        uint4 u = *reinterpret_cast<const uint4*>(ap);
        if (!use) u = make_uint4(0u, 0u, 0u, 0u);
        rpre[nt][h2] = u;
      }
    {
      const bool usem = P.pixmul != nullptr && pok;
      const float mv = *(usem ? P.pixmul + opl : reinterpret_cast<const float*>(P.weight));
      mpre = usem ? mv : 1.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[nt][r];
      if (P.act == GPEMSR_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (P.act == GPEMSR_ACT_LRELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.1f * v[r]);
      }
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * gp + j]), __float_as_uint(v[4 * gp + 4 + j]), false, false);
          w8[j] = __uint_as_float(sw[0]); w8[4 + j] = __uint_as_float(sw[1]);
        }
        const uint4 u = rpre[nt][gp >> 1];
        w8[0] += xbf_lo(u.x); w8[1] += xbf_hi(u.x); w8[2] += xbf_lo(u.y); w8[3] += xbf_hi(u.y);
        w8[4] += xbf_lo(u.z); w8[5] += xbf_hi(u.z); w8[6] += xbf_lo(u.w); w8[7] += xbf_hi(u.w);
#pragma unroll
        for (int k = 0; k < 8; ++k) w8[k] *= mpre;
        if (pok) *reinterpret_cast<uint4*>(op + nt * 32 + 8 * gp) =
            make_uint4(xcvt_pk_bf16(w8[0], w8[1]), xcvt_pk_bf16(w8[2], w8[3]), xcvt_pk_bf16(w8[4], w8[5]), xcvt_pk_bf16(w8[6], w8[7]));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer (3x3, stride 1, 64 input channels, weights resident) with its phases OVERLAPPED.  Stamps of the kernel above:
// a tile spends 42 % of its time in the MFMA stages, 26 % waiting for halo images, 8 % issuing them and 24 % in the epilogue
// (store-throughput bound: 64 KB of results per tile) -- one after the other, because all eight waves are in the same phase.
// Here a 768-thread workgroup holds TWO groups of four multiplying waves (one wave per SIMD each) and four loader waves:
//   * group g walks its own stream of 8 x 32-pixel tiles in FOUR intervals per tile -- chunk 0, chunk 1, epilogue of the wave's first
//     pixel row, epilogue of its second -- and group 1 runs two intervals behind group 0:  (C0 | Ea) (C1 | Eb) (Ea | C0) (Eb | C1) ...
//     In every interval exactly one group feeds the matrix pipe while the other stores half of its results.  One workgroup-wide
//     barrier ends each interval.  (Round 3's stamps of the three-interval form (C0 | E) (C1 | C0) (E | C1) with the general epilogue: a
//     whole epilogue took 1.9x a chunk's MFMAs, so two of three intervals were epilogue-long.  With r2_store_row the epilogue is a
//     quarter of that and both forms measure the same; -DR2_IV=3 builds the three-interval one.)
//   * the loader waves issue the halo images (10 x 34 pixels x 32 channels, two buffers per group) for tile k + 1 as soon as
//     the barrier has freed a buffer -- four intervals before it is needed -- and absorb the DMA issue stalls and the waits.
// LDS: weights 73,728 + 4 x 21,760 + bias <= 163,840 bytes.
// AXF: the halo images are staged through the loader waves' registers and get the per-(image, channel) affine map + ReLU of a folded
// GroupNorm apply on the way (see conv_bf16_kernel): loads go out one interval before the transformed image is written, two before
// it is read.
template <bool LEAN, bool AXF = false>
__global__ __launch_bounds__(768, 3) void conv64_resident2_kernel(XParams P) {
  constexpr int HALO_W = 34, HALO_H = 10, HALO_PX = HALO_W * HALO_H, R = 4;
  constexpr int A_BYTES = HALO_PX * R * 16;            // 21,760
  constexpr int W_BYTES = 2 * 9 * 4 * 64 * 16;         // 73,728
  constexpr int NA = (HALO_PX * R + 255) / 256;        // 6 slots per loader thread per image
  constexpr int NW = W_BYTES / 16 / 256;               // 18 slots per loader thread for the weight slab
  constexpr int MT = 2, NT = 2;
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  float* const bias_lds = reinterpret_cast<float*>(xsm + W_BYTES + 4 * A_BYTES);
  const unsigned xsm_lds = xlds_addr(xsm);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order (bijective; workgroup b runs on XCD b % 8): logically consecutive workgroups -- they walk neighbouring tiles at the
  // same time -- share an XCD, so the halo rows / columns two tiles have in common are L2 hits instead of a second HBM read
  int bid = (int)blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int tn = bid % P.tiles_n, sg = bid / P.tiles_n;
  const int n0 = tn * 64;
  const int gpt = P.gpt, NS = P.ns;
  const int T_me = (NS - sg + gpt - 1) / gpt;          // 8 x 32 tiles of this workgroup: tile j = 2 k + g goes to group g
  const int T0 = (T_me + 1) / 2, T1 = T_me / 2;
  // IV intervals per tile, group 1 DL intervals behind group 0: (3, 1) = (C0 | E) (C1 | C0) (E | C1); (4, 2) = (C0 | Ea) (C1 | Eb) (Ea | C0) (Eb | C1)
#ifndef R2_IV
#define R2_IV 4
#endif
  constexpr int IV = R2_IV, DL = IV == 4 ? 2 : 1;
  const int NI = (IV * T0 > IV * T1 + DL) ? IV * T0 : IV * T1 + DL;      // intervals (T1 <= T0 <= T1 + 1)

  x_stage_bias(P, bias_lds, P.nbias, 768);

  auto tile_geo = [&](int j) -> XGeo {
    int t = sg + j * gpt, tx, ty;
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    xdivmod(t, P.tiles_y, P.mg_y, t, ty);
    XGeo g;
    g.img = t; g.n0 = n0; g.oy0 = ty * 8; g.ox0 = tx * 32; g.tile_in_img = ty * P.tiles_x + tx;
    return g;
  };
  auto end_interval = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (AXF && wave >= 8) {
    // ------------------------------------------------ loader waves, register-staged + transformed images ------------------------------------------------
    const int dtid = tid - 512, dwave = wave - 8;
    const unsigned lds0 = xuni(xsm_lds + (unsigned)dwave * 1024u);
    const unsigned pixb = (unsigned)P.ld[0] * 2u;
    {   // weight slab (LDS-DMA, as below)
      const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.weight));
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int e = dtid + i * 256;
        int row = n0 + (e & 63);
        row = row < P.cout ? row : P.cout - 1;
        xglds16((unsigned)((e >> 6) * P.cout + row) * 16u, wp, lds0 + i * 4096u);
      }
    }
    const int q = dtid & 3;                              // this thread's logical 16-byte piece (8 channels) of every halo pixel
    const float lo = P.ax_relu ? 0.f : -3.0e38f;
    int a_pix[2][NA];
    int a_img[2] = {0, 0};
    uint4 raw[2][NA];
    float sc[2][8], sh[2][8];
    unsigned padmask[2] = {0u, 0u};
    int pend[2] = {-1, -1};                              // chunk of the image held in raw[g] (-1: none)
    auto enter = [&](const int g, int k) {               // g compile-time
      const XGeo t = tile_geo(2 * k + g);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = dtid + i * 256;
        int pix = -1;
        if (e < HALO_PX * R) {
          const int hp = e / R;
          const int iy = t.oy0 - 1 + hp / HALO_W, ix = t.ox0 - 1 + hp % HALO_W;
          if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) pix = iy * P.w + ix;
        }
        a_pix[g][i] = pix;
      }
      a_img[g] = t.img;
    };
    auto load = [&](const int g, const int c) {          // chunk c of group g's current tile -> registers
      const char* sp = reinterpret_cast<const char*>(P.src[0] + (long long)a_img[g] * P.img_stride[0] + c * 32 + 8 * q);
      const float* scp = P.axs + (long long)a_img[g] * 64 + c * 32 + 8 * q;
      const float* shp = P.axh + (long long)a_img[g] * 64 + c * 32 + 8 * q;
      unsigned pm = 0u;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int pix = a_pix[g][i];
        pm |= (pix < 0 ? 1u : 0u) << i;
        raw[g][i] = *reinterpret_cast<const uint4*>(sp + (size_t)(unsigned)(pix < 0 ? 0 : pix) * pixb);
      }
      padmask[g] = pm;
      const float4 s0 = *reinterpret_cast<const float4*>(scp), s1 = *reinterpret_cast<const float4*>(scp + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(shp), h1 = *reinterpret_cast<const float4*>(shp + 4);
      sc[g][0] = s0.x; sc[g][1] = s0.y; sc[g][2] = s0.z; sc[g][3] = s0.w; sc[g][4] = s1.x; sc[g][5] = s1.y; sc[g][6] = s1.z; sc[g][7] = s1.w;
      sh[g][0] = h0.x; sh[g][1] = h0.y; sh[g][2] = h0.z; sh[g][3] = h0.w; sh[g][4] = h1.x; sh[g][5] = h1.y; sh[g][6] = h1.z; sh[g][7] = h1.w;
      pend[g] = c;
    };
    auto write = [&](const int g) {                      // registers -> transformed image (g, pend[g])
      if (pend[g] < 0) return;
      char* ab = xsm + W_BYTES + (2 * g + pend[g]) * A_BYTES;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = dtid + i * 256;
        if (e < HALO_PX * R) {
          const int hp = e / R;
          const unsigned in[4] = {raw[g][i].x, raw[g][i].y, raw[g][i].z, raw[g][i].w};
          unsigned o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float a = fmaxf(fmaf(__uint_as_float(in[k] << 16), sc[g][2 * k], sh[g][2 * k]), lo);
            const float b = fmaxf(fmaf(__uint_as_float(in[k] & 0xFFFF0000u), sc[g][2 * k + 1], sh[g][2 * k + 1]), lo);
            o[k] = xcvt_pk_bf16(a, b);
          }
          const bool pad = (padmask[g] >> i) & 1u;
          *reinterpret_cast<uint4*>(ab + hp * 64 + ((q ^ ((hp >> 2) & 3)) * 16)) = pad ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
      pend[g] = -1;
    };
    // prologue: tile 0 of both groups
    enter(0, 0); load(0, 0); write(0); load(0, 1); write(0);
    if (T1 > 0) { enter(1, 0); load(1, 0); write(1); load(1, 1); write(1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weight slab
    end_interval();
    for (int i = 0; i < NI; ++i) {
      write(0); write(1);                                // images loaded during the previous interval; read from the next one on
      // refill the buffers the last barrier freed: group g multiplied chunk (i-1-DL g) % IV of its tile (i-1-DL g) / IV in interval i-1
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int u = i - 1 - DL * g;
        if (u >= 0) {
          const int k = u / IV, c = u - IV * k;
          if (c < 2 && k + 1 < (g ? T1 : T0)) {
            if (c == 0) { enter(g, k + 1); load(g, 0); } else load(g, 1);
          }
        }
      }
      end_interval();
    }
    return;
  }
  if (wave >= 8) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    const int dtid = tid - 512, dwave = wave - 8;
    const unsigned lds0 = xuni(xsm_lds + (unsigned)dwave * 1024u);
    const unsigned pixb = (unsigned)P.ld[0] * 2u;
    int issued = 0;
    {   // weight slab: rows n0 .. n0+63 of every (chunk, tap, piece) -- 72 runs of 1 KiB (rows past cout clamped)
      const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.weight));
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int e = dtid + i * 256;
        int row = n0 + (e & 63);
        row = row < P.cout ? row : P.cout - 1;
        xglds16((unsigned)((e >> 6) * P.cout + row) * 16u, wp, lds0 + i * 4096u);
      }
      issued += NW;
    }
    int a_pix[2][NA];                                  // per group: the tile whose images are being issued
    int a_img[2] = {0, 0}, a_cnt[2] = {0, 0};
    bool a_pad[2] = {false, false};
    int mark[2][2] = {{0, 0}, {0, 0}};
    auto enter = [&](const int g, int k) {             // g compile-time
      const XGeo t = tile_geo(2 * k + g);
      int cnt = 0; bool pad = false;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = dtid + i * 256;
        int pix = -1;
        if (e < HALO_PX * R) {
          const int hp = e / R;
          const int iy = t.oy0 - 1 + hp / HALO_W, ix = t.ox0 - 1 + hp % HALO_W;
          if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) pix = iy * P.w + ix;
          pad = pad || pix < 0;
        }
        a_pix[g][i] = pix;
        cnt += (__ballot(pix >= 0) != 0ull) ? 1 : 0;
      }
      a_img[g] = t.img; a_cnt[g] = cnt; a_pad[g] = __ballot(pad) != 0ull;
    };
    auto issue = [&](const int g, const int c) {       // image of chunk c of group g's current tile -> buffer (g, c)
      const unsigned short* sp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.src[0] + (long long)a_img[g] * P.img_stride[0] + c * 32));
      const unsigned la = xuni(lds0 + (unsigned)(W_BYTES + (2 * g + c) * A_BYTES));
      if (a_pad[g]) {
        char* ab = xsm + W_BYTES + (2 * g + c) * A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i)
          if (dtid + i * 256 < HALO_PX * R && a_pix[g][i] < 0) *reinterpret_cast<float4*>(ab + (dtid + i * 256) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (a_pix[g][i] >= 0) {
          const int e = dtid + i * 256, hp = e / R;
          const unsigned q = (unsigned)((e % R) ^ ((hp >> 2) & 3));
          xglds16((unsigned)a_pix[g][i] * pixb + 16u * q, sp, la + i * 4096u);
        }
      issued += a_cnt[g];
      mark[g][c] = issued;
    };
    // prologue: tile 0 of both groups, in the order of use
    enter(0, 0); issue(0, 0); issue(0, 1);
    if (T1 > 0) { enter(1, 0); issue(1, 0); issue(1, 1); }
    xwait_vmcnt(issued - mark[0][0]);
    end_interval();
    R2L_DECL;
    for (int i = 0; i < NI; ++i) {
      // refill the buffers the last barrier freed: group g multiplied chunk (i-1-DL g) % IV of its tile (i-1-DL g) / IV in interval i-1
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int u = i - 1 - DL * g;
        if (u >= 0) {
          const int k = u / IV, c = u - IV * k;
#ifdef GP16_STAMP
          if (P.dbg == 103 || P.dbg == 104) continue;   // diagnostic: no refills (stale images; timing only)
#endif
          if (c < 2 && k + 1 < (g ? T1 : T0)) {
            if (c == 0) { enter(g, k + 1); issue(g, 0); } else issue(g, 1);
          }
        }
      }
      R2L(0);
      // the images interval i + 1 reads must have landed
      int need = 0;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int u = i + 1 - DL * g;
        if (u >= 0) {
          const int k = u / IV, c = u - IV * k;
          if (c < 2 && k < (g ? T1 : T0)) { const int m = c ? mark[g][1] : mark[g][0]; need = m > need ? m : need; }
        }
      }
      xwait_vmcnt(issued - need);
      R2L(1);
      end_interval();
      R2L(2);
    }
    R2L_FLUSH;
    return;
  }

  // ------------------------------------------------ multiplying waves ------------------------------------------------
  if (AXF) __builtin_amdgcn_s_setprio(2);              // (the transforming loader waves must not take their issue slots)
  const int g = wave >> 2, w4 = wave & 3;              // group, wave of the group: pixel rows 2 w4, 2 w4 + 1 of the 8 x 32 tile
  const int T_g = g ? T1 : T0;
  // fragment byte offsets (tile-invariant): one entry per (halo row, kx); k-step 1 flips bit 5
  unsigned aoff[MT + 2][3];
#pragma unroll
  for (int rr = 0; rr < MT + 2; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int hp = (2 * w4 + rr) * HALO_W + li + kx;
      aoff[rr][kx] = (unsigned)(W_BYTES + 2 * g * A_BYTES + hp * 64 + ((lh ^ ((hp >> 2) & 3)) * 16));   // (all bases are multiples of 64)
    }
  const unsigned b_frag = (unsigned)(li * 16 + lh * 1024);

  end_interval();                                      // prologue barrier (weights + first image have landed)
  for (int s = 0; s < DL * g; ++s) end_interval();     // group 1 runs DL intervals behind
  XSEG_DECL;
  for (int k = 0; k < T_g; ++k) {
    f32x16 acc[MT][NT];                                  // start from the bias (zeros when the layer has none: x_stage_bias)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int bi = n0 + nt * 32 + 8 * q + 4 * lh;              // (couts past the end of a ragged slab: any valid entry, never stored)
        bi = bi < P.nbias ? bi : 0;
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + bi);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][nt][4 * q] = b4.x; acc[mt][nt][4 * q + 1] = b4.y; acc[mt][nt][4 * q + 2] = b4.z; acc[mt][nt][4 * q + 3] = b4.w; }
      }
#pragma unroll 1
    for (int chunk = 0; chunk < 2; ++chunk) {
      const unsigned A = (unsigned)(chunk * A_BYTES);
      const unsigned B = b_frag + (unsigned)(chunk * (9 * 4 * 1024));
      unsigned arow[MT + 2][3];
#pragma unroll
      for (int rr = 0; rr < MT + 2; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) arow[rr][kx] = xsm_lds + A + aoff[rr][kx];
      const unsigned bbase = xsm_lds + B;
      bf16x8 fa[3][MT], fb[3][NT];
#ifdef GP16_STAMP
      if (P.dbg != 101) {
#endif
      r2_reads<0>(fa[0], fb[0], arow, 0u, bbase);
      r2_reads<1>(fa[1], fb[1], arow, 0u, bbase);
      asm volatile("s_nop 4" ::: "memory");              // (VALU-written accumulators -> first MFMA: the hazard recognizer does not see into asm)
      r2_steps<0>(acc, fa, fb, arow, 0u, bbase);
      // (the accumulators were written by asm: the compiler does not know that VALU reads of them need the matrix pipe drained)
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#ifdef GP16_STAMP
      }
#endif
      XSEG(2);
      end_interval();
      XSEG(3);
    }
    const XGeo geo = tile_geo(2 * k + g);
    // opaque per-tile copies of the lane coordinates: everything the epilogue derives from them (8 store addresses, residual
    // addresses, ...) is tile-invariant, and hoisted above the MFMA loop it costs ~30 registers the 168-register budget does not
    // have -- the spill reloads then sit behind the epilogue's own stores on the in-order vmcnt (measured: 1.35x slower)
    int li2 = li, lh2 = lh;
    asm volatile("" : "+v"(li2), "+v"(lh2));
#ifdef GP16_STAMP
    if (P.dbg == 102 || P.dbg == 104) { XSEG(5); end_interval(); XSEG(4); XSEG(5); end_interval(); XSEG(4); continue; }
#endif
    if (P.gn_ws || !LEAN || P.cout - n0 < 64) {          // general epilogue, in one piece (the bias is in the accumulators already)
      XParams Q = P;
      Q.bias = nullptr;
      x_epilogue<MT, NT, false, false, LEAN, true>(Q, geo, acc, w4 * 64, 0, geo.tile_in_img * 4 + w4, bias_lds, li2, lh2);
      if (IV == 4) end_interval();
    } else if (P.residual || P.pixmul || P.act == GPEMSR_ACT_LRELU) {
      r2_store_row<NT, false>(P, geo, acc[0], w4 * 64, li2, lh2);
      if (IV == 4) {
        XSEG(5);
        end_interval();
        XSEG(4);
      }
      asm volatile("" : "+v"(li2), "+v"(lh2));
      r2_store_row<NT, false>(P, geo, acc[1], w4 * 64 + 32, li2, lh2);
    } else {
      r2_store_row<NT, true>(P, geo, acc[0], w4 * 64, li2, lh2);
      if (IV == 4) {
        XSEG(5);
        end_interval();
        XSEG(4);
      }
      asm volatile("" : "+v"(li2), "+v"(lh2));
      r2_store_row<NT, true>(P, geo, acc[1], w4 * 64 + 32, li2, lh2);
    }
    XSEG(5);
    end_interval();
    XSEG(4);
  }
  XSEG_FLUSH;
  for (int s = IV * T_g + DL * g; s < NI; ++s) end_interval();
}


template <int CK, int BN, int WM, int WN, int TH, int TPS, bool CONVT = false, bool GEMM = false, int NL = 0, bool LEAN = false, int SS = 0, bool AXF = false,
          bool TCOMP = false, int XEPI = 0>
static int launch_x(const XParams& P, size_t lds, hipStream_t st) {
  auto kfn = conv_bf16_kernel<CK, BN, WM, WN, TH, TPS, CONVT, GEMM, NL, LEAN, SS, AXF, TCOMP, XEPI>;
  if (lds > 64 * 1024) {
    static dev_once_t done{0};
    if (dev_once_begin(done)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "conv2d_bf16: cannot raise the dynamic LDS limit");
      dev_once_done(done);
    }
  }
  // persistent workgroups: as many as the chip holds at once (two per CU when the LDS footprint allows), each walking
  // tiles blockIdx.x, +gridDim.x, ...
  const int cus = device_cus();
  constexpr int NTH = (WM * WN + NL) * 64;
  const int slots = cus * ((NTH == 256 && lds <= 80 * 1024) ? 2 : 1);
  const int grid = P.nblocks < slots ? P.nblocks : slots;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(NTH), lds, st, P);
  return check_launch("conv_bf16_kernel");
}

}  // namespace gpemsr

using namespace gpemsr;

namespace {
struct XPlan { int CK, BN, TH, TPS, WM, WN, NL; bool tr, gemm, resident, lean, axf, axf_ok, tres, tcomp, gdirect; int res_form; size_t lds; };

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
  bool ck64 = gemm && d->variant != 1 && d->cout > 64;       // 1x1 / matrix products with 128-column tiles: 64-deep stages
  for (int s = 0; s < d->nsrc; ++s) ck64 = ck64 && (d->src[s].c % 64 == 0);
  const int CK = ck64 ? 64 : (ck32 ? 32 : 16);
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
  P.gn_ws = d->gn_partials; P.gn_cpg = d->gn_cpg;
  P.axs = d->a_scale; P.axh = d->a_shift; P.ax_relu = d->a_relu;
  P.rowmax = d->rowmax;
  L.axf = d->a_scale != nullptr || d->a_shift != nullptr;
  P.nbias = d->cout;
  const int bias_bytes = ((d->cout + 7) & ~7) * 4;
  int BN, TH, TPS, WM, WN, NL = 0;
  L.resident = false; L.tcomp = false; L.gdirect = false;
  const int var = d->variant;                // 0 = default tile choice; > 0: alternatives (A/B tuning, scripts/conv16_microbench.py)
  if (tr) {
    P.kw = 2; P.kk = 4; P.stride = 1; P.pad = 0; P.cout = 4 * d->cout;
    P.oh = d->h; P.ow = d->w; P.OH = 2 * d->h; P.OW = 2 * d->w;
    P.store_mode = XS_CONVT; P.cq = d->cout; BN = 128; TH = 4; TPS = 2; WM = 4; WN = 1;
    if (var != 1) { TH = 8; WM = 8; NL = 4; TPS = (var == 2) ? 2 : 4; }   // 8x32 px x (4 phases x 32 couts), 8 multiplying + 4 loader waves;
                                                            // a stage = all four taps of a chunk (variant 2: tap pairs, round 2's form)
    // whole-chunk stages with 32-channel chunks read the COMPACT weight form (nine non-zero (tap, phase) blocks, behind the staged
    // form; weight_forms bit 1); without it the layer takes the tap-pair stages of the staged form
    L.tcomp = var != 1 && TPS == 4 && CK == 32 && (d->weight_forms & 2) != 0;
    if (var != 1 && TPS == 4 && CK == 32 && !L.tcomp) TPS = 2;
  } else {
    P.kw = d->ksize; P.kk = d->ksize * d->ksize; P.stride = d->stride; P.pad = d->ksize / 2; P.cout = d->cout;
    P.oh = (d->h + 2 * P.pad - d->ksize) / P.stride + 1; P.ow = (d->w + 2 * P.pad - d->ksize) / P.stride + 1;
    P.store_mode = d->pixel_shuffle ? XS_PIXSHUF : (d->kpack ? XS_KPACK : XS_PLAIN); P.cq = d->cout / 4;
    P.OH = d->pixel_shuffle ? 2 * P.oh : P.oh; P.OW = d->pixel_shuffle ? 2 * P.ow : P.ow;
    BN = d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128);
    if (gemm) {
      TH = 4; TPS = 1; WM = BN == 32 ? 4 : 2; WN = BN == 32 ? 1 : 2;
      if (var != 1 && BN == 128) { TH = 8; WM = 4; WN = 2; NL = 4; }        // 256 px x 128 columns, 8 + 4 waves
      else if (var != 1 && BN == 64) { TH = 16; WM = 8; WN = 1; NL = 4; }   // 512 px x 64 columns
      // 64-channel sources (an even number >= 4 of chunks), couts in blocks of 256: activations straight into registers, only the weights
      // through LDS (gemm_bf16.hip) -- 256 px x 256 columns on 8 waves.  variant 14 / GPEMSR_GEMM_DIRECT=0 keep the ring kernel (A/B).
      const char* gd_e = getenv("GPEMSR_GEMM_DIRECT");     // (read per launch: A/B runs switch it inside one process)
      const bool gd_env = !(gd_e && gd_e[0] == '0');
      if (ck64 && gd_env && var == 0 && d->cout % 256 == 0 && nchunk_total >= 4 && nchunk_total % 2 == 0 && !d->pixmul && !d->gn_partials && !L.axf &&
          bias_bytes <= 16 * 1024) {
        L.gdirect = true; BN = 256; TH = 8; WM = 8; WN = 1; NL = 0;
      }
    }
    else if (d->ksize == 7) {
      if (var == 1) {
        // 256-thread form: single-chunk layers (cin <= 32) keep ONE A image and a 2-deep ring of 7-tap row stages; wider inputs
        // use 32-cout blocks
        TH = 4; TPS = 7;
        if (nchunk_total > 1 || BN == 32) { BN = 32; WM = 4; WN = 1; } else { BN = 64; WM = 2; WN = 2; }
      } else {
        // big tile: 16x32 pixels, 8 waves (one workgroup per CU): the 49-tap weights are staged once per 512 pixels
        TH = 16; TPS = 7; WM = 8; WN = 1;
        if (BN == 128) BN = 64;
        // loader waves only for 16-channel chunks: with 32-channel chunks a loader thread owns 14 halo slots, the slot loops stop
        // being unrolled and the whole cursor state goes to scratch (1.1 KB per lane; measured 4x slower)
        if (var != 4 && CK == 16) NL = 4;
        // 32-channel chunks, couts <= 32 (SpyNet 64 -> 32 and 32 -> 16): EIGHT loader waves (7 halo slots per loader thread, 16 waves per
        // CU = 128 registers, which the 32-accumulator multiplying waves fit)
        // 64 couts (SpyNet 32 -> 64): two 32-cout tiles on the 8-loader form -- since the packed epilogue 1.02 vs 0.91 PFLOP/s for the
        // 64-cout tile whose multiplying waves issue their own DMA (22 % of its time; variant 7 keeps that form for A/B)
        if (var != 7 && CK == 32 && BN == 64) BN = 32;
        if (var != 4 && CK == 32 && BN == 32) NL = 8;
      }
    }
    else if (d->stride == 2) {
      TH = 2; WM = 2; WN = 2; TPS = (BN == 128) ? 1 : 3;
      // loader-wave form: 4 x 32 output pixels (9 x 65 halo) x 128 / 64 couts on 4 x 2 multiplying waves
      // (64 couts: a stage is the whole 32-channel chunk -- 18 instead of 6 MFMAs of a wave between barriers; variant 2: row stages)
      if (var != 1 && BN >= 64 && CK == 32) { TH = 4; WM = 4; WN = 2; TPS = (BN == 64 && var != 2 && var != 12 && nchunk_total >= 2) ? 9 : 3; NL = 4; }
    }
    else if (BN == 128) {
      if (var == 1) { TH = 8; TPS = 1; WM = 4; WN = 1; }          // 8x32 px, wave = 64 px x 128 couts, tap stages
      else if (var == 2) { TH = 4; TPS = 3; WM = 2; WN = 2; }     // 4x32 px x 128 couts, 256 threads, row stages
      else if (var == 4) { BN = 64; TH = 16; TPS = 3; WM = 8; WN = 1; }   // big tile, 8 waves that all load and multiply
      // big tile: 16x32 px x 64 couts, 8 multiplying + 4 loader waves; a stage is a whole 32-channel chunk (9 taps: one barrier per
      // 72 MFMAs of a wave; +10-13 % over row stages), variant 5: row stages
      else { BN = 64; TH = 16; TPS = (var == 5 || nchunk_total < 2) ? 3 : 9; WM = 8; WN = 1; NL = 4; }
    }
    else if (var == 1 || var == 2) { TH = 8; TPS = (var == 1) ? 1 : 3; WM = 4; WN = 1; }   // 256-thread forms
    else { TH = 16; TPS = (var == 5 || BN == 32 || nchunk_total < 2) ? 3 : 9; WM = 8; WN = 1; NL = 4; }   // (one chunk: row stages keep the ring busy)   // couts <= 64 on the same loader-wave big tile
  }
  L.BN = BN; L.TH = TH; L.TPS = TPS; L.WM = WM; L.WN = WN; L.NL = NL;
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
  P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y; P.mg_n = 0xFFFFFFFFu / (unsigned)P.tiles_n;
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d_bf16: grid too large");
  P.nblocks = (int)nb;
  P.gn_parts = P.tiles_y * P.tiles_x * WM;                   // one row of partial sums per wave row of a tile
  if (d->gn_partials) GP_REQUIRE(P.store_mode == XS_PLAIN && !tr && d->cout % 4 == 0 && (reinterpret_cast<uintptr_t>(d->gn_partials) & 15) == 0,
                                 "conv2d_bf16: gn partial sums need the plain store, cout %% 4 == 0 and a 16-byte aligned workspace");
  const int R = CK / 8;
  const int nth = (WM * WN + NL) * 64;
  const int dth = NL ? NL * 64 : nth;                       // threads that issue DMA
  P.na = cdiv((long long)P.halo_px * R, dth);
  P.nb = cdiv(L.tcomp ? (long long)R * 288 : (long long)TPS * R * BN, dth);
  GP_REQUIRE(P.na <= XA_LOADS && P.nb <= XB_LOADS, "conv2d_bf16: tile too large (na=%d nb=%d)", P.na, P.nb);
  P.a_bytes = P.halo_px * R * 16;                          // lanes past the image are masked: no padding to whole DMA pieces
  P.b_bytes = L.tcomp ? R * 288 * 16 : TPS * R * BN * 16;
  if (L.tcomp) P.weight = reinterpret_cast<const unsigned short*>(d->weight) + 16ll * nchunk_total * CK * d->cout;    // behind the staged form
  GP_REQUIRE(P.kk % TPS == 0, "conv2d_bf16: taps per stage must divide the tap count");
  P.spc = P.kk / TPS;
  // Ring depth / number of halo images: as deep as 80 KiB per workgroup (two workgroups per CU) allows, at most 4; the
  // persistent stream looks at most ONE tile ahead, hence ring <= stages per tile and images <= chunks per tile.
  const int budget = (nth >= 512 ? 158 : 80) * 1024 - bias_bytes;      // 512 / 768 threads: one workgroup per CU
  const int stages = nchunk_total * P.spc;
  int ring = stages < 4 ? (stages < 2 ? 2 : stages) : 4;
  // conv: two halo images; a single-chunk layer (SpyNet's 16 / 32-channel 7x7 convolutions) used to keep ONE and to wait for the next
  // tile's image with the matrix pipe idle -- its cursor now runs two tiles ahead (variant 8: the single image)
  int n_abuf = gemm ? (nchunk_total < 4 ? nchunk_total : 4) : 2;
  // transposed, whole-chunk stages: 18 MFMAs of a wave per stage (~1.5k cycles) cannot cover the round trip of the NEXT chunk's halo image
  // (in-kernel stamps, 128 -> 64 at 256^2: 32 % of a tile in the stages, 46 % waiting for images issued one stage earlier): keep up to
  // four images, i.e. issue three stages ahead (the compact weight stages leave the LDS for it); variant 5: the two-image form
  if (tr && L.tcomp && var != 5) n_abuf = nchunk_total < 4 ? (nchunk_total < 2 ? 2 : nchunk_total) : 4;
  if (!tr && !gemm && d->ksize == 3 && d->stride == 2 && var == 12 && nchunk_total >= 3) n_abuf = 3;     // experiment: row stages, three halo images
  if (!gemm && nchunk_total < 2 && (var == 8 || 2 * P.a_bytes + 2 * P.b_bytes > budget)) n_abuf = 1;
  while (n_abuf * P.a_bytes + ring * P.b_bytes > budget && (ring > 2 || (gemm && n_abuf > 2))) {
    if (gemm && n_abuf > 2 && n_abuf >= ring) --n_abuf;      // matrix products: both rings advance per stage, keep them level
    else if (ring > 2) --ring;
    else --n_abuf;
  }
  if (stages < 2) ring = 2;                                  // (a second slot that is simply never filled)
  P.ring = ring;
  P.n_abuf = n_abuf;
  size_t lds = (size_t)n_abuf * P.a_bytes + (size_t)ring * P.b_bytes + bias_bytes;
  GP_REQUIRE(lds <= 160 * 1024, "conv2d_bf16: LDS %zu too large", lds);
  // 64 input channels, 3x3, stride 1, one source: the weights-resident kernel (one 64-cout slab per workgroup)
  if (!tr && !gemm && d->ksize == 3 && d->stride == 1 && d->nsrc == 1 && d->src[0].c == 64 && d->weight_image_stride == 0 && var != 3 &&
      bias_bytes <= 8 * 1024) {
    L.resident = true;
    // form 2 (default): two staggered groups of multiplying waves + loader waves on 8 x 32 tiles; form 1 (variant 6, or a bias
    // array that does not fit beside four halo buffers): eight waves in lockstep on 16 x 32 tiles
    L.res_form = (var == 6 || bias_bytes > 3072) ? 1 : 2;
    const int rows = L.res_form == 2 ? 8 : 16;
    P.tiles_n = cdiv(P.cout, 64);
    P.tiles_x = cdiv(P.ow, 32); P.tiles_y = cdiv(P.oh, rows);
    P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y; P.mg_n = 0xFFFFFFFFu / (unsigned)P.tiles_n;
    const long long ns = (long long)d->n * P.tiles_y * P.tiles_x;
    GP_REQUIRE(ns < (1ll << 31), "conv2d_bf16: grid too large");
    P.ns = (int)ns;
    P.gn_parts = P.tiles_y * P.tiles_x * (L.res_form == 2 ? 4 : 8);
    lds = L.res_form == 2 ? 73728 + 4 * 21760 + bias_bytes : 73728 + 2 * 39168 + bias_bytes;
  }
  if (L.gdirect) { P.ring = 4; lds = (size_t)4 * (8 * 256 * 16) + bias_bytes; }     // four 32 KB weight stages, no A images
  L.lds = lds;
  // the lean epilogue (see x_epilogue) covers this descriptor?
  L.lean = P.store_mode == XS_PLAIN && !d->out_f32 && !d->out32 && (!d->residual || !d->res_f32) && d->cout % 8 == 0 &&
           (d->act == GPEMSR_ACT_NONE || d->act == GPEMSR_ACT_RELU || d->act == GPEMSR_ACT_LRELU);
  if (tr) L.lean = !d->out_f32 && !d->out32 && !d->residual && !d->pixmul && !d->gn_partials &&
                   (d->act == GPEMSR_ACT_NONE || d->act == GPEMSR_ACT_RELU || d->act == GPEMSR_ACT_LRELU);
  // transposed, 64 input channels, 64-cout slabs, resident-form weights present behind the staged form: the weights-resident kernel
  L.tres = tr && L.lean && (d->weight_forms & 1) && d->nsrc == 1 && d->src[0].c == 64 && d->cout % 64 == 0 && var != 3 && bias_bytes <= 4096 &&
           d->out_ld % 8 == 0;
  if (L.tres) {
    P.tiles_n = d->cout / 64;
    P.tiles_x = cdiv(d->w, 32); P.tiles_y = cdiv(d->h, 8);
    P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y; P.mg_n = 0xFFFFFFFFu / (unsigned)P.tiles_n;
    const long long ns = (long long)d->n * P.tiles_y * P.tiles_x;
    GP_REQUIRE(ns < (1ll << 31), "conv2d_bf16: grid too large");
    P.ns = (int)ns;
    P.nbias = d->cout;
    // the resident slabs follow the staged form [cin/CK][4 taps][CK/8][4 cout][8] = 16 cin cout elements
    P.weight = reinterpret_cast<const unsigned short*>(d->weight) + 16ll * 64 * d->cout;
    L.lds = 73728 + 4 * 19008 + bias_bytes;
  }
  // source transform (a_scale / a_shift: a folded GroupNorm apply): kernels that stage the halo image through registers exist for the
  // 64-channel weights-resident form and for the wide 3x3 loader-wave tile -- every second convolution of a VQGAN block
  L.axf_ok = !tr && !gemm && d->ksize == 3 && d->stride == 1 && d->nsrc == 1 && d->src_image_stride[0] < 0 && d->weight_image_stride == 0 && L.lean &&
             (L.resident ? L.res_form == 2
                         : (NL == 4 && BN == 64 && WM == 8 && WN == 1 && TH == 16 && TPS == 9 && CK == 32 && n_abuf == 2 && ring <= stages));
  if (L.axf) GP_REQUIRE(L.axf_ok && d->a_scale && d->a_shift && (reinterpret_cast<uintptr_t>(d->a_scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->a_shift) & 15) == 0,
                        "conv2d_bf16: no source-transform kernel for this layer (ask gpemsr_conv2d_bf16_axf_ok) or misaligned scale / shift tables");
  return GPEMSR_OK;
}
}  // namespace

static_assert(sizeof(gpemsr_conv16_desc) == 272, "gpemsr_conv16_desc layout changed: update gpemsr_amd/_abi.py");

extern "C" int gpemsr_conv2d_bf16_axf_ok(const gpemsr_conv16_desc* d) {
  XParams P{}; XPlan L{};
  gpemsr_conv16_desc t = *d;
  t.a_scale = nullptr; t.a_shift = nullptr;
  const int rc = plan_x(&t, P, L);
  return rc == GPEMSR_OK ? (L.axf_ok ? 1 : 0) : rc;
}

extern "C" int gpemsr_conv2d_bf16_gn_parts(const gpemsr_conv16_desc* d) {
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  return rc == GPEMSR_OK ? P.gn_parts : rc;
}

// (value, column) records per GEMM row a launch with d->rowmax leaves: tiles_n x WN; < 0: error / no such kernel for this geometry
extern "C" int gpemsr_conv2d_bf16_rowmax_parts(const gpemsr_conv16_desc* d) {
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  if (rc != GPEMSR_OK) return rc;
  if (!L.gdirect && !(L.NL == 4 && L.gemm && L.BN == 128 && L.WM == 4 && L.WN == 2 && L.TH == 8 && L.TPS == 1 && L.CK == 64))
    return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: row maxima need the 1x1 form with cout > 64 and 64-channel sources");
  return P.tiles_n * L.WN;
}

namespace gpemsr {
__global__ __launch_bounds__(256) void rowmax_finish_kernel(const float2* ws, long long rows, int parts, int* idx) {
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    float best = -3.4e38f; int bc = 0x7fffffff;
    for (int p = 0; p < parts; ++p) {
      const float2 v = ws[r * parts + p];
      const int c = __float_as_int(v.y);
      if (v.x > best || (v.x == best && c < bc)) { best = v.x; bc = c; }
    }
    idx[r] = bc;
  }
}
}  // namespace gpemsr

extern "C" int gpemsr_rowmax_finish(const float* ws, int64_t rows, int parts, int32_t* idx, void* stream) {
  GP_REQUIRE(ws && idx && rows > 0 && parts > 0 && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, "rowmax_finish: bad args");
  const long long nb = (rows + 255) / 256;
  hipLaunchKernelGGL(gpemsr::rowmax_finish_kernel, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2*>(ws), (long long)rows, parts, idx);
  return check_launch("rowmax_finish_kernel");
}

// the kernel instantiation gpemsr_conv2d_bf16 would launch for `d`, as text (bench.py's per-kernel table); nothing is launched
extern "C" int gpemsr_conv2d_bf16_kernel_name(const gpemsr_conv16_desc* d, char* buf, int cap) {
  GP_REQUIRE(buf != nullptr && cap > 0, "conv2d_bf16_kernel_name: no buffer");
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  if (rc != GPEMSR_OK) return rc;
  if (L.gdirect) snprintf(buf, (size_t)cap, "gemm_direct_bf16_kernel<256x256,%s>", d->rowmax ? "ROWMAX" : (L.lean ? "LEAN" : "GENERAL"));
  else if (L.tres) snprintf(buf, (size_t)cap, "convt64_resident_kernel");
  else if (L.resident) {
    if (L.res_form == 2) snprintf(buf, (size_t)cap, "conv64_resident2_kernel<%s,%s>", (L.lean || L.axf) ? "true" : "false", L.axf ? "true" : "false");
    else snprintf(buf, (size_t)cap, "conv64_resident_kernel<%s>", L.lean ? "true" : "false");
  } else {
    snprintf(buf, (size_t)cap, "conv_bf16_kernel<CK=%d,BN=%d,WM=%d,WN=%d,TH=%d,TPS=%d,%s,NL=%d%s>", L.CK, L.BN, L.WM, L.WN, L.TH, L.TPS,
             L.tr ? (L.tcomp ? "CONVT,TCOMP" : "CONVT") : (L.gemm ? (d->rowmax ? "GEMM,ROWMAX" : "GEMM") : (d->stride == 2 ? "S2" : "CONV")), L.NL, L.axf ? ",AXF" : "");
  }
  return GPEMSR_OK;
}

extern "C" int gpemsr_conv2d_bf16(const gpemsr_conv16_desc* d, void* stream) {
  XParams P{}; XPlan L{};
  const int rc = plan_x(d, P, L);
  if (rc != GPEMSR_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = L.lds;
  if (L.tres) return launch_convt64_resident(P, lds, st);
  if (L.gdirect) return launch_gemm_direct(P, L.lean && !d->rowmax, d->rowmax != nullptr, lds, st);
  if (L.resident) {
    static dev_once_t attr{0};
    if (dev_once_begin(attr)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident2_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "conv2d_bf16: cannot raise the dynamic LDS limit");
      dev_once_done(attr);
    }
    // one workgroup per CU; workgroups of a cout slab split the spatial tiles between them
    const int cus = device_cus();
    int gpt = cus / P.tiles_n;
    if (gpt < 1) gpt = 1;
    if (gpt > P.ns) gpt = P.ns;
    P.gpt = gpt;
    P.dbg = d->variant;
    if (L.res_form == 2) {
      if (L.axf) hipLaunchKernelGGL((conv64_resident2_kernel<true, true>), dim3(gpt * P.tiles_n), dim3(768), lds, st, P);
      else if (L.lean) hipLaunchKernelGGL(conv64_resident2_kernel<true>, dim3(gpt * P.tiles_n), dim3(768), lds, st, P);
      else hipLaunchKernelGGL(conv64_resident2_kernel<false>, dim3(gpt * P.tiles_n), dim3(768), lds, st, P);
      return check_launch("conv64_resident2_kernel");
    }
    if (L.lean) hipLaunchKernelGGL(conv64_resident_kernel<true>, dim3(gpt * P.tiles_n), dim3(512), lds, st, P);
    else hipLaunchKernelGGL(conv64_resident_kernel<false>, dim3(gpt * P.tiles_n), dim3(512), lds, st, P);
    return check_launch("conv64_resident_kernel");
  }
#define GP_X(BNv, WMv, WNv, THv, TPSv, TRv, GEMMv) \
  (L.CK == 32 ? launch_x<32, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv>(P, lds, st) : launch_x<16, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv>(P, lds, st))
#define GP_IS(BNv, WMv, WNv, THv, TPSv) (L.BN == BNv && L.WM == WMv && L.WN == WNv && L.TH == THv && L.TPS == TPSv)
#define GP_XL(BNv, WMv, WNv, THv, TPSv, TRv, GEMMv) \
  (L.CK == 32 ? launch_x<32, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv, 4>(P, lds, st) : launch_x<16, BNv, WMv, WNv, THv, TPSv, TRv, GEMMv, 4>(P, lds, st))
  if (L.axf) return launch_x<32, 64, 8, 1, 16, 9, false, false, 4, true, 0, true>(P, lds, st);      // (plan_x admitted exactly this tile)
  if (L.NL == 8) {
    if (!L.tr && !L.gemm && GP_IS(32, 8, 1, 16, 7) && L.CK == 32)
      return L.lean ? launch_x<32, 32, 8, 1, 16, 7, false, false, 8, true>(P, lds, st) : launch_x<32, 32, 8, 1, 16, 7, false, false, 8, false>(P, lds, st);
    return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: no 8-loader kernel for this tile");
  }
  if (L.lean) {               // lean-epilogue instantiations of the hottest tiles (anything else falls through to the general ones)
    if (L.NL == 4 && !L.tr && !L.gemm && GP_IS(64, 8, 1, 16, 9))
      return L.CK == 32 ? launch_x<32, 64, 8, 1, 16, 9, false, false, 4, true>(P, lds, st) : launch_x<16, 64, 8, 1, 16, 9, false, false, 4, true>(P, lds, st);
    if (L.NL == 4 && !L.tr && !L.gemm && GP_IS(64, 8, 1, 16, 3))
      return L.CK == 32 ? launch_x<32, 64, 8, 1, 16, 3, false, false, 4, true>(P, lds, st) : launch_x<16, 64, 8, 1, 16, 3, false, false, 4, true>(P, lds, st);
    if (L.NL == 4 && L.gemm && GP_IS(128, 4, 2, 8, 1) && L.CK == 64) return launch_x<64, 128, 4, 2, 8, 1, false, true, 4, true>(P, lds, st);
    if (L.NL == 0 && !L.tr && !L.gemm && GP_IS(32, 8, 1, 16, 7) && L.CK == 32) return launch_x<32, 32, 8, 1, 16, 7, false, false, 0, true>(P, lds, st);
    if (L.NL == 0 && !L.tr && !L.gemm && GP_IS(64, 8, 1, 16, 7) && L.CK == 32) return launch_x<32, 64, 8, 1, 16, 7, false, false, 0, true>(P, lds, st);
    if (L.NL == 4 && !L.tr && !L.gemm && GP_IS(32, 8, 1, 16, 7) && L.CK == 16) return launch_x<16, 32, 8, 1, 16, 7, false, false, 4, true>(P, lds, st);
  }
  if (d->rowmax) {            // row maxima instead of a stored result: the 256 x 128 GEMM tile with 64-deep stages (the logits GEMM)
    if (L.NL == 4 && L.gemm && GP_IS(128, 4, 2, 8, 1) && L.CK == 64) return launch_x<64, 128, 4, 2, 8, 1, false, true, 4, false, 0, false, false, 1>(P, lds, st);
    return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: row maxima need the 1x1 form with cout > 64 and 64-channel sources");
  }
  if (L.NL == 4) {            // loader-wave forms
    if (L.tr && L.tcomp)
      return L.lean ? launch_x<32, 128, 8, 1, 8, 4, true, false, 4, true, 0, false, true>(P, lds, st)
                    : launch_x<32, 128, 8, 1, 8, 4, true, false, 4, false, 0, false, true>(P, lds, st);
    if (L.tr && L.TPS == 4 && L.lean) return launch_x<16, 128, 8, 1, 8, 4, true, false, 4, true>(P, lds, st);
    if (L.tr && L.TPS == 4) return launch_x<16, 128, 8, 1, 8, 4, true, false, 4>(P, lds, st);
    if (L.tr) return GP_XL(128, 8, 1, 8, 2, true, false);
    if (L.gemm) {
      if (GP_IS(128, 4, 2, 8, 1) && L.CK == 64) return launch_x<64, 128, 4, 2, 8, 1, false, true, 4>(P, lds, st);
      if (GP_IS(128, 4, 2, 8, 1)) return GP_XL(128, 4, 2, 8, 1, false, true);
      if (GP_IS(64, 8, 1, 16, 1)) return GP_XL(64, 8, 1, 16, 1, false, true);
    }
    else if (GP_IS(128, 4, 2, 4, 3) && L.CK == 32) return launch_x<32, 128, 4, 2, 4, 3, false, false, 4, false, 2>(P, lds, st);
    else if (GP_IS(64, 4, 2, 4, 3) && L.CK == 32) return launch_x<32, 64, 4, 2, 4, 3, false, false, 4, false, 2>(P, lds, st);
    else if (GP_IS(64, 4, 2, 4, 9) && L.CK == 32)
      return L.lean ? launch_x<32, 64, 4, 2, 4, 9, false, false, 4, true, 2>(P, lds, st) : launch_x<32, 64, 4, 2, 4, 9, false, false, 4, false, 2>(P, lds, st);
    else if (GP_IS(64, 8, 1, 16, 9)) return GP_XL(64, 8, 1, 16, 9, false, false);
    else if (GP_IS(64, 8, 1, 16, 3)) return GP_XL(64, 8, 1, 16, 3, false, false);
    else if (GP_IS(32, 8, 1, 16, 3)) return GP_XL(32, 8, 1, 16, 3, false, false);
    else if (GP_IS(64, 8, 1, 16, 7) && L.CK == 16) return launch_x<16, 64, 8, 1, 16, 7, false, false, 4>(P, lds, st);
    else if (GP_IS(32, 8, 1, 16, 7) && L.CK == 16) return launch_x<16, 32, 8, 1, 16, 7, false, false, 4>(P, lds, st);
    return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: no loader-wave kernel for BN=%d WM=%d WN=%d TH=%d TPS=%d", L.BN, L.WM, L.WN, L.TH, L.TPS);
  }
  if (L.tr) return GP_X(128, 4, 1, 4, 2, true, false);
  if (L.gemm) {
    if (GP_IS(32, 4, 1, 4, 1)) return GP_X(32, 4, 1, 4, 1, false, true);
    if (GP_IS(64, 2, 2, 4, 1)) return GP_X(64, 2, 2, 4, 1, false, true);
    return GP_X(128, 2, 2, 4, 1, false, true);
  }
  if (GP_IS(64, 8, 1, 16, 3)) return GP_X(64, 8, 1, 16, 3, false, false);
  if (GP_IS(64, 8, 1, 16, 7)) return GP_X(64, 8, 1, 16, 7, false, false);
  if (GP_IS(32, 8, 1, 16, 7)) return GP_X(32, 8, 1, 16, 7, false, false);
  if (GP_IS(32, 4, 1, 4, 7)) return GP_X(32, 4, 1, 4, 7, false, false);
  if (GP_IS(64, 2, 2, 4, 7)) return GP_X(64, 2, 2, 4, 7, false, false);
  if (GP_IS(64, 2, 2, 2, 3)) return GP_X(64, 2, 2, 2, 3, false, false);
  if (GP_IS(128, 2, 2, 2, 1)) return GP_X(128, 2, 2, 2, 1, false, false);
  if (GP_IS(32, 4, 1, 8, 3)) return GP_X(32, 4, 1, 8, 3, false, false);
  if (GP_IS(32, 4, 1, 8, 1)) return GP_X(32, 4, 1, 8, 1, false, false);
  if (GP_IS(64, 4, 1, 8, 3)) return GP_X(64, 4, 1, 8, 3, false, false);
  if (GP_IS(64, 4, 1, 8, 1)) return GP_X(64, 4, 1, 8, 1, false, false);
  if (GP_IS(128, 2, 2, 4, 3)) return GP_X(128, 2, 2, 4, 3, false, false);
  if (GP_IS(128, 4, 1, 8, 1)) return GP_X(128, 4, 1, 8, 1, false, false);
  return fail(GPEMSR_EUNSUPPORTED, "conv2d_bf16: no kernel for BN=%d WM=%d WN=%d TH=%d TPS=%d", L.BN, L.WM, L.WN, L.TH, L.TPS);
#undef GP_X
#undef GP_XL
#undef GP_IS
}

#ifdef GP16_STAMP
extern "C" int gpemsr_debug_read_xstamps(unsigned long long* host, int nblocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gpemsr::g_xstamps), sizeof(unsigned long long) * 8 * (size_t)nblocks);
}
#endif
