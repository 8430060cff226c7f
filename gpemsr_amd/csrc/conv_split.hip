// 3x3 / 7x7 stride-1 convolution on the bf16 matrix pipe with fp32-grade accuracy ("split-bf16"):
// every fp32 operand x is represented as hi + lo with hi = bf16(x), lo = bf16(x - hi) and the product is
// evaluated as hi*hi + hi*lo + lo*hi with three v_mfma_f32_32x32x16_bf16 into an fp32 accumulator (the dropped
// lo*lo term is 2^-16 relative).  NSPLIT = 1 keeps only hi*hi (plain bf16 inputs, fp32 accumulate).
// 16x the fp32 MFMA rate / 3 products = 5.3x less matrix-pipe time than conv_mfma.hip for the same layer.
//
// Same tiling and data flow as conv_mfma.hip (activations stay fp32 NHWC in HBM):
//   * the fp32 halo tile of a 16-channel chunk arrives by LDS-DMA (raw image), is split ONCE into hi/lo bf16
//     images by a cooperative pass and is then reused by the 3 taps of each of the 3 filter-row stages.  Image layout
//     [k-half][pixel][8 bf16]: a lane's MFMA fragment (8 consecutive k of one pixel) is one ds_read_b128 and lanes
//     0-31 (k-half 0) / 32-63 (k-half 1) each read 512 contiguous bytes.  ([pixel][16 bf16] rows put the 32 lanes of a
//     half on a 32-B stride = 2-way bank conflict in every ds_read_b128 lane group: SQ_LDS_BANK_CONFLICT was 46 % of
//     SQ_LDS_IDX_ACTIVE; the half-plane layout took the 256->256 layer from 308 to 336-351 TF algorithmic);
//   * weights are split on the host (gpemsr_amd/packing.py::pack_conv_split) into [plane][cin/16][tap][k-half][cout][8]
//     bf16 -- the order they are staged in, so every LDS-DMA instruction reads 1 KiB of consecutive global memory and
//     lands as the conflict-free [tap][k-half][BN][8] image -- into a 2-deep ring;
//   * wave tile 64 pixels x 64 (or 32) couts, MFMA operand maps of guide section 3 (A[row = lane&31][k = 8*(lane>>5)+j]);
//   * epilogue identical to conv_mfma.hip (LDS-staged coalesced float4 rows, batched residual loads, PixelShuffle).
// Parity: tests/test_ops_gpu.py::test_conv_split_* (<= 3e-5 relative vs fp32 torch for NSPLIT = 2).
#include "common.h"
#include <stdlib.h>

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int SW = 32;            // tile width (pixels)
constexpr int SA_LOADS = 6;       // raw-A float4 slots per thread (10x34 halo x 4 / 256 -> 6; GEMM: 128 px x 8 / 256 -> 4)
constexpr int SB_LOADS = 4;       // weight 16-B slots per thread per plane (3 taps x 128 couts x 2 / 256 -> 3; 7 taps x 64 x 2 / 256 -> 4)

struct SplitParams {
  const float* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w, oh, ow, OH, OW;
  int cin_pad, cout;
  const unsigned short* weight;        // [plane][cin/16][tap][k-half][cout][8] bf16
  long long plane_stride;              // elements between the hi and lo planes
  long long w_img_stride;              // elements between the weights of consecutive images (0: shared; GEMM with per-image B)
  int sub_plane_bytes;                 // GEMM: bytes of one 16-channel sub-chunk image (both k-halves)
  const float* bias; int act;
  const float* residual; int res_ld;
  const float* pixmul;
  int store_mode, cq;
  float* out; int out_ld;
  int out_vec, res_vec;
  int tiles_x, tiles_y, tiles_n;
  int halo_h, halo_w, pad;
  int tw_lg;                           // log2 of the tile width: 5 (TH x 32 pixels) or 4 (2*TH x 16, maps <= 16 wide), as conv_mfma.hip
  int na, nb;                          // DMA slots per thread: raw A image, one weight plane of one stage
  int raw_bytes, sp_plane_bytes, b_plane_bytes;
  int nblocks;
};

__device__ __forceinline__ void sglds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

// two fp32 -> packed bf16 pair (lo half = a, hi half = b), round to nearest even: one v_cvt_pk_bf16_f32
typedef unsigned bf16x2_t;
__device__ __forceinline__ bf16x2_t cvt_pk_bf16(float a, float b) {
  bf16x2_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// CONVT: ConvTranspose2d(k3,s2,p1,op1) in the phase-stacked 2x2-tap form of conv_mfma.hip (KW = 2, BN = 128 = 4 phases x 32
// couts per wave, tap (dy,dx) feeds the phases with py >= dy, px >= dx; masked N tiles are neither read nor multiplied).
// GEMM: 1x1 convolution / Linear / batched matrix product (per-image B through w_img_stride).  There is no halo and no
// tap reuse, so a stage covers KW 16-channel sub-chunks instead of KW filter taps: the raw image is [pixel][16*KW fp32]
// (one contiguous 64*KW-byte run per pixel), the split images are [sub-chunk][k-half][pixel][8 bf16], and "tap" kx of the
// inner loop selects sub-chunk kx at the SAME pixel.  The staged weight image [sub-chunk][k-half][BN][8] is KW consecutive
// chunks of the global [cin/16][1][k-half][cout][8] order.
template <int BN, int WM, int WN, int TH, int NSPLIT, int KW, bool CONVT, bool GEMM = false>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(SplitParams P) {
  constexpr int NPIX = TH * SW;
  constexpr int PM = NPIX / WM;
  constexpr int MT = PM / 32;
  constexpr int WNT = BN / WN;
  constexpr int NT = WNT / 32;
  constexpr int G = GEMM ? 1 : KW;    // conv: KW taps per stage (one filter row), G = KW stages per chunk
  constexpr int CK = GEMM ? 16 * KW : 16;   // channels per chunk
  constexpr int PCS = CK / 4;         // 16-byte pieces per pixel of a raw chunk image

  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* const raw = smem_c;                                   // fp32 [halo_px][16]
  char* const asp = smem_c + P.raw_bytes;                     // NSPLIT planes of bf16 [halo_px][16]
  char* const bring = asp + NSPLIT * P.sp_plane_bytes;        // 2 ring slots x NSPLIT planes of bf16 [tap][BN][16]
  const int b_slot_bytes = NSPLIT * P.b_plane_bytes;

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
  const int oy0 = ty * (NPIX >> tw_lg), ox0 = tx << tw_lg, n0 = tn * BN;
  const int iy0 = oy0 - P.pad, ix0 = ox0 - P.pad;
  const int halo_px = P.halo_h * P.halo_w;

  // ---- per-thread staging slots ----
  int a_pix[SA_LOADS];          // pixel index inside the image or -1 (also -1 for slots past the image)
  bool a_slot[SA_LOADS];        // slot exists (e < halo_px*4)
#pragma unroll
  for (int i = 0; i < SA_LOADS; ++i) {
    const int e = tid + i * 256;
    a_pix[i] = -1; a_slot[i] = false;
    if (i < P.na && e < halo_px * PCS) {
      a_slot[i] = true;
      const int hp = e / PCS;
      const int iy = iy0 + hp / P.halo_w, ix = ix0 + hp % P.halo_w;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pix[i] = iy * P.w + ix;
    }
  }
  int b_goff[SB_LOADS];         // element offset (bf16) inside a weight plane, filter row 0, chunk 0; or -1
#pragma unroll
  for (int i = 0; i < SB_LOADS; ++i) {
    const int e = tid + i * 256;
    b_goff[i] = -1;
    if (i < P.nb && e < KW * BN * 2) {
      const int tt = e / (2 * BN), j = (e / BN) & 1, nn = e % BN;      // LDS image [tap][k-half][BN][8 bf16]
      if (n0 + nn < P.cout) b_goff[i] = ((tt * 2 + j) * P.cout + n0 + nn) * 8;       // global [chunk][tap][k-half][cout][8]
    }
  }
  int na_w = 0, nb_w = 0;
#pragma unroll
  for (int i = 0; i < SA_LOADS; ++i) na_w += (__ballot(a_pix[i] >= 0) != 0ull) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < SB_LOADS; ++i) nb_w += (__ballot(b_goff[i] >= 0) != 0ull) ? 1 : 0;
  (void)na_w; (void)nb_w;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_c + (unsigned)wave * 1024u);
  const unsigned lds_b0 = lds0 + (unsigned)(P.raw_bytes + NSPLIT * P.sp_plane_bytes);

  // zero-fill the slots no DMA ever writes (out-of-image halo pixels, rows past cout, padding)
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < SA_LOADS; ++i)
      if (i < P.na && a_pix[i] < 0) *reinterpret_cast<float4*>(raw + (tid + i * 256) * 16) = z;
#pragma unroll
    for (int i = 0; i < SB_LOADS; ++i)
      if (i < P.nb && b_goff[i] < 0)
        for (int sl = 0; sl < 2; ++sl)
          for (int pl = 0; pl < NSPLIT; ++pl) *reinterpret_cast<float4*>(bring + sl * b_slot_bytes + pl * P.b_plane_bytes + (tid + i * 256) * 16) = z;
  }

  // ---- chunk / stage cursors ----
  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / CK;
  const int nstages = nchunks * G;
  int an_src = 0, an_c0 = 0;                                // A cursor: chunk whose raw image is issued next
  int bn_c0 = 0, bn_cpad = 0, bn_src = 0, bn_grp = 0, bn_stage = 0;   // B cursor: next stage to issue

  auto issue_a = [&]() {
    const float* sp = P.src[an_src] + (long long)img * P.img_stride[an_src] + an_c0;
    const unsigned pixb = (unsigned)P.ld[an_src] * 4u;
#pragma unroll
    for (int i = 0; i < SA_LOADS; ++i)
      if (a_pix[i] >= 0) sglds16((unsigned)a_pix[i] * pixb + 16u * (unsigned)((tid + i * 256) % PCS), sp, lds0 + i * 4096u);
    an_c0 += CK;
    if (an_c0 >= P.c[an_src] && an_src + 1 < P.nsrc) { an_c0 = 0; ++an_src; }
  };
  auto issue_b = [&]() {
    const unsigned short* wp = P.weight + (long long)img * P.w_img_stride +
                               (GEMM ? (long long)((bn_cpad + bn_c0) >> 4) : ((long long)((bn_cpad + bn_c0) >> 4) * (G * KW) + bn_grp * KW)) * (16 * P.cout);
    const unsigned lb = lds_b0 + (unsigned)((bn_stage & 1) * b_slot_bytes);
#pragma unroll
    for (int pl = 0; pl < NSPLIT; ++pl) {
      const unsigned short* wpp = wp + pl * P.plane_stride;
#pragma unroll
      for (int i = 0; i < SB_LOADS; ++i)
        if (b_goff[i] >= 0) sglds16((unsigned)b_goff[i] * 2u, wpp, lb + (unsigned)(pl * P.b_plane_bytes) + i * 4096u);
    }
    ++bn_stage;
    if (++bn_grp == G) {
      bn_grp = 0; bn_c0 += CK;
      if (bn_c0 >= P.c[bn_src] && bn_src + 1 < P.nsrc) { bn_cpad += P.c[bn_src]; bn_c0 = 0; ++bn_src; }
    }
  };
  // raw fp32 image -> hi (+ lo) bf16 images; 4 channels per slot -> one 8-byte store per plane
  auto split_pass = [&]() {
#pragma unroll
    for (int i = 0; i < SA_LOADS; ++i) {
      if (a_slot[i]) {
        const int e = tid + i * 256;
        const float4 v = *reinterpret_cast<const float4*>(raw + e * 16);
        // hi = bf16(x) (RNE, v_cvt_pk_bf16_f32), lo = bf16(x - hi)
        const bf16x2_t h01 = cvt_pk_bf16(v.x, v.y), h23 = cvt_pk_bf16(v.z, v.w);
        const int q = e % PCS, hp = e / PCS;                     // q: 4-channel piece of the pixel's chunk
        const int off = GEMM ? (q >> 2) * P.sub_plane_bytes + ((q >> 1) & 1) * (P.sub_plane_bytes >> 1) + hp * 16 + (q & 1) * 8
                             : (q >> 1) * (P.sp_plane_bytes >> 1) + hp * 16 + (q & 1) * 8;   // [k-half][pixel][8 bf16]
        *reinterpret_cast<uint2*>(asp + off) = make_uint2(h01, h23);
        if (NSPLIT == 2) {
          const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xFFFF0000u);
          const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xFFFF0000u);
          *reinterpret_cast<uint2*>(asp + P.sp_plane_bytes + off) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
        }
      }
    }
  };

  // ---- fragment offsets (bytes) ----
  int a_frag[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = wm * PM + mt * 32 + li;
    a_frag[mt] = ((p >> tw_lg) * P.halo_w + (p & tw_mask)) * 16 + lh * ((GEMM ? P.sub_plane_bytes : P.sp_plane_bytes) >> 1);
  }
  const int b_frag = (wn * WNT + li) * 16 + lh * (BN * 16);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- prologue: chunk 0 raw + stage 0 weights, split ----
  issue_a();
  issue_b();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  split_pass();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int chunk = 0, grp = 0;
  for (int stage = 0; stage < nstages; ++stage) {
    if (stage + 1 < nstages) issue_b();                        // next stage's weights -> other ring slot
    if (grp == 0 && chunk + 1 < nchunks) issue_a();            // next chunk's raw image (the raw buffer is free: split(chunk) is done)

    const char* A = asp + (grp * P.halo_w) * 16;               // filter row `grp`
    const char* B = bring + (stage & 1) * b_slot_bytes + b_frag;
    bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
    auto tap_mask = [&](int kx) -> unsigned {      // CONVT: N tiles (phases q = 2py+px) fed by tap (dy = grp, dx = kx)
      if (!CONVT) return 0xFu;
      return grp ? (kx ? 0x8u : 0xCu) : (kx ? 0xAu : 0xFu);
    };
    auto load_tap = [&](int set, int kx) {
      const unsigned mask = tap_mask(kx);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[set][mt] = *reinterpret_cast<const bf16x8*>(A + a_frag[mt] + kx * (GEMM ? P.sub_plane_bytes : 16));
        if (NSPLIT == 2) al[set][mt] = *reinterpret_cast<const bf16x8*>(A + P.sp_plane_bytes + a_frag[mt] + kx * (GEMM ? P.sub_plane_bytes : 16));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (CONVT && !((mask >> nt) & 1u)) continue;
        bh[set][nt] = *reinterpret_cast<const bf16x8*>(B + kx * BN * 32 + nt * 512);
        if (NSPLIT == 2) bl[set][nt] = *reinterpret_cast<const bf16x8*>(B + P.b_plane_bytes + kx * BN * 32 + nt * 512);
      }
    };
    auto mma_tap = [&](int set, int kx) {
      const unsigned mask = tap_mask(kx);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (CONVT && !((mask >> nt) & 1u)) continue;
          if (NSPLIT == 2) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][mt], bh[set][nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][mt], bl[set][nt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][mt], bh[set][nt], acc[mt][nt], 0, 0, 0);
        }
    };
    load_tap(0, 0);
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {          // software pipelined by two (static register sets)
      if (kx + 1 < KW) load_tap((kx + 1) & 1, kx + 1);
      mma_tap(kx & 1, kx);
    }

    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __builtin_amdgcn_s_barrier();                                                // ... and everybody's; all reads of this stage done
    asm volatile("" ::: "memory");
    if (++grp == G) {
      grp = 0; ++chunk;
      if (chunk < nchunks) {            // the raw image of the new chunk landed during its predecessor: split it now
        split_pass();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }

  // ---- epilogue (same as conv_mfma.hip) ----
  constexpr int EW = BN < 64 ? BN : 64;
  constexpr int EPIX = EW + 4;
  constexpr int NCP = BN / EW;
  constexpr int NPP = NPIX / 128;
  constexpr int NV = EW / 4;
  float* E = reinterpret_cast<float*>(smem_c);
  __syncthreads();
  constexpr int ITER = (128 * NV) / 256;
  constexpr int PSTEP = 256 / NV;
  const int ej = tid % NV, ep0 = tid / NV;
  const long long img_pix0 = (long long)img * P.OH * P.OW;
  const float* res_img = P.residual ? P.residual + img_pix0 * P.res_ld : nullptr;
  const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
  float* out_img = P.out + img_pix0 * P.out_ld;
#pragma unroll 1
  for (int pass = 0; pass < NCP * NPP; ++pass) {
    const int cpass = pass % NCP, ppass = pass / NCP;
    const int nidx = n0 + cpass * EW + 4 * ej;
    int ch = nidx, bidx = nidx, sy = 0, sx = 0;
    if (P.store_mode == 1) { const int q = nidx / P.cq; ch = nidx - q * P.cq; sy = q >> 1; sx = q & 1; }
    else if (P.store_mode == 2) { const int blk = nidx >> 7, q = (nidx & 127) >> 5; ch = blk * 32 + (nidx & 31); bidx = ch; sy = q >> 1; sx = q & 1; }
    const int nvalid = (P.cout - nidx) < 4 ? (P.cout - nidx) : 4;
    const bool full = nvalid == 4;
    int opix[ITER];
    float4 rres[ITER];
    float rmul[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int p = ppass * 128 + ep0 + it * PSTEP;
      const int oy = oy0 + (p >> tw_lg), ox = ox0 + (p & tw_mask);
      opix[it] = -1; rres[it] = make_float4(0.f, 0.f, 0.f, 0.f); rmul[it] = 1.f;
      if (oy < P.oh && ox < P.ow && nvalid > 0) {
        const int Y = P.store_mode == 0 ? oy : 2 * oy + sy, X = P.store_mode == 0 ? ox : 2 * ox + sx;
        opix[it] = Y * P.OW + X;
        if (P.residual) {
          const float* rp = res_img + (long long)opix[it] * P.res_ld + ch;
          if (P.res_vec && full) rres[it] = *reinterpret_cast<const float4*>(rp);
          else {
            if (nvalid > 0) rres[it].x = rp[0];
            if (nvalid > 1) rres[it].y = rp[1];
            if (nvalid > 2) rres[it].z = rp[2];
            if (nvalid > 3) rres[it].w = rp[3];
          }
        }
        if (P.pixmul) rmul[it] = mul_img[opix[it]];
      }
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < nvalid) bv[k] = P.bias[bidx + k];
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
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (opix[it] < 0) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(E + (ep0 + it * PSTEP) * EPIX + 4 * ej);
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k] + bv[k], P.act);
      v[0] += rres[it].x; v[1] += rres[it].y; v[2] += rres[it].z; v[3] += rres[it].w;
      v[0] *= rmul[it]; v[1] *= rmul[it]; v[2] *= rmul[it]; v[3] *= rmul[it];
      float* op = out_img + (long long)opix[it] * P.out_ld + ch;
      if (P.out_vec && full) {
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nvalid) op[k] = v[k];
      }
    }
  }
}

template <int BN, int WM, int WN, int TH, int NSPLIT, int KW, bool CONVT = false, bool GEMM = false>
static int launch_split(const SplitParams& P, size_t lds, hipStream_t st) {
  auto kfn = conv_split_kernel<BN, WM, WN, TH, NSPLIT, KW, CONVT, GEMM>;
  if (lds > 64 * 1024) {
    static dev_once_t done{0};
    if (dev_once_begin(done)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(GPEMSR_ELAUNCH, "conv2d_split: cannot raise the dynamic LDS limit");
      dev_once_done(done);
    }
  }
  hipLaunchKernelGGL(kfn, dim3(P.nblocks), dim3(256), lds, st, P);
  return check_launch("conv_split_kernel");
}


// fp32 rows [n][rows][K] -> the split kernel's staged weight order [n][plane (hi, lo)][K/16][k-half][rows][8] bf16: the B
// operand of a batched matrix product (attention: k for q.k^T, v^T for P.v) is produced at run time, so it is split and
// re-ordered by this pass instead of on the host.  One thread = 8 consecutive k of one row.
__global__ __launch_bounds__(256) void split_pack_rows_kernel(const float* src, int rows, int K, int ld, long long img_stride,
                                                              unsigned short* dst, long long total) {
  const int kb_n = K >> 3;
  const long long plane = (long long)rows * K;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx % rows);
    const long long t = idx / rows;
    const int kb = (int)(t % kb_n);
    const long long img = t / kb_n;
    const float* sp = src + img * img_stride + (long long)r * ld + kb * 8;
    const float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
    const bf16x2_t h0 = cvt_pk_bf16(a.x, a.y), h1 = cvt_pk_bf16(a.z, a.w), h2 = cvt_pk_bf16(b.x, b.y), h3 = cvt_pk_bf16(b.z, b.w);
    const bf16x2_t l0 = cvt_pk_bf16(a.x - __uint_as_float(h0 << 16), a.y - __uint_as_float(h0 & 0xFFFF0000u));
    const bf16x2_t l1 = cvt_pk_bf16(a.z - __uint_as_float(h1 << 16), a.w - __uint_as_float(h1 & 0xFFFF0000u));
    const bf16x2_t l2 = cvt_pk_bf16(b.x - __uint_as_float(h2 << 16), b.y - __uint_as_float(h2 & 0xFFFF0000u));
    const bf16x2_t l3 = cvt_pk_bf16(b.z - __uint_as_float(h3 << 16), b.w - __uint_as_float(h3 & 0xFFFF0000u));
    unsigned short* dp = dst + img * 2 * plane + ((long long)kb * rows + r) * 8;      // (kb>>1, kb&1) = (chunk, k-half)
    *reinterpret_cast<uint4*>(dp) = make_uint4(h0, h1, h2, h3);
    *reinterpret_cast<uint4*>(dp + plane) = make_uint4(l0, l1, l2, l3);
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_split_pack_rows(const float* src, int n, int rows, int k, int ld, int64_t img_stride, void* dst_bf16, void* stream) {
  GP_REQUIRE(src && dst_bf16 && n > 0 && rows > 0 && k > 0 && k % 16 == 0 && ld % 4 == 0 && img_stride % 4 == 0,
             "split_pack_rows: bad args (k %% 16 == 0, ld %% 4 == 0)");
  GP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_bf16)) & 15) == 0, "split_pack_rows: alignment");
  const long long total = (long long)n * rows * (k / 8);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(split_pack_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, rows, k, ld, (long long)img_stride,
                     reinterpret_cast<unsigned short*>(dst_bf16), total);
  return check_launch("split_pack_rows");
}

extern "C" int gpemsr_conv2d_split(const gpemsr_conv_desc* d, const void* weight_bf16, int64_t plane_stride, int nsplit, void* stream) {
  GP_REQUIRE(d && weight_bf16 && d->out, "conv2d_split: null pointer");
  GP_REQUIRE(nsplit == 1 || nsplit == 2, "conv2d_split: nsplit=%d", nsplit);
  const bool tr = d->transposed != 0;
  const bool gemm = !tr && d->ksize == 1;
  GP_REQUIRE(d->weight_image_stride == 0 || gemm, "conv2d_split: per-image weights only for 1x1 (GEMM)");
  if (tr) GP_REQUIRE(d->ksize == 3 && !d->pixel_shuffle && !d->pixmul && d->cout % 32 == 0, "conv2d_split: transposed needs k=3, cout%%32==0");
  else GP_REQUIRE((d->ksize == 1 || d->ksize == 3 || d->ksize == 7) && d->stride == 1, "conv2d_split: only 1x1 / 3x3 / 7x7 stride-1 convolutions (or transposed 3x3)");
  if (gemm) GP_REQUIRE(!d->pixel_shuffle, "conv2d_split: 1x1 has no pixel_shuffle form");
  if (d->ksize == 7) GP_REQUIRE(!d->pixel_shuffle, "conv2d_split: 7x7 has no pixel_shuffle form");
  GP_REQUIRE(d->nsrc >= 1 && d->nsrc <= GPEMSR_MAX_SRC && d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "conv2d_split: bad geometry");
  if (d->pixel_shuffle) GP_REQUIRE(d->cout % 16 == 0, "conv2d_split: pixel_shuffle needs cout%%16==0");
  SplitParams P{};
  int cin_pad = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].c % (gemm ? 32 : 16) == 0 && d->src[s].ld % 4 == 0 &&
               ((reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0), "conv2d_split: source %d must have c%%16==0 (1x1: c%%32==0) and 16-B aligned rows", s);
    P.src[s] = d->src[s].ptr; P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    GP_REQUIRE(P.img_stride[s] % 4 == 0 && (long long)d->h * d->w * P.ld[s] * 4 < (1ll << 32), "conv2d_split: source %d too large / misaligned", s);
    cin_pad += d->src[s].c;
  }
  GP_REQUIRE((reinterpret_cast<uintptr_t>(weight_bf16) & 15) == 0 && plane_stride % 8 == 0, "conv2d_split: weight alignment");
  GP_REQUIRE((long long)d->ksize * d->ksize * d->cout * cin_pad * 2 * (tr ? 2 : 1) < (1ll << 32), "conv2d_split: weight plane too large");
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w; P.oh = d->h; P.ow = d->w;
  P.OH = (d->pixel_shuffle || tr) ? 2 * d->h : d->h; P.OW = (d->pixel_shuffle || tr) ? 2 * d->w : d->w;
  P.cin_pad = cin_pad; P.cout = tr ? 4 * d->cout : d->cout;        // transposed: 4*Cout phase-stacked GEMM columns
  P.weight = reinterpret_cast<const unsigned short*>(weight_bf16); P.plane_stride = plane_stride;
  P.w_img_stride = d->weight_image_stride;      // bf16 elements (1x1 with per-image B only)
  P.bias = d->bias; P.act = d->act; P.residual = d->residual; P.res_ld = d->res_ld; P.pixmul = d->pixmul;
  P.store_mode = tr ? 2 : (d->pixel_shuffle ? 1 : 0); P.cq = d->cout / 4;
  P.out = d->out; P.out_ld = d->out_ld;
  P.out_vec = (d->out_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->out) & 15) == 0);
  P.res_vec = d->residual && (d->res_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->residual) & 15) == 0);
  const int KW = tr ? 2 : (gemm ? 2 : d->ksize);                  // GEMM: 2 sub-chunks of 16 channels per stage
  // 7x7: 32-cout blocks of 4x32 pixels keep the (larger) halo + 7-tap weight images at two workgroups per CU
  // 7x7: 32-cout blocks; plain bf16 (one plane) has room for 64-cout blocks at two workgroups per CU (the fp32 halo tile is then
  // staged and converted once per 64 couts instead of twice) -- GPEMSR_SPLIT_NO_7x7_BN64 restores 32
  static const bool no_bn64_7 = getenv("GPEMSR_SPLIT_NO_7x7_BN64") != nullptr;     // read once per process
  const bool bn64_7 = (nsplit == 1 && d->cout >= 64 && !no_bn64_7);
  const int BN = tr ? 128 : (gemm ? (d->cout <= 64 ? 64 : 128) : (KW == 7 ? (bn64_7 ? 64 : 32) : (d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128))));
  const int TH = (BN == 128 || KW == 7 || gemm) ? 4 : 8;
  P.pad = (tr || gemm) ? 0 : KW / 2;
  // narrow maps (16x16 latents of training crops): 2*TH x 16-pixel tiles instead of half-empty 32-wide ones
  const int pad32 = cdiv(P.ow, 32) * 32, pad16 = cdiv(P.ow, 16) * 16;
  const int TW = (!tr && !gemm && (pad32 - pad16) * 4 >= pad32) ? 16 : SW;
  P.tw_lg = TW == 16 ? 4 : 5;
  const int TROWS = TH * SW / TW;
  P.halo_h = gemm ? TH : TROWS + KW - 1; P.halo_w = gemm ? SW : TW + KW - 1;
  P.tiles_x = cdiv(P.ow, TW); P.tiles_y = cdiv(P.oh, TROWS); P.tiles_n = cdiv(P.cout, BN);
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d_split: grid too large");
  P.nblocks = (int)nb;
  const int halo_px = P.halo_h * P.halo_w;
  P.na = cdiv((long long)halo_px * (gemm ? 4 * KW : 4), 256);
  P.nb = cdiv((long long)KW * BN * 2, 256);
  GP_REQUIRE(P.na <= SA_LOADS && P.nb <= SB_LOADS, "conv2d_split: tile too large");
  P.raw_bytes = P.na * 4096;
  P.sub_plane_bytes = (halo_px * 32 + 255) & ~255;
  P.sp_plane_bytes = gemm ? KW * P.sub_plane_bytes : P.sub_plane_bytes;
  P.b_plane_bytes = P.nb * 4096;
  size_t lds = (size_t)P.raw_bytes + (size_t)nsplit * P.sp_plane_bytes + 2 * (size_t)nsplit * P.b_plane_bytes;
  const size_t epi = 128 * (size_t)((BN < 64 ? BN : 64) + 4) * 4;
  if (epi > lds) lds = epi;
  GP_REQUIRE(lds <= 160 * 1024, "conv2d_split: LDS %zu too large", lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (tr) return nsplit == 2 ? launch_split<128, 4, 1, 4, 2, 2, true>(P, lds, st) : launch_split<128, 4, 1, 4, 1, 2, true>(P, lds, st);
  if (gemm) {
    if (BN == 64) return nsplit == 2 ? launch_split<64, 2, 2, 4, 2, 2, false, true>(P, lds, st) : launch_split<64, 2, 2, 4, 1, 2, false, true>(P, lds, st);
    return nsplit == 2 ? launch_split<128, 2, 2, 4, 2, 2, false, true>(P, lds, st) : launch_split<128, 2, 2, 4, 1, 2, false, true>(P, lds, st);
  }
  if (KW == 7 && BN == 64) return launch_split<64, 4, 1, 4, 1, 7>(P, lds, st);
  if (KW == 7) return nsplit == 2 ? launch_split<32, 4, 1, 4, 2, 7>(P, lds, st) : launch_split<32, 4, 1, 4, 1, 7>(P, lds, st);
  if (nsplit == 2) {
    if (BN == 32) return launch_split<32, 4, 1, 8, 2, 3>(P, lds, st);
    if (BN == 64) return launch_split<64, 4, 1, 8, 2, 3>(P, lds, st);
    return launch_split<128, 2, 2, 4, 2, 3>(P, lds, st);
  }
  if (BN == 32) return launch_split<32, 4, 1, 8, 1, 3>(P, lds, st);
  if (BN == 64) return launch_split<64, 4, 1, 8, 1, 3>(P, lds, st);
  return launch_split<128, 2, 2, 4, 1, 3>(P, lds, st);
}
