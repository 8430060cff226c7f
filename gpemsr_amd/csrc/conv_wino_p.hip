// PERSISTENT form of conv_wino2_f32_kernel (conv_wino.hip): Winograd F(2x2, 3x3), 8 x 32 output pixels x 64 couts per tile, fp32.
//
// The one-tile-per-workgroup kernel pays, per tile, the round trip of its first chunks (nothing to multiply until they land) and the
// epilogue.  On the 64-channel layers a tile is only 8 chunks long (VGG relu1_2, HRconv, the up-convolutions, the fusion and feature
// blocks: ~150 ms of the fp32 step): those layers ran at 72-75 TFLOP/s executed against 102 for the 256/512-channel ones.  Here a
// workgroup walks tiles  blockIdx.x, + gridDim.x, ...  and the NEXT tile's chunks 0 and 1 are issued during the LAST TWO stages of the
// current tile, into the ring slots those stages have just freed -- their round trip runs under two stages of matrix work and the
// epilogue.  For that the epilogue's exchange buffer must not overlay slots 0 and 1: with a chunk count = 2 (mod 3) the last two chunks
// of a tile sit in slots 0 and 1, slot 2 is free from the top of the third-last stage on, and the exchange buffer -- cut to ONE row tile
// (69,632 bytes, two passes) -- lives in slot 2 + the tail of the LDS allocation.  Chunk 2 of the next tile follows after the epilogue.
// Everything else (fragment geometry, transforms, launch order, GroupNorm / patch-cosine / PixelShuffle epilogues) is the one-tile
// kernel's, which stays the form for chunk counts 0 and 1 (mod 3).  Results are bit-identical to it.
#include "conv_wino.h"

namespace gpemsr {

constexpr int WP_EOFF = 2 * W2_STAGE;                            // 87,296: ring slot 2
constexpr int WP_EHALF = 4 * 2 * 32 * W2_EPIX * 4;               // 69,632: [xi][j][32 blocks][68] floats, one row tile
constexpr int WP_RED = WP_EOFF + WP_EHALF;                       // 156,928: cross-wave sums (GroupNorm / patch cosine), 4 KB
constexpr int WP_LDS = WP_RED + 4096;                            // 161,024

struct WpGeo { int img, ty0, tx0, tn; };

__global__ __launch_bounds__(512, 2) void conv_wino2p_f32_kernel(WinoParams P) {
  constexpr int NTI = 256;                                     // issuing threads: waves 0-3, one per SIMD (the ASYM form)
  constexpr int W2_NA = (W2_ASLOTS + NTI - 1) / NTI;           // 3
  constexpr int W2_NB = W2_BSLOTS / NTI;                       // 8
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int xi = wave & 3, mh = wave >> 2;
  const bool issuer = wave < 4;

  auto decode = [&](int tile) -> WpGeo {
    int t = tile;
    {
      const int nwg = P.nblocks, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
    WpGeo g;
    int tn_lo, tn_hi;
    wn_divmod(t, P.tn_group, P.mg_g, t, tn_lo);            // (host-side reciprocals: the compiler's own run-time divisions keep ~40 scalar registers live)
    wn_divmod(t, P.tiles_x, P.mg_x, t, g.tx0);
    wn_divmod(t, P.tiles_y, P.mg_y, t, g.ty0);
    wn_divmod(t, P.n, P.mg_n, tn_hi, g.img);
    g.tn = tn_hi * P.tn_group + tn_lo;
    return g;
  };

  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / 8;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wsm;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  const unsigned b_step = (unsigned)((NTI / 128) * P.cout * 8) * 4u;

  // ---- issue state: the tile whose chunks are being issued (the current one, or the next one during the last two stages) ----
  int a_pk[W2_NA];                                             // 2 * pixel + channel quad of the slot, -1: outside the image / not an issuer
  int n_issue = 0, i_img = 0;
  unsigned b_off0 = 0;
  int f_src = 0, f_c0 = 0, f_cpad = 0, f_slot = 0;
  auto setup_issue = [&](const WpGeo& g) {
    const int oy0 = g.ty0 * 8, ox0 = g.tx0 * 32, n0 = g.tn * 64;
    int na_w = 0;
    int tq = tid;                                              // opaque: the slot decode below is tile-invariant, but hoisted out of the tile loop it
    asm volatile("" : "+v"(tq));                               // would stay live across the MFMA stages (the kernel sits at the 256-register limit)
#pragma unroll
    for (int i = 0; i < W2_NA; ++i) {
      const int s = tq + i * NTI;
      a_pk[i] = -1;
      if (issuer && s < W2_ASLOTS) {
        const int hx2 = s % WN_HW2, r1 = s / WN_HW2;
        const int hy = r1 % W2_HH, r2 = r1 / W2_HH;
        const int par = r2 & 1, qd = r2 >> 1;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + 2 * hx2 + par;
        if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pk[i] = 2 * (iy * P.w + ix) + qd;
      }
      na_w += (__ballot(a_pk[i] >= 0) != 0ull) ? 1 : 0;
    }
    n_issue = issuer ? na_w + W2_NB : 0;
    i_img = g.img;
    b_off0 = (unsigned)((((tq & (NTI - 1)) >> 7) * P.cout + n0 + ((tq >> 1) & 63)) * 8 + 4 * (tq & 1)) * 4u;
    f_src = 0; f_c0 = 0; f_cpad = 0; f_slot = 0;
  };
  auto issue_chunk = [&]() {
    const unsigned la = lds0 + (unsigned)(f_slot * W2_STAGE);
    const float* sp = P.src[f_src] + (long long)i_img * P.img_stride[f_src] + f_c0;
    const unsigned pixb = (unsigned)P.ld[f_src] * 4u;
    if (issuer) {
#pragma unroll
      for (int i = 0; i < W2_NA; ++i) {
        if (a_pk[i] >= 0) wn_glds16((unsigned)(a_pk[i] >> 1) * pixb + 16u * (unsigned)(a_pk[i] & 1), sp, la + i * (NTI * 16u));
        else if (tid + i * NTI < W2_ASLOTS)                  // zero padding of THIS tile (the slot may hold another tile's pixels)
          *reinterpret_cast<float4*>(wsm + f_slot * W2_STAGE + (tid + i * NTI) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* wp = P.weight + (long long)(f_cpad + f_c0) * (16 * P.cout);      // chunk (f_cpad + f_c0) / 8 of U[cin / 8][16][cout][8]
      unsigned bo = b_off0;
      asm volatile("" : "+v"(bo));                       // opaque: eight hoisted offsets would not fit the register file
#pragma unroll
      for (int i = 0; i < W2_NB; ++i) { wn_glds16(bo, wp, la + (unsigned)W2_ABYTES + i * (NTI * 16u)); bo += b_step; }
    }
    f_c0 += 8;
    f_slot = f_slot == W2_RING - 1 ? 0 : f_slot + 1;
    if (f_c0 >= P.c[f_src] && f_src + 1 < P.nsrc) { f_cpad += P.c[f_src]; f_c0 = 0; ++f_src; }
  };

  const int aA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int aB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sB = xi == 1 ? 1.f : -1.f;
  // fragment offsets of this lane, recomputed where they are used (from an opaque copy of the lane id): two registers the stage loop does not have
  auto frag_bases = [&](unsigned& d_base, unsigned& u_base) {
    int l2 = lane;
    asm volatile("" : "+v"(l2));
    const int li_ = l2 & 31, lh_ = l2 >> 5;
    const int ty = 2 * mh + (li_ >> 4), tx = li_ & 15;
    d_base = (unsigned)((((lh_ * 2) * W2_HH + 2 * ty) * WN_HW2 + tx) * 16);
    u_base = (unsigned)(W2_ABYTES + ((4 * xi * 64 + li_) * 2 + lh_) * 16);
  };

  auto load_raw = [&](const char* st, float4 (&r4)[4], unsigned d_base) {
    const char* dp = st + d_base;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int cofs = ((b & 1) * W2_HH * WN_HW2 + (b >> 1)) * 16;
      const float4 va = *reinterpret_cast<const float4*>(dp + cofs + aA * (WN_HW2 * 16));
      const float4 vb = *reinterpret_cast<const float4*>(dp + cofs + aB * (WN_HW2 * 16));
      r4[b] = make_float4(fmaf(sB, vb.x, va.x), fmaf(sB, vb.y, va.y), fmaf(sB, vb.z, va.z), fmaf(sB, vb.w, va.w));
    }
  };
  auto load_u = [&](const char* st, float4 (&U)[4][2], unsigned u_base) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) U[nu][nt] = *reinterpret_cast<const float4*>(st + u_base + nu * (64 * 2 * 16) + nt * (32 * 2 * 16));
  };

  // ---- prologue of the first tile (host: nchunks >= 2, nchunks % 3 == 2) ----
  setup_issue(decode((int)blockIdx.x));
  issue_chunk();
  issue_chunk();
  int infl = 0;
  if (nchunks > 2) { issue_chunk(); infl = n_issue; }
  wn_wait_vmcnt(infl);                                   // chunks 0 and 1 have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float4 U_cur[4][2], r_cur[4], U_alt[4][2], r_alt[4];
  {
    unsigned d_base, u_base;
    frag_bases(d_base, u_base);
    load_u(wsm, U_cur, u_base);
    load_raw(wsm, r_cur, d_base);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave holds chunk 0's fragments: its slot may be refilled
  asm volatile("" ::: "memory");

  for (int tile = (int)blockIdx.x;;) {
    const int next_tile = tile + (int)gridDim.x;
    const bool has_next = next_tile < P.nblocks;
    f32x16 acc[4][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;

    int slot_n = 1;                                      // ring slot of chunk ck + 1
    auto stage = [&](const int ck, float4 (&Uc)[4][2], float4 (&rc)[4], float4 (&Un)[4][2], float4 (&rn)[4]) {
      int issued = 0;
      if (ck + 3 < nchunks) { issue_chunk(); issued = n_issue; }                 // -> the slot of chunk ck (its fragments are in registers)
      else if (has_next && ck == nchunks - 2) { setup_issue(decode(next_tile)); issue_chunk(); issued = n_issue; }   // next tile, chunk 0 -> slot 0
      else if (has_next && ck == nchunks - 1) { issue_chunk(); issued = n_issue; }                                    // next tile, chunk 1 -> slot 1
      const char* sn = wsm + slot_n * W2_STAGE;
      const bool more = ck + 1 < nchunks;                // (wave-uniform)
      unsigned d_base, u_base;
      frag_bases(d_base, u_base);
      if (more) {
        load_raw(sn, rn, d_base);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) Un[nu][0] = *reinterpret_cast<const float4*>(sn + u_base + nu * (64 * 2 * 16));
      }
      // position by position (the one-tile kernel runs cout tile by cout tile with all four V live: 12 registers more, which this kernel --
      // one loop around stages AND epilogue -- does not have; per accumulator the k order is the same, so the results are bit-identical)
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        float4 V;
        if (nu == 0) V = make_float4(rc[0].x - rc[2].x, rc[0].y - rc[2].y, rc[0].z - rc[2].z, rc[0].w - rc[2].w);
        else if (nu == 1) V = make_float4(rc[1].x + rc[2].x, rc[1].y + rc[2].y, rc[1].z + rc[2].z, rc[1].w + rc[2].w);
        else if (nu == 2) V = make_float4(rc[2].x - rc[1].x, rc[2].y - rc[1].y, rc[2].z - rc[1].z, rc[2].w - rc[1].w);
        else V = make_float4(rc[1].x - rc[3].x, rc[1].y - rc[3].y, rc[1].z - rc[3].z, rc[1].w - rc[3].w);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.x, Uc[nu][nt].x, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.y, Uc[nu][nt].y, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.z, Uc[nu][nt].z, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.w, Uc[nu][nt].w, acc[nu][nt], 0, 0, 0);
        }
        if (nu == 1 && more) {                           // second half of the next chunk's U fragments: under the last 16 MFMAs
#pragma unroll
          for (int n2 = 0; n2 < 4; ++n2) Un[n2][1] = *reinterpret_cast<const float4*>(sn + u_base + n2 * (64 * 2 * 16) + 32 * 2 * 16);
        }
      }
      // chunk ck + 2 must have landed before the next stage reads it: only this stage's issues may stay in flight
      wn_wait_vmcnt(issued);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      slot_n = slot_n == W2_RING - 1 ? 0 : slot_n + 1;
    };
    for (int ck = 0; ck < nchunks; ck += 2) {            // (host: nchunks is even; % 3 == 2)
      stage(ck, U_cur, r_cur, U_alt, r_alt);
      stage(ck + 1, U_alt, r_alt, U_cur, r_cur);
    }

    // ---- epilogue of `cur`, one row tile (mh) per pass through the half-size exchange buffer in slot 2 ----
    const WpGeo cur = decode(tile);                          // (decoded again here rather than carried through the stage loop: scalar registers are as tight as vector ones)
    const int oy0 = cur.ty0 * 8, ox0 = cur.tx0 * 32, n0 = cur.tn * 64;
    int tid2 = tid, lane2 = lane, li2 = li, lh2 = lh;         // opaque per-tile copies: keeps the epilogue's addressing out of the stage loop's live set
    asm volatile("" : "+v"(tid2), "+v"(lane2), "+v"(li2), "+v"(lh2));
    float* E = reinterpret_cast<float*>(wsm + WP_EOFF);
    float* red = reinterpret_cast<float*>(wsm + WP_RED);
    const long long img_pix0 = (long long)cur.img * P.h * P.w;
    const float* res_img = P.residual ? P.residual + img_pix0 * P.res_ld : nullptr;
    const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
    float* out_img = P.out + img_pix0 * (P.pixshuf ? 4 : 1) * P.out_ld;
    const int act = P.act;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};      // GroupNorm partial sums of this thread's 4 channels (both passes share the cout quad)
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      if (mh == it) {                                   // (wave-uniform) the four xi waves of row tile `it` hand over their column-combined sums
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh2;
            const float z0 = (acc[0][nt][r] + acc[1][nt][r]) + acc[2][nt][r];
            const float z1 = (acc[1][nt][r] - acc[2][nt][r]) - acc[3][nt][r];
            E[((xi * 2 + 0) * 32 + row) * W2_EPIX + nt * 32 + li2] = z0;
            E[((xi * 2 + 1) * 32 + row) * W2_EPIX + nt * 32 + li2] = z1;
          }
      }
      __syncthreads();
      const int bl = tid2 >> 4, cq = tid2 & 15;            // (block of this row tile, cout quad): 16 consecutive threads = the 256 bytes of one pixel
      const int blk = it * 32 + bl;
      const int by = blk >> 4, bx = blk & 15;
      if (P.cos_ws) {
        // R:model/GPEMSR.py:387-395 without the second relu1_2 map in memory (record layout of the one-tile kernel): strip `it` of the tile
        const int ch = 4 * cq;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ch);
        float ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 z[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) z[x] = *reinterpret_cast<const float4*>(E + ((x * 2 + j) * 32 + bl) * W2_EPIX + 4 * cq);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float4 v;
            if (i == 0) v = make_float4((z[0].x + z[1].x) + z[2].x, (z[0].y + z[1].y) + z[2].y, (z[0].z + z[1].z) + z[2].z, (z[0].w + z[1].w) + z[2].w);
            else v = make_float4((z[1].x - z[2].x) - z[3].x, (z[1].y - z[2].y) - z[3].y, (z[1].z - z[2].z) - z[3].z, (z[1].w - z[2].w) - z[3].w);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            if (act == GPEMSR_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (act == GPEMSR_ACT_LRELU) { v.x = fmaxf(v.x, 0.1f * v.x); v.y = fmaxf(v.y, 0.1f * v.y); v.z = fmaxf(v.z, 0.1f * v.z); v.w = fmaxf(v.w, 0.1f * v.w); }
            const long long pix = (long long)(oy0 + 2 * by + i) * P.w + (ox0 + 2 * bx + j);
            const float4 r = *reinterpret_cast<const float4*>(res_img + pix * P.res_ld + ch);
            ab += (v.x * r.x + v.y * r.y) + (v.z * r.z + v.w * r.w);
            aa += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
            bb += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
        }
        for (int o = 1; o < 64; o <<= 1) { ab += __shfl_xor(ab, o); aa += __shfl_xor(aa, o); bb += __shfl_xor(bb, o); }
        if (lane2 == 0) { red[(wave * 2 + it) * 3] = ab; red[(wave * 2 + it) * 3 + 1] = aa; red[(wave * 2 + it) * 3 + 2] = bb; }
      } else {
        const int ch = n0 + 4 * cq;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ch);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 z[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) z[x] = *reinterpret_cast<const float4*>(E + ((x * 2 + j) * 32 + bl) * W2_EPIX + 4 * cq);
          const int ox = ox0 + 2 * bx + j;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + 2 * by + i;
            float4 v;
            if (i == 0) v = make_float4((z[0].x + z[1].x) + z[2].x, (z[0].y + z[1].y) + z[2].y, (z[0].z + z[1].z) + z[2].z, (z[0].w + z[1].w) + z[2].w);
            else v = make_float4((z[1].x - z[2].x) - z[3].x, (z[1].y - z[2].y) - z[3].y, (z[1].z - z[2].z) - z[3].z, (z[1].w - z[2].w) - z[3].w);
            if (oy < P.h && ox < P.w) {
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
              if (P.gn_ws) {                                 // first pass of the GroupNorm that follows (R:model/blocks.py:5-6): sums of conv + bias
                gs[0] += v.x; gs[1] += v.y; gs[2] += v.z; gs[3] += v.w;
                gq[0] = fmaf(v.x, v.x, gq[0]); gq[1] = fmaf(v.y, v.y, gq[1]); gq[2] = fmaf(v.z, v.z, gq[2]); gq[3] = fmaf(v.w, v.w, gq[3]);
              }
              if (act == GPEMSR_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              else if (act == GPEMSR_ACT_LRELU) { v.x = fmaxf(v.x, 0.1f * v.x); v.y = fmaxf(v.y, 0.1f * v.y); v.z = fmaxf(v.z, 0.1f * v.z); v.w = fmaxf(v.w, 0.1f * v.w); }
              else if (act != GPEMSR_ACT_NONE) { v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act); }
              const long long pix = (long long)oy * P.w + ox;
              if (P.pixshuf) {                               // PixelShuffle(2): cout block q = ch / cq goes to sub-pixel (q >> 1, q & 1)
                const int q = ch / P.cq, c2 = ch - q * P.cq;
                *reinterpret_cast<float4*>(out_img + ((long long)(2 * oy + (q >> 1)) * (2 * P.w) + 2 * ox + (q & 1)) * P.out_ld + c2) = v;
                continue;
              }
              if (res_img) { const float4 rr = *reinterpret_cast<const float4*>(res_img + pix * P.res_ld + ch); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
              if (mul_img) { const float m = mul_img[pix]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
              *reinterpret_cast<float4*>(out_img + pix * P.out_ld + ch) = v;
            }
          }
        }
      }
      __syncthreads();                                   // the exchange buffer is rewritten by the next pass (or receives the next tile's chunk 2)
    }
    if (P.cos_ws) {
      if (tid2 < 12) {                                    // (strip, patch column, quantity); `red` was published by the barrier above
        const int k = tid2 % 3, pc = (tid2 / 3) & 1, st = tid2 / 6;
        float tot = 0.f;
        for (int wv = 0; wv < 8; ++wv)
          if (((wv >> 1) & 1) == pc) tot += red[(wv * 2 + st) * 3 + k];
        P.cos_ws[(((long long)cur.img * (P.tiles_y * 2) + cur.ty0 * 2 + st) * (P.tiles_x * 2) + cur.tx0 * 2 + pc) * 4 + k] = tot;
      }
      __syncthreads();                                   // `red` is rewritten by the next tile's epilogue
    } else if (P.gn_ws) {
      // per (tile, channel) sums over the tile's valid pixels: lanes with the same cout quad (lane2 & 15) by shuffles, the 8 waves through LDS
#pragma unroll
      for (int k = 0; k < 4; ++k)
        for (int o = 16; o < 64; o <<= 1) { gs[k] += __shfl_xor(gs[k], o); gq[k] += __shfl_xor(gq[k], o); }
      if (lane2 < 16) {
        float* r = red + (wave * 16 + lane2) * 8;
        *reinterpret_cast<float4*>(r) = make_float4(gs[0], gq[0], gs[1], gq[1]);
        *reinterpret_cast<float4*>(r + 4) = make_float4(gs[2], gq[2], gs[3], gq[3]);
      }
      __syncthreads();
      if (tid2 < 16) {
        float4 a = *reinterpret_cast<const float4*>(red + tid2 * 8), b = *reinterpret_cast<const float4*>(red + tid2 * 8 + 4);
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) {
          const float4 a2 = *reinterpret_cast<const float4*>(red + (wv * 16 + tid2) * 8), b2 = *reinterpret_cast<const float4*>(red + (wv * 16 + tid2) * 8 + 4);
          a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        const int part = cur.ty0 * P.tiles_x + cur.tx0;
        float* wsp = P.gn_ws + (((long long)cur.img * P.gn_parts + part) * P.cout + n0 + 4 * tid2) * 2;
        *reinterpret_cast<float4*>(wsp) = a;
        *reinterpret_cast<float4*>(wsp + 4) = b;
      }
      __syncthreads();
    }
    if (!has_next) break;

    // ---- the rest of the next tile's prologue: chunk 2 -> slot 2 (the exchange buffer is done), fragments of chunk 0 ----
    int infl2 = 0;
    if (nchunks > 2) { issue_chunk(); infl2 = n_issue; }
    wn_wait_vmcnt(infl2);                                // chunks 0 and 1 (issued two stages and an epilogue ago) have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      unsigned d_base, u_base;
      frag_bases(d_base, u_base);
      load_u(wsm, U_cur, u_base);
      load_raw(wsm, r_cur, d_base);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tile = next_tile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_wino2_persistent(const WinoParams& P, hipStream_t st) {
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2p_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d (Winograd form): cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
  const int cus = device_cus();          // (one process per GPU: dev_once above makes the same assumption)
  const int grid = P.nblocks < cus ? P.nblocks : cus;
  hipLaunchKernelGGL(conv_wino2p_f32_kernel, dim3(grid), dim3(512), WP_LDS, st, P);
  return check_launch("conv_wino2p_f32_kernel");
}

}  // namespace gpemsr
