// Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolution on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// The fp32 path (BASELINE.json configs[1], the official bench value) sits at 0.8 of the fp32 matrix peak on every large layer: nothing
// is left to tune towards, only arithmetic to remove.  F(2x2, 3x3) computes a 2x2 output block from a 4x4 input block with 16
// multiplies per (cin, cout) instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; correlation form, pad 1: d starts one pixel up / left of the block)
// i.e. sixteen independent GEMMs  M_p[tile][cout] = sum_cin V_p[tile][cin] U_p[cin][cout],  p = 4 xi + nu, on 1/4 of the pixels.
// Executed MFMA FLOPs = 16/36 of the algorithmic ones (both are reported: bench.py `executed_tflops`).  All arithmetic is fp32; the
// transforms add and subtract only (G's halves are folded into U on the host in float64), error ~1e-6 of the result (bar: 1e-3).
//
//   * workgroup = 8 waves, output tile 16 x 32 pixels = 128 Winograd blocks (4 MFMA row tiles of 32) x 32 couts;
//     wave w owns position row xi = w & 3 (its four positions nu = 0..3) for the two row tiles of half w >> 2: 8 accumulator tiles;
//   * NO transformed-input buffer: a lane builds its A operand V[xi][nu] (one block, 4 channels = 4 k-steps) in registers from eight
//     float4 of the raw halo image -- the row combination B^T[xi] of two image rows (one fma with a wave-uniform sign), then the four
//     column combinations: 32 vector operations per 16 MFMAs (the f32 MFMA runs at the vector rate, so this is ~6 % on top);
//   * raw halo image of a chunk of 8 channels in LDS as [channel quad][column parity][row][column / 2] 16-byte slots: the 16 blocks of
//     a row tile read consecutive slots (a plain [pixel][8 ch] image would be read at a 64-byte stride: 4-way bank conflicts).  LDS-DMA
//     writes lane-linear, so the layout is a permutation of the per-lane SOURCE address;
//   * U_p rows of the workgroup's 32 couts ride beside it ([pos][cout][8 ch], 16 KB per chunk); images in a 4-deep ring (fragments of the next chunk are read one stage early), one counted
//     vmcnt + barrier per chunk (32 MFMAs of a wave = 2k cycles of matrix time per SIMD between barriers);
//   * epilogue: column half of A^T M A in registers (same lane, same register index across the four nu tiles), the row half across the
//     four xi waves through LDS, then bias / activation / residual / per-pixel multiplier and float4 stores of whole 128-byte pixel rows.
//
// Replaces gpemsr_conv2d's direct form for the 3x3 stride-1 layers of R:model/GPEMSR.py:323-456 whose sources are multiples of 8
// channels and whose cout is a multiple of 32 (descriptor.transposed = 3; weight = packing.pack_winograd).
#include "conv_wino.h"

namespace gpemsr {

__global__ __launch_bounds__(512, 2) void conv_wino_f32_kernel(WinoParams P) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int xi = wave & 3, mh = wave >> 2;

  // ---- block -> (image, tile row, tile column, cout block); XCD-aware (consecutive logical blocks share an L2: the cout blocks of a tile) ----
  int bid = blockIdx.x;
  {
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = bid;
  const int tn = t % P.tiles_n; t /= P.tiles_n;
  const int tx0 = t % P.tiles_x; t /= P.tiles_x;
  const int ty0 = t % P.tiles_y; t /= P.tiles_y;
  const int img = t;
  const int oy0 = ty0 * 16, ox0 = tx0 * 32, n0 = tn * 32;

  // ---- DMA slots (constant per thread): raw image slot -> pixel inside the source image (or -1), channel quad; U slot -> float offset ----
  int a_pix[WN_NA], a_q[WN_NA];
#pragma unroll
  for (int i = 0; i < WN_NA; ++i) {
    const int s = tid + i * 512;
    a_pix[i] = -1; a_q[i] = 0;
    if (s < WN_ASLOTS) {
      const int hx2 = s % WN_HW2, r1 = s / WN_HW2;
      const int hy = r1 % WN_HH, r2 = r1 / WN_HH;
      const int par = r2 & 1, qd = r2 >> 1;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + 2 * hx2 + par;
      a_q[i] = qd;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pix[i] = iy * P.w + ix;
    }
  }
  int b_off[WN_NB];
#pragma unroll
  for (int i = 0; i < WN_NB; ++i) {
    const int s = tid + i * 512;                       // [pos][cout][quad]
    const int qd = s & 1, co = (s >> 1) & 31, pos = s >> 6;
    b_off[i] = (pos * P.cout + n0 + co) * 8 + 4 * qd;            // inside one chunk's [position][cout][8] block of U
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wsm;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  // out-of-image halo slots: zero once in every ring slot, never written again (their lanes stay masked in every DMA)
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < WN_NA; ++i)
      if (tid + i * 512 < WN_ASLOTS && a_pix[i] < 0)
        for (int rs = 0; rs < WN_RING; ++rs) *reinterpret_cast<float4*>(wsm + rs * WN_STAGE + (tid + i * 512) * 16) = z;
  }
  int na_w = 0;
#pragma unroll
  for (int i = 0; i < WN_NA; ++i) na_w += (__ballot(a_pix[i] >= 0) != 0ull) ? 1 : 0;
  const int n_issue = na_w + WN_NB;                    // DMA instructions this wave issues per chunk

  // ---- chunk cursor over the virtual concat of the sources ----
  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / 8;
  int f_src = 0, f_c0 = 0, f_cpad = 0, f_chunk = 0;    // next chunk to issue
  auto issue_chunk = [&]() {
    const unsigned la = lds0 + (unsigned)((f_chunk % WN_RING) * WN_STAGE);
    const float* sp = P.src[f_src] + (long long)img * P.img_stride[f_src] + f_c0;
    const unsigned pixb = (unsigned)P.ld[f_src] * 4u;
#pragma unroll
    for (int i = 0; i < WN_NA; ++i)
      if (a_pix[i] >= 0) wn_glds16((unsigned)a_pix[i] * pixb + 16u * (unsigned)a_q[i], sp, la + i * 8192u);
    const float* wp = P.weight + (long long)(f_cpad + f_c0) * (16 * P.cout);      // chunk (f_cpad + f_c0) / 8 of U[cin / 8][16][cout][8]
#pragma unroll
    for (int i = 0; i < WN_NB; ++i) wn_glds16((unsigned)b_off[i] * 4u, wp, la + (unsigned)WN_ABYTES + i * 8192u);
    ++f_chunk; f_c0 += 8;
    if (f_c0 >= P.c[f_src] && f_src + 1 < P.nsrc) { f_cpad += P.c[f_src]; f_c0 = 0; ++f_src; }
  };

  // ---- fragment addressing (tile-invariant) ----
  // row combination of position row xi: V-row = d[aA] + sB * d[aB]
  const int aA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int aB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sB = xi == 1 ? 1.f : -1.f;
  // lane's block inside row tile mtl (= 2 mh + mtl): ty = 2 (2 mh + mtl) + (li >> 4), tx = li & 15; d[a][b] = slot ((lh * 2 + (b & 1)) * 18 + 2 ty + a) * 17 + tx + (b >> 1)
  unsigned d_base[2];
#pragma unroll
  for (int mtl = 0; mtl < 2; ++mtl) {
    const int ty = 2 * (2 * mh + mtl) + (li >> 4), tx = li & 15;
    d_base[mtl] = (unsigned)((((lh * 2) * WN_HH + 2 * ty) * WN_HW2 + tx) * 16);
  }
  const unsigned u_base = (unsigned)(WN_ABYTES + ((4 * xi * 32 + li) * 2 + lh) * 16);

  f32x16 acc[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][mtl][r] = 0.f;

  // ---- main loop, software-pipelined across the stage barrier ----
  // Fragments of a chunk are needed 32 MFMAs apart only; with both waves of a SIMD leaving the barrier together, a stage that starts with
  // its own 20 LDS reads + transform leaves the matrix pipe idle for their latency (first version: 58 % of the fp32 peak on the executed
  // work).  So the raw rows of row tile 0 and the four U fragments of chunk c + 1 are read during stage c, behind the MFMAs of row tile 1:
  // a stage opens with 16 vector operations and its first MFMA.  Ring of four images: chunk c + 3 is issued at the top of stage c into the
  // slot chunk c - 1 left at the last barrier; the wait at the end of stage c leaves only that chunk in flight (c + 1, c + 2 have landed).
  auto load_raw = [&](const char* st, int mtl, float4 (&r4)[4]) {          // row combination B^T[xi] of the two image rows, four columns
    const char* dp = st + d_base[mtl];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int cofs = ((b & 1) * WN_HH * WN_HW2 + (b >> 1)) * 16;
      const float4 va = *reinterpret_cast<const float4*>(dp + cofs + aA * (WN_HW2 * 16));
      const float4 vb = *reinterpret_cast<const float4*>(dp + cofs + aB * (WN_HW2 * 16));
      r4[b] = make_float4(fmaf(sB, vb.x, va.x), fmaf(sB, vb.y, va.y), fmaf(sB, vb.z, va.z), fmaf(sB, vb.w, va.w));
    }
  };
  auto load_u = [&](const char* st, float4 (&U)[4]) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) U[nu] = *reinterpret_cast<const float4*>(st + u_base + nu * (32 * 2 * 16));
  };
  auto mma = [&](const float4 (&r4)[4], const float4 (&U)[4], int mtl) {
    float4 V[4];                                         // column combinations
    V[0] = make_float4(r4[0].x - r4[2].x, r4[0].y - r4[2].y, r4[0].z - r4[2].z, r4[0].w - r4[2].w);
    V[1] = make_float4(r4[1].x + r4[2].x, r4[1].y + r4[2].y, r4[1].z + r4[2].z, r4[1].w + r4[2].w);
    V[2] = make_float4(r4[2].x - r4[1].x, r4[2].y - r4[1].y, r4[2].z - r4[1].z, r4[2].w - r4[1].w);
    V[3] = make_float4(r4[1].x - r4[3].x, r4[1].y - r4[3].y, r4[1].z - r4[3].z, r4[1].w - r4[3].w);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      acc[nu][mtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].x, U[nu].x, acc[nu][mtl], 0, 0, 0);
      acc[nu][mtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].y, U[nu].y, acc[nu][mtl], 0, 0, 0);
      acc[nu][mtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].z, U[nu].z, acc[nu][mtl], 0, 0, 0);
      acc[nu][mtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].w, U[nu].w, acc[nu][mtl], 0, 0, 0);
    }
  };
  issue_chunk();
  if (nchunks > 1) issue_chunk();
  int infl = 0;
  if (nchunks > 2) { issue_chunk(); infl = n_issue; }
  wn_wait_vmcnt(infl);                                   // chunks 0 and 1 have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float4 U_cur[4], r0_cur[4];
  load_u(wsm, U_cur);
  load_raw(wsm, 0, r0_cur);

  for (int ck = 0; ck < nchunks; ++ck) {
    int issued = 0;
    if (ck + 3 < nchunks) { issue_chunk(); issued = n_issue; }        // -> slot (ck + 3) % 4, left by chunk ck - 1 at the last barrier
    const char* st = wsm + (ck % WN_RING) * WN_STAGE;
    float4 r1[4];
    load_raw(st, 1, r1);
    mma(r0_cur, U_cur, 0);
    float4 U_nxt[4], r0_nxt[4];
    if (ck + 1 < nchunks) {                              // (wave-uniform) next chunk's first fragments: their latency lies under the MFMAs below
      const char* sn = wsm + ((ck + 1) % WN_RING) * WN_STAGE;
      load_u(sn, U_nxt);
      load_raw(sn, 0, r0_nxt);
    }
    mma(r1, U_cur, 1);
    // everything issued BEFORE this stage has landed (this wave's part); the barrier publishes all parts and frees chunk ck's slot
    wn_wait_vmcnt(issued);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) { U_cur[q] = U_nxt[q]; r0_cur[q] = r0_nxt[q]; }
  }

  // ---- epilogue ----
  // column half of A^T M A in registers: Z[0] = M0 + M1 + M2, Z[1] = M1 - M2 - M3 (per lane and register index); the row half (over
  // the four xi waves) through LDS: E[xi][j][block 0..127][cout 0..31 (+4)]
  float* E = reinterpret_cast<float*>(wsm);
#pragma unroll
  for (int mtl = 0; mtl < 2; ++mtl) {
    const int mt = 2 * mh + mtl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float z0 = (acc[0][mtl][r] + acc[1][mtl][r]) + acc[2][mtl][r];
      const float z1 = (acc[1][mtl][r] - acc[2][mtl][r]) - acc[3][mtl][r];
      E[((xi * 2 + 0) * 128 + mt * 32 + row) * WN_EPIX + li] = z0;
      E[((xi * 2 + 1) * 128 + mt * 32 + row) * WN_EPIX + li] = z1;
    }
  }
  __syncthreads();
  const long long img_pix0 = (long long)img * P.h * P.w;
  const float* res_img = P.residual ? P.residual + img_pix0 * P.res_ld : nullptr;
  const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
  float* out_img = P.out + img_pix0 * P.out_ld;
  const int act = P.act;
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * 512;                   // (block, cout quad): 8 consecutive threads = the 128 bytes of one pixel
    const int blk = item >> 3, cq = item & 7;
    const int ty = blk >> 4, tx = blk & 15;
    const int ch = n0 + 4 * cq;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ch);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 z[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) z[x] = *reinterpret_cast<const float4*>(E + ((x * 2 + j) * 128 + blk) * WN_EPIX + 4 * cq);
      const int ox = ox0 + 2 * tx + j;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * ty + i;
        float4 v;
        if (i == 0) v = make_float4((z[0].x + z[1].x) + z[2].x, (z[0].y + z[1].y) + z[2].y, (z[0].z + z[1].z) + z[2].z, (z[0].w + z[1].w) + z[2].w);
        else v = make_float4((z[1].x - z[2].x) - z[3].x, (z[1].y - z[2].y) - z[3].y, (z[1].z - z[2].z) - z[3].z, (z[1].w - z[2].w) - z[3].w);
        if (oy < P.h && ox < P.w) {
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          if (act == GPEMSR_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (act == GPEMSR_ACT_LRELU) { v.x = fmaxf(v.x, 0.1f * v.x); v.y = fmaxf(v.y, 0.1f * v.y); v.z = fmaxf(v.z, 0.1f * v.z); v.w = fmaxf(v.w, 0.1f * v.w); }
          else if (act != GPEMSR_ACT_NONE) { v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act); }
          const long long pix = (long long)oy * P.w + ox;
          if (res_img) { const float4 rr = *reinterpret_cast<const float4*>(res_img + pix * P.res_ld + ch); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
          if (mul_img) { const float m = mul_img[pix]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
          *reinterpret_cast<float4*>(out_img + pix * P.out_ld + ch) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same form with 64 couts per workgroup (cout % 64 == 0: every large layer of the network).  The first kernel's lane builds one
// V fragment (32 vector operations per row tile and chunk) for 16 MFMAs; the f32 MFMA shares the vector ALUs, so those two vector
// operations per MFMA cost ~30 % (measured: 0.53 of the fp32 peak on the executed work).  Here a wave owns ONE row tile x TWO cout
// tiles: the same V feeds 32 MFMAs (one vector operation per MFMA).  Workgroup = 8 x 32 output pixels (64 blocks, two row tiles) x 64
// couts; raw image [2 quads][2 parities][10 rows][17] = 10.9 KB, U image [16 positions][64 couts][8 ch] = 32 KB per chunk, ring of
// three; ALL fragments of chunk c + 1 (the row-combined image rows and the eight U fragments) are read during stage c, so chunk c's
// slot is free at the top of stage c and receives chunk c + 3.
// ---------------------------------------------------------------------------------------------------------------------------------

// ASYM: only waves 0-3 (one per SIMD) issue the LDS-DMA of a stage, twice as many instructions each; waves 4-7 go straight to their
// MFMAs.  The DMA intake of a CU is ~27 B/clk whoever issues (43 KB per stage = ~1,600 of the stage's 4,096 matrix clocks) and an issuing wave
// sits in the queue for that long: with all eight waves issuing, both waves of every SIMD sat there together and the matrix pipe idled;
// with one issuer per SIMD the other wave's 32 MFMAs cover the queue time.
template <bool ASYM>
__global__ __launch_bounds__(512, 2) void conv_wino2_f32_kernel(WinoParams P) {
  constexpr int NTI = ASYM ? 256 : 512;                        // issuing threads
  constexpr int W2_NA = (W2_ASLOTS + NTI - 1) / NTI;           // 3 : 2
  constexpr int W2_NB = W2_BSLOTS / NTI;                       // 8 : 4
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int xi = wave & 3, mh = wave >> 2;
  const bool issuer = !ASYM || wave < 4;                       // (wave-uniform)

  int bid = blockIdx.x;
  {
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  // pixel tile fastest, cout block slowest: the workgroups that run together on an XCD read the SAME U slice (16 positions x 64 couts x cin:
  // 2 MB at 512 channels, it stays in the XCD's 4 MB L2) and different image tiles.  With the cout block fastest every workgroup of an XCD
  // streamed a different slice -- the whole 16.8 MB U tensor per pixel tile, 265 GB of fabric reads per step (profiles/r04a_fp32_pmc_summary.json)
  // ... and `tn_group` (<= 4) cout blocks of one pixel tile run side by side: they share the tile's input chunks in L2, so the image is
  // streamed tiles_n / tn_group times instead of tiles_n times (512 channels: 8 -> 2; the U chunks of the group, 4 x 32 KB, are shared by the
  // ~8 pixel tiles in flight on the XCD).  Counters: profiles/r04_fp32_pmc_summary.json (289 GB read by this kernel with tn_group = 1)
  int t = bid;
  const int tn_lo = t % P.tn_group; t /= P.tn_group;
  const int tx0 = t % P.tiles_x; t /= P.tiles_x;
  const int ty0 = t % P.tiles_y; t /= P.tiles_y;
  const int img = t % P.n; t /= P.n;
  const int tn = t * P.tn_group + tn_lo;
  const int oy0 = ty0 * 8, ox0 = tx0 * 32, n0 = tn * 64;

  int a_pk[W2_NA];                                           // 2 * pixel + channel quad of the slot, -1: outside the image / not an issuer
#pragma unroll
  for (int i = 0; i < W2_NA; ++i) {
    const int s = tid + i * NTI;
    a_pk[i] = -1;
    if (issuer && s < W2_ASLOTS) {
      const int hx2 = s % WN_HW2, r1 = s / WN_HW2;
      const int hy = r1 % W2_HH, r2 = r1 / W2_HH;
      const int par = r2 & 1, qd = r2 >> 1;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + 2 * hx2 + par;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pk[i] = 2 * (iy * P.w + ix) + qd;
    }
  }
  // U slot s = tid + i * NTI of [pos][cout 64][quad]: pos = (tid >> 7) + i * NTI / 128
  const unsigned b_off0 = (unsigned)((((tid & (NTI - 1)) >> 7) * P.cout + n0 + ((tid >> 1) & 63)) * 8 + 4 * (tid & 1)) * 4u;
  const unsigned b_step = (unsigned)((NTI / 128) * P.cout * 8) * 4u;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wsm;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < W2_NA; ++i)
      if (issuer && tid + i * NTI < W2_ASLOTS && a_pk[i] < 0)
        for (int rs = 0; rs < W2_RING; ++rs) *reinterpret_cast<float4*>(wsm + rs * W2_STAGE + (tid + i * NTI) * 16) = z;
  }
  int na_w = 0;
#pragma unroll
  for (int i = 0; i < W2_NA; ++i) na_w += (__ballot(a_pk[i] >= 0) != 0ull) ? 1 : 0;
  const int n_issue = issuer ? na_w + W2_NB : 0;

  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / 8;
  int f_src = 0, f_c0 = 0, f_cpad = 0, f_chunk = 0, f_slot = 0;
  auto issue_chunk = [&]() {
    const unsigned la = lds0 + (unsigned)(f_slot * W2_STAGE);
    const float* sp = P.src[f_src] + (long long)img * P.img_stride[f_src] + f_c0;
    const unsigned pixb = (unsigned)P.ld[f_src] * 4u;
    if (issuer) {
#pragma unroll
      for (int i = 0; i < W2_NA; ++i)
        if (a_pk[i] >= 0) wn_glds16((unsigned)(a_pk[i] >> 1) * pixb + 16u * (unsigned)(a_pk[i] & 1), sp, la + i * (NTI * 16u));
      const float* wp = P.weight + (long long)(f_cpad + f_c0) * (16 * P.cout);      // chunk (f_cpad + f_c0) / 8 of U[cin / 8][16][cout][8]
      unsigned bo = b_off0;
      asm volatile("" : "+v"(bo));                       // opaque: eight hoisted offsets would not fit the register file
#pragma unroll
      for (int i = 0; i < W2_NB; ++i) { wn_glds16(bo, wp, la + (unsigned)W2_ABYTES + i * (NTI * 16u)); bo += b_step; }
    }
    ++f_chunk; f_c0 += 8;
    f_slot = f_slot == W2_RING - 1 ? 0 : f_slot + 1;
    if (f_c0 >= P.c[f_src] && f_src + 1 < P.nsrc) { f_cpad += P.c[f_src]; f_c0 = 0; ++f_src; }
  };

  const int aA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int aB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sB = xi == 1 ? 1.f : -1.f;
  const int ty = 2 * mh + (li >> 4), tx = li & 15;
  const unsigned d_base = (unsigned)((((lh * 2) * W2_HH + 2 * ty) * WN_HW2 + tx) * 16);
  const unsigned u_base = (unsigned)(W2_ABYTES + ((4 * xi * 64 + li) * 2 + lh) * 16);

  f32x16 acc[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;

  auto load_raw = [&](const char* st, float4 (&r4)[4]) {
    const char* dp = st + d_base;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int cofs = ((b & 1) * W2_HH * WN_HW2 + (b >> 1)) * 16;
      const float4 va = *reinterpret_cast<const float4*>(dp + cofs + aA * (WN_HW2 * 16));
      const float4 vb = *reinterpret_cast<const float4*>(dp + cofs + aB * (WN_HW2 * 16));
      r4[b] = make_float4(fmaf(sB, vb.x, va.x), fmaf(sB, vb.y, va.y), fmaf(sB, vb.z, va.z), fmaf(sB, vb.w, va.w));
    }
  };
  auto load_u = [&](const char* st, float4 (&U)[4][2]) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) U[nu][nt] = *reinterpret_cast<const float4*>(st + u_base + nu * (64 * 2 * 16) + nt * (32 * 2 * 16));
  };

  issue_chunk();
  if (nchunks > 1) issue_chunk();
  int infl = 0;
  if (nchunks > 2) { issue_chunk(); infl = n_issue; }
  wn_wait_vmcnt(infl);                                   // chunks 0 and 1 have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float4 U_cur[4][2], r_cur[4];
  load_u(wsm, U_cur);
  load_raw(wsm, r_cur);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave holds chunk 0's fragments: its slot may be refilled
  asm volatile("" ::: "memory");

  int slot_n = 1;                                        // ring slot of chunk ck + 1
  // one stage; (Uc, rc) = this chunk's fragments, (Un, rn) receive the next chunk's.  The loop below alternates two register sets
  // (a copy "cur = next" per stage was 48 vector moves per wave on the pipe the f32 MFMA shares)
  auto stage = [&](const int ck, float4 (&Uc)[4][2], float4 (&rc)[4], float4 (&Un)[4][2], float4 (&rn)[4]) {
    int issued = 0;
    if (ck + 3 < nchunks && !(WINO_DBG(P) & 4)) { issue_chunk(); issued = n_issue; }        // -> the slot of chunk ck (its fragments are in registers)
    const char* sn = wsm + slot_n * W2_STAGE;
    const bool more = ck + 1 < nchunks && !(WINO_DBG(P) & 1);  // (wave-uniform)
    if (more) {                                          // next chunk's image rows and the U fragments of cout tile 0: under the first 16 MFMAs
      load_raw(sn, rn);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) Un[nu][0] = *reinterpret_cast<const float4*>(sn + u_base + nu * (64 * 2 * 16));
    }
    float4 V[4];
    V[0] = make_float4(rc[0].x - rc[2].x, rc[0].y - rc[2].y, rc[0].z - rc[2].z, rc[0].w - rc[2].w);
    V[1] = make_float4(rc[1].x + rc[2].x, rc[1].y + rc[2].y, rc[1].z + rc[2].z, rc[1].w + rc[2].w);
    V[2] = make_float4(rc[2].x - rc[1].x, rc[2].y - rc[1].y, rc[2].z - rc[1].z, rc[2].w - rc[1].w);
    V[3] = make_float4(rc[1].x - rc[3].x, rc[1].y - rc[3].y, rc[1].z - rc[3].z, rc[1].w - rc[3].w);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {                     // cout tile 0 first: its U registers are free for the second half of the prefetch
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].x, Uc[nu][nt].x, acc[nu][nt], 0, 0, 0);
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].y, Uc[nu][nt].y, acc[nu][nt], 0, 0, 0);
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].z, Uc[nu][nt].z, acc[nu][nt], 0, 0, 0);
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[nu].w, Uc[nu][nt].w, acc[nu][nt], 0, 0, 0);
      }
      if (nt == 0 && more) {
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) Un[nu][1] = *reinterpret_cast<const float4*>(sn + u_base + nu * (64 * 2 * 16) + 32 * 2 * 16);
      }
    }
    // chunk ck + 2 must have landed before the next stage reads it: only this stage's issues may stay in flight
    wn_wait_vmcnt(issued);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(WINO_DBG(P) & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot_n = slot_n == W2_RING - 1 ? 0 : slot_n + 1;
  };
  float4 U_alt[4][2], r_alt[4];
  for (int ck = 0; ck < nchunks; ck += 2) {
    stage(ck, U_cur, r_cur, U_alt, r_alt);
    if (ck + 1 < nchunks) stage(ck + 1, U_alt, r_alt, U_cur, r_cur);
  }

  float* E = reinterpret_cast<float*>(wsm);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float z0 = (acc[0][nt][r] + acc[1][nt][r]) + acc[2][nt][r];
      const float z1 = (acc[1][nt][r] - acc[2][nt][r]) - acc[3][nt][r];
      E[((xi * 2 + 0) * 64 + mh * 32 + row) * W2_EPIX + nt * 32 + li] = z0;
      E[((xi * 2 + 1) * 64 + mh * 32 + row) * W2_EPIX + nt * 32 + li] = z1;
    }
  __syncthreads();
  const long long img_pix0 = (long long)img * P.h * P.w;
  const float* res_img = P.residual ? P.residual + img_pix0 * P.res_ld : nullptr;
  const float* mul_img = P.pixmul ? P.pixmul + img_pix0 : nullptr;
  float* out_img = P.out + img_pix0 * (P.pixshuf ? 4 : 1) * P.out_ld;
  const int act = P.act;
  if (P.cos_ws) {
    // R:model/GPEMSR.py:387-395 without the second relu1_2 map in memory (the direct kernel's XEPI = 2 epilogue, same record layout
    // [n][h/4 strips][w/16 patch columns][4] for gpemsr_patch_cosine_finish): b = act(this convolution + bias), a = the `residual` operand;
    // sums of a.b, a.a, b.b over the 4-row strips and 16-pixel patch columns of this 8 x 32 tile (64 couts = all channels).  Item `it` of a
    // thread lies in strip `it`, a wave's items in patch column (wave >> 1) & 1; fixed reduction order -> bit-stable.
    float* red = reinterpret_cast<float*>(wsm + W2_EBYTES);        // [8 waves][2 strips][3]
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      const int item = tid + it * 512;
      const int blk = item >> 4, cq = item & 15;
      const int by = blk >> 4, bx = blk & 15;
      const int ch = 4 * cq;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ch);
      float ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 z[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) z[x] = *reinterpret_cast<const float4*>(E + ((x * 2 + j) * 64 + blk) * W2_EPIX + 4 * cq);
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
      if (lane == 0) { red[(wave * 2 + it) * 3] = ab; red[(wave * 2 + it) * 3 + 1] = aa; red[(wave * 2 + it) * 3 + 2] = bb; }
    }
    __syncthreads();
    if (tid < 12) {                                    // (strip, patch column, quantity)
      const int k = tid % 3, pc = (tid / 3) & 1, st = tid / 6;
      float tot = 0.f;
      for (int wv = 0; wv < 8; ++wv)
        if (((wv >> 1) & 1) == pc) tot += red[(wv * 2 + st) * 3 + k];
      P.cos_ws[(((long long)img * (P.tiles_y * 2) + ty0 * 2 + st) * (P.tiles_x * 2) + tx0 * 2 + pc) * 4 + k] = tot;
    }
    return;
  }
  float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};      // GroupNorm partial sums of this thread's 4 channels (both items share the cout quad)
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * 512;                   // (block, cout quad): 16 consecutive threads = the 256 bytes of one pixel
    const int blk = item >> 4, cq = item & 15;
    const int by = blk >> 4, bx = blk & 15;
    const int ch = n0 + 4 * cq;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ch);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 z[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) z[x] = *reinterpret_cast<const float4*>(E + ((x * 2 + j) * 64 + blk) * W2_EPIX + 4 * cq);
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
  if (P.gn_ws) {
    // per (tile, channel) sums over the tile's valid pixels: lanes with the same cout quad (lane & 15) by shuffles, the 8 waves through LDS
    // behind the exchange buffer, fixed order (bit-stable); one record per 8 x 32 tile: gn_parts = tiles per image
    float* red = reinterpret_cast<float*>(wsm + W2_EBYTES);        // [8 waves][16 quads][8]
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int o = 16; o < 64; o <<= 1) { gs[k] += __shfl_xor(gs[k], o); gq[k] += __shfl_xor(gq[k], o); }
    __syncthreads();                                               // (the patch-cosine path shares this corner of LDS; E reads are done)
    if (lane < 16) {
      float* r = red + (wave * 16 + lane) * 8;
      *reinterpret_cast<float4*>(r) = make_float4(gs[0], gq[0], gs[1], gq[1]);
      *reinterpret_cast<float4*>(r + 4) = make_float4(gs[2], gq[2], gs[3], gq[3]);
    }
    __syncthreads();
    if (tid < 16) {
      float4 a = *reinterpret_cast<const float4*>(red + tid * 8), b = *reinterpret_cast<const float4*>(red + tid * 8 + 4);
#pragma unroll
      for (int wv = 1; wv < 8; ++wv) {
        const float4 a2 = *reinterpret_cast<const float4*>(red + (wv * 16 + tid) * 8), b2 = *reinterpret_cast<const float4*>(red + (wv * 16 + tid) * 8 + 4);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      const int part = ty0 * P.tiles_x + tx0;
      float* wsp = P.gn_ws + (((long long)img * P.gn_parts + part) * P.cout + n0 + 4 * tid) * 2;
      *reinterpret_cast<float4*>(wsp) = a;
      *reinterpret_cast<float4*>(wsp + 4) = b;
    }
  }
}

// descriptor.transposed == 3: called from gpemsr_conv2d (conv_mfma.hip)
int conv2d_winograd(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap, int* parts_only) {
  GP_REQUIRE(d->ksize == 3 && d->stride == 1 && d->weight_image_stride == 0, "conv2d (Winograd form): 3x3, stride 1, one weight set");
  if (d->gn_partials || parts_only) {
    GP_REQUIRE(d->cout % 64 == 0 && d->act == GPEMSR_ACT_NONE && !d->residual && !d->pixmul && !d->pixel_shuffle && !d->cos_partials,
               "conv2d (Winograd form): GroupNorm partial sums need cout %% 64 == 0, the plain store, no activation / residual / multiplier");
    if (parts_only) { *parts_only = cdiv(d->h, 8) * cdiv(d->w, 32); return GPEMSR_OK; }
    GP_REQUIRE((reinterpret_cast<uintptr_t>(d->gn_partials) & 15) == 0, "conv2d (Winograd form): gn_partials must be 16-byte aligned");
  }
  GP_REQUIRE(!d->cos_partials || (d->cout == 64 && d->h % 16 == 0 && d->w % 32 == 0 && d->residual && !d->pixel_shuffle && !d->pixmul &&
                                  (reinterpret_cast<uintptr_t>(d->cos_partials) & 15) == 0),
             "conv2d (Winograd form): patch-cosine sums need 64 output channels, the operand in `residual`, height %% 16 == 0, width %% 32 == 0");
  GP_REQUIRE(!d->pixel_shuffle || (d->cout % 64 == 0 && (d->cout / 4) % 4 == 0 && !d->residual && !d->pixmul),
             "conv2d (Winograd form): pixel_shuffle needs cout %% 64 == 0 and no residual / multiplier");
  GP_REQUIRE(d->cout % 32 == 0, "conv2d (Winograd form): cout %% 32 == 0 (got %d)", d->cout);
  WinoParams P{};
  int cin = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].c % 8 == 0 && d->src[s].ld % 4 == 0 && d->src[s].ld >= d->src[s].c &&
               (reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0, "conv2d (Winograd form): source %d needs c %% 8 == 0, 16-byte aligned rows", s);
    P.src[s] = d->src[s].ptr; P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    GP_REQUIRE(P.img_stride[s] % 4 == 0 && (long long)d->h * d->w * P.ld[s] * 4 < (1ll << 32), "conv2d (Winograd form): source %d too large / misaligned", s);
    cin += d->src[s].c;
  }
  GP_REQUIRE((long long)16 * d->cout * cin * 4 < (1ll << 32), "conv2d (Winograd form): weight tensor too large for 32-bit offsets");
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0 && d->out_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 15) == 0 &&
             (!d->bias || (reinterpret_cast<uintptr_t>(d->bias) & 15) == 0) &&
             (!d->residual || (d->res_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->residual) & 15) == 0)), "conv2d (Winograd form): 16-byte alignment of weight / out / bias / residual");
  static int force1 = -1;
  if (force1 < 0) { const char* e = getenv("GPEMSR_WINO_FORM"); force1 = (e && e[0] == '1') ? 1 : 0; }      // A/B: the 32-cout kernel everywhere
  const bool wide = d->cout % 64 == 0 && (!force1 || d->pixel_shuffle || d->cos_partials || d->gn_partials);      // 64 couts per workgroup: one vector operation per MFMA instead of two
  // 64-cout form, 64 input channels (8 chunks; the kernel takes any even chunk count = 2 mod 3): the persistent kernel overlaps the next tile's
  // first chunks with the last two stages and the epilogue of the current one (conv_wino_p.hip).  Measured per layer on one box
  // (profiles/r05_winograd_persistent_layers.log): 64-channel layers +1-2 % (VGG relu1_2 76.8 -> 75.5 ms), 256-channel layers (32 chunks)
  // 2 % SLOWER (its position-by-position MFMA order), so only the 8-chunk layers take it.  GPEMSR_WINO_PERSIST=0: one tile per workgroup;
  // =2: every eligible chunk count.
  const char* pe = getenv("GPEMSR_WINO_PERSIST");             // (read per launch: A/B runs switch it inside one process)
  const int nch = cin / 8;
  const int pmode = pe ? atoi(pe) : 1;
  const bool persist = wide && pmode != 0 && nch >= 8 && nch % 3 == 2 && nch % 2 == 0 && (nch == 8 || pmode == 2);
  if (name_buf) { snprintf(name_buf, (size_t)name_cap, wide ? (persist ? "conv_wino2p_f32_kernel" : "conv_wino2_f32_kernel") : "conv_wino_f32_kernel"); return GPEMSR_OK; }
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w; P.cin_pad = cin; P.cout = d->cout;
  P.weight = d->weight; P.bias = d->bias; P.act = d->act; P.residual = d->residual; P.res_ld = d->res_ld; P.pixmul = d->pixmul;
  P.out = d->out; P.out_ld = d->out_ld;
  P.pixshuf = d->pixel_shuffle; P.cq = d->cout / 4;
  P.cos_ws = d->cos_partials;
  P.gn_ws = d->gn_partials; P.gn_parts = cdiv(d->h, 8) * cdiv(d->w, 32);
#ifdef GPEMSR_WINO_PROBE
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GPEMSR_WINO_DBG"); dbg = e ? atoi(e) : 0; } P.dbg = dbg; }
#else
  P.dbg = 0;                       // the shipped library has no way to skip reads / barriers / DMA (each makes the results wrong on purpose)
#endif
  P.tiles_x = cdiv(d->w, 32); P.tiles_y = cdiv(d->h, wide ? 8 : 16); P.tiles_n = d->cout / (wide ? 64 : 32);
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d (Winograd form): grid too large");
  P.nblocks = (int)nb;
  {
    const char* e = getenv("GPEMSR_WINO_TN_GROUP");          // (read per launch: the tests switch it inside one process)
    int grp = e ? atoi(e) : 4;
    if (grp < 1) grp = 1;
    P.tn_group = 1;
    for (int a = 4; a >= 2; a >>= 1) if (a <= grp && P.tiles_n % a == 0) { P.tn_group = a; break; }
  }
  P.mg_g = 0xFFFFFFFFu / (unsigned)P.tn_group; P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y;
  P.mg_n = 0xFFFFFFFFu / (unsigned)d->n;
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2_f32_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2_f32_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d (Winograd form): cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
  if (persist) return launch_wino2_persistent(P, reinterpret_cast<hipStream_t>(stream));
  if (wide) {
    const size_t lds2 = (size_t)(W2_EBYTES > W2_RING * W2_STAGE ? W2_EBYTES : W2_RING * W2_STAGE) + 4096;     // + the partial sums' cross-wave exchange
    const char* es = getenv("GPEMSR_WINO_SYM");
    const bool sym = es && atoi(es) != 0;                    // A/B: all eight waves issue the DMA (the first form)
    if (sym) hipLaunchKernelGGL(conv_wino2_f32_kernel<false>, dim3(P.nblocks), dim3(512), lds2, reinterpret_cast<hipStream_t>(stream), P);
    else hipLaunchKernelGGL(conv_wino2_f32_kernel<true>, dim3(P.nblocks), dim3(512), lds2, reinterpret_cast<hipStream_t>(stream), P);
    return check_launch("conv_wino2_f32_kernel");
  }
  const size_t lds = (size_t)(WN_EBYTES > WN_RING * WN_STAGE ? WN_EBYTES : WN_RING * WN_STAGE);
  hipLaunchKernelGGL(conv_wino_f32_kernel, dim3(P.nblocks), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("conv_wino_f32_kernel");
}

}  // namespace gpemsr
