// 1x1 convolution / Linear / batched matrix product of the bf16 data path with the ACTIVATIONS READ STRAIGHT INTO REGISTERS.
//
//   D[pixel][cout] = sum_c In[pixel][c] * W[cout][c]          v_mfma_f32_32x32x16_bf16, D^T accumulators (weights = row operand)
//
// The ring kernel of conv_bf16.hip stages BOTH operands through LDS: a 256 pixel x 128 cout tile with 64-deep stages moves 48 KB by LDS-DMA
// per 16 MFMAs of a wave = 47 B per matrix clock against the ~27 B/clk a CU's global -> LDS path takes in, so the 1x1 layers of the VQGAN
// prior (q / k / v / proj_out of the NonLocalBlocks, R:model/blocks.py:61-83; the 1x1 layers of indexer and decoder, R:model/indexer.py:89-96,
// R:model/decoder.py:16; nn.Linear 512 -> 1024, R:model/indexer.py:100) sat at 0.5 PFLOP/s where hipBLASLt reaches 0.73-1.27 on the same
// shapes (scripts/gemm_yardstick.py).  With no filter taps a pixel's fragment is used by ONE wave only -- LDS buys nothing for it.  Here:
//   * a workgroup = 8 waves = 256 pixels x 256 couts; wave w owns pixels 32 w .. 32 w + 31 and ALL 256 couts (8 accumulator tiles, 128
//     registers); its A fragments (8 consecutive channels of its pixel: one 16-byte global load, the four k-steps of a 64-channel chunk use
//     the pixel's whole 128-byte line) are requested one chunk ahead into a second register set;
//   * only the weights go through LDS: a 4-deep ring of 32 KB stage images [piece][256 couts][8] (the staged order the host packs,
//     packing.pack_conv_bf16), filled by LDS-DMA -- 16 B per matrix clock.  There are no loader waves (128 accumulators + two fragment sets
//     need the 256 registers of 8 waves per CU): every wave issues its four DMA instructions per stage itself;
//   * one barrier per chunk (32 MFMAs of a wave); plain loads and DMA share the in-order vmcnt and only the loads are visible to the
//     compiler: the order k-step 0 / refill DMA / next fragments' loads / k-steps 1-3 makes every wait the compiler inserts sufficient and
//     never early (see `chunk`);
//   * persistent workgroups walk tiles as ONE stream of chunks: the next tile's first fragments and weight stages are in flight while the
//     current tile is finished and stored; XCD-aware tile order (the cout blocks of a pixel tile are neighbours: the activations are read from
//     HBM once);
//   * epilogues of the family (conv_bf16_epi.h): bias / activation / bf16 or fp32 store / bf16 residual / B-operand ("kpack") store, or the row
//     maxima of the logits GEMM instead of a stored result.
// Chosen by plan_x (conv_bf16.hip) for 1x1 descriptors whose sources are multiples of 64 channels (>= 4 chunks) and whose cout is a multiple
// of 256; everything else keeps the ring kernel.
#include "conv_bf16_epi.h"

namespace gpemsr {

constexpr int GD_BN = 256, GD_NPIX = 256, GD_CK = 64, GD_NT = GD_BN / 32, GD_RING = 4;
constexpr int GD_STAGE = 8 * GD_BN * 16;          // bytes of one weight stage image: [8 pieces][256 couts][8 bf16]

template <bool LEAN, int XEPI>
__global__ __launch_bounds__(512, 1) void gemm_direct_bf16_kernel(XParams P) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  float* const bias_lds = reinterpret_cast<float*>(gsm + GD_RING * GD_STAGE);
  x_stage_bias(P, bias_lds, P.nbias, 512);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int hw = P.oh * P.ow;

  const unsigned short* srcp[GPEMSR_MAX_SRC];
  long long src_istride[GPEMSR_MAX_SRC];
  unsigned src_pixb[GPEMSR_MAX_SRC];
  int src_nch[GPEMSR_MAX_SRC];
  int nchunks = 0;
#pragma unroll
  for (int s = 0; s < GPEMSR_MAX_SRC; ++s) {
    srcp[s] = P.src[s]; src_istride[s] = P.img_stride[s]; src_pixb[s] = (unsigned)P.ld[s] * 2u; src_nch[s] = s < P.nsrc ? P.c[s] / GD_CK : 0;
    nchunks += src_nch[s];
  }

  const int ntiles = P.nblocks, grid = gridDim.x;
  const int T_me = (ntiles - (int)blockIdx.x + grid - 1) / grid;
  const int TS = T_me * nchunks;                       // chunks (= weight stages) of this workgroup's whole stream
  auto tile_geo = [&](int ti) -> XGeo {
    int t = (int)blockIdx.x + ti * grid;
    {   // XCD-aware remap (bijective): consecutive logical tiles of concurrently running workgroups share an XCD / L2
      const int q = ntiles / 8, r = ntiles % 8, xcd = t % 8;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t / 8;
    }
    int tn, tx;
    xdivmod(t, P.tiles_n, P.mg_n, t, tn);
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    XGeo g;
    g.img = t; g.n0 = tn * GD_BN; g.oy0 = 0; g.ox0 = tx * GD_NPIX; g.tile_in_img = tx;
    return g;
  };
  XGeo cur = tile_geo(0);
  XGeo nxt = T_me > 1 ? tile_geo(1) : cur;

  // ---- A cursor: chunk `a_next` of the stream -> registers ----
  int a_next = 0, a_chunk = 0, a_src = 0, a_c0 = 0;
  bool a_in_nxt = false;                               // the cursor has crossed into the tile after `cur`
  auto load_a = [&](bf16x8 (&fa)[4]) {
    const XGeo& g = a_in_nxt ? nxt : cur;
    const unsigned short* sp = nullptr; unsigned pixb = 0; int cs = 0;
#pragma unroll
    for (int s = 0; s < GPEMSR_MAX_SRC; ++s)
      if (s == a_src) { sp = srcp[s] + (long long)g.img * src_istride[s] + a_c0; pixb = src_pixb[s]; cs = src_nch[s] * GD_CK; }
    int p = g.ox0 + wave * 32 + li;
    p = p < hw ? p : hw - 1;                           // rows past the image: any valid address (their results are never stored)
    // plain loads (the source pointer comes straight from the kernel arguments, so they are global_load, not flat_load: a flat load
    // counts on BOTH wait counters and would drain the DMA queue)
    const char* base = reinterpret_cast<const char*>(sp) + ((size_t)((unsigned)p * pixb) + (unsigned)lh * 16u);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const bf16x8*>(base + ks * 32);
    ++a_next; a_c0 += GD_CK;
    if (a_c0 >= cs) { a_c0 = 0; ++a_src; }
    if (++a_chunk == nchunks) { a_chunk = 0; a_src = 0; a_c0 = 0; a_in_nxt = true; }
  };
  // ---- B cursor: weight stage `b_next` -> ring slot b_next % RING, 4 DMA instructions per wave ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)gsm + (unsigned)wave * 1024u);
  int b_next = 0, b_chunk = 0, b_dst = 0;
  bool b_in_nxt = false;
  auto issue_b = [&]() {
    const XGeo& g = b_in_nxt ? nxt : cur;
    const unsigned short* wp = P.weight + (long long)g.img * P.w_img_stride + (long long)b_chunk * 8 * 8 * P.cout;      // [chunk][piece][cout][8]
    const void* wpu = xuni_ptr(wp);
    const unsigned lb = xuni(lds0 + (unsigned)(b_dst * GD_STAGE));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 512;                     // slot = piece e / 256, row e % 256 of the stage image
      xglds16((unsigned)((e >> 8) * P.cout + g.n0 + (e & 255)) * 16u, wpu, lb + i * (512 * 16u));
    }
    ++b_next;
    if (++b_dst == GD_RING) b_dst = 0;
    if (++b_chunk == nchunks) { b_chunk = 0; b_in_nxt = true; }
  };

  // ---- prologue: A(0) -> registers, stages 0 .. RING-2 ----
  bf16x8 fa0[4], fa1[4];
  load_a(fa0);
  for (int j = 0; j < GD_RING - 1 && b_next < TS; ++j) issue_b();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const unsigned gsm_lds = xlds_addr(gsm);
  const unsigned b_frag = (unsigned)(li * 16 + lh * (GD_BN * 16));
  int slot = 0, s_glob = 0;

  // One chunk: (Ac) = this chunk's fragments, (An) receives the next chunk's.  WAIT DISCIPLINE: the LDS-DMA instructions are inline asm, so the
  // compiler's vmcnt bookkeeping sees only the plain A loads; vmcnt is in order, so a wait the compiler computes for an A load also covers
  // every DMA issued BEFORE it and none issued after.  Hence, per chunk: k-step 0 first (its wait for Ac -- loaded a chunk ago -- finds
  // nothing younger in the queue than the DMA of that same chunk, which has had a whole chunk to land), THEN the refill DMA of stage
  // s + RING - 1, THEN the loads of the next chunk's fragments, then k-steps 1-3.  A wave that has passed k-step 0 of chunk s has therefore
  // seen all its DMA up to stage s + RING - 2 land; the barrier at the end of chunk s makes that true for every wave before stage s + 1 is read.
  auto chunk = [&](f32x16 (&acc)[1][GD_NT], bf16x8 (&Ac)[4], bf16x8 (&An)[4]) {
    const unsigned B = gsm_lds + (unsigned)(slot * GD_STAGE) + b_frag;
    // all four fragments of this chunk are waited for HERE (an empty statement that reads them), i.e. before the refill DMA is issued: a
    // wait placed at k-step 1-3 would be computed without the DMA instructions in between and drain them
    asm volatile("" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Ac[2]), "+v"(Ac[3]));
    bf16x8 fb[2][4];
    auto load_half = [&](int set, int hs) {            // half-step hs = 2 ks + h2: weight fragments of couts 128 h2 .. 128 h2 + 127
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[set][j] = xlds_read16(B + (unsigned)(2 * (hs >> 1) * (GD_BN * 16) + (4 * (hs & 1) + j) * 512));
    };
    load_half(0, 0);
#pragma unroll
    for (int hs = 0; hs < 8; ++hs) {                   // software pipelined by one half-step (static register sets)
      if (hs + 1 < 8) load_half((hs + 1) & 1, hs + 1);
      if (hs == 2) {
        if (b_next < TS) issue_b();                    // -> the slot the last barrier freed
        if (a_next < TS) load_a(An);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[0][4 * (hs & 1) + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[hs & 1][j], Ac[hs >> 1], acc[0][4 * (hs & 1) + j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (++slot == GD_RING) slot = 0;
    ++s_glob;
  };

  for (int ti = 0; ti < T_me; ++ti) {
    f32x16 acc[1][GD_NT];
#pragma unroll
    for (int nt = 0; nt < GD_NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][nt][r] = 0.f;
    // (host: the chunk count is even, so the two fragment sets alternate without a copy)
    for (int cc = 0; cc < nchunks; cc += 2) {
      chunk(acc, fa0, fa1);
      chunk(acc, fa1, fa0);
    }
    const XGeo g = cur;
    cur = nxt;
    a_in_nxt = false; b_in_nxt = false;                // both cursors are inside the new `cur` by now (RING - 1 <= chunks per tile)
    if (ti + 2 < T_me) nxt = tile_geo(ti + 2);
    {
      int li2 = li, lh2 = lh;                          // opaque per-tile copies: keeps the epilogue's addressing out of the main loop's live set
      asm volatile("" : "+v"(li2), "+v"(lh2));
      if constexpr (XEPI == 1) x_epilogue_rowmax<1, GD_NT>(P, g, acc, wave * 32, 0, 0, 1, bias_lds, li2, lh2);
      else x_epilogue<1, GD_NT, true, false, LEAN>(P, g, acc, wave * 32, 0, g.tile_in_img * 8 + wave, bias_lds, li2, lh2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (no LDS-DMA of this workgroup may still be in flight when its LDS is handed on)
}

int launch_gemm_direct(const XParams& P, bool lean, bool rowmax, size_t lds, hipStream_t st) {
  static dev_once_t done{0};
  if (dev_once_begin(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_direct_bf16_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_direct_bf16_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_direct_bf16_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d_bf16: cannot raise the dynamic LDS limit");
    dev_once_done(done);
  }
  const int cus = device_cus();
  const int grid = P.nblocks < cus ? P.nblocks : cus;
  if (rowmax) hipLaunchKernelGGL((gemm_direct_bf16_kernel<false, 1>), dim3(grid), dim3(512), lds, st, P);
  else if (lean) hipLaunchKernelGGL((gemm_direct_bf16_kernel<true, 0>), dim3(grid), dim3(512), lds, st, P);
  else hipLaunchKernelGGL((gemm_direct_bf16_kernel<false, 0>), dim3(grid), dim3(512), lds, st, P);
  return check_launch("gemm_direct_bf16_kernel");
}

}  // namespace gpemsr
