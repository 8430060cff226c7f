// Flash-style single-head attention of the VQGAN NonLocalBlock (R:model/blocks.py:61-83), bf16 data path, C = 512 channels:
//     A[i] = sum_j softmax_j(q_i . k_j) v_j      (the C^-1/2 of :76 is folded into q's weights; v's bias is added after the product)
// with the T x T score matrix never written to memory (round 2 materialised it: 10.8 GB of HBM traffic per step and three launches --
// q.k^T at 0.21, softmax, P.v at 0.35 of the bf16 matrix peak).
//
// One 256-thread workgroup (4 waves, ONE wave per SIMD: each wave has the whole 512-register file) owns 128 queries, a wave 32 of
// them, and walks the keys in tiles of 32:
//   S^T[key][query] = K_tile . Q^T        32 x v_mfma_f32_32x32x16_bf16: K rows from LDS (A operand), Q^T held in registers (B operand,
//                                          128 registers); the accumulator has lane = query, registers = 16 of the 32 keys (the lane
//                                          32 away holds the other 16) -- the softmax statistics of a query are lane-local scalars
//   online softmax                         tile max (15 v_max + one half-wave exchange), running max m / sum l, O rescaled only when a
//                                          maximum actually grows (wave-uniform branch), p = exp2(s * log2e - m * log2e)
//   P^T as the next B operand              8 v_cvt_pk_bf16_f32, NO cross-lane traffic: the k-slot order of the P.V product is whatever
//                                          order the accumulator registers hold the keys in, and V^T is stored in that order (below)
//   O^T[d][query] += V^T_tile . P^T        32 MFMAs: V^T rows (channels d) from LDS, 16 accumulator tiles = 256 registers
// K / V^T tiles (32 KB each) arrive by LDS-DMA into a 2-deep ring, issued between the MFMAs of the previous tile; one barrier per tile.
//
// Operand layouts (produced by the 1x1 convolutions' "kpack" epilogue, gpemsr_conv2d_bf16):
//   q    [n][T][C]          bf16 rows (NHWC)
//   kp   [n][C/8][T][8]     16-byte piece = 8 consecutive channels of one key  -> A fragment of S^T by one ds_read_b128
//   vtp  [n][T/8][C][8]     16-byte piece = 8 keys of one channel, keys of every 16-group stored in the order
//                           0-3, 8-11, 4-7, 12-15 (gpemsr_pack_rows_bf16_ex(..., perm16 = 1) on the B operand of the v^T product):
//                           piece (2 kk + lh) of a 32-key tile then holds exactly the keys whose scores sit in accumulator
//                           registers 8 kk .. 8 kk + 7 of lane half lh.
// Replaces torch.bmm x2 + softmax of R:model/blocks.py:75-80 for T % 128 == 0, C == 512 (the engine keeps the three-launch path and
// the fp32 ragged path for everything else).
#include "bf16_common.h"
#include "attn_agpr.inc"

namespace gpemsr {

struct AttnParams {
  const unsigned short* q; int q_ld;        // [n][T][q_ld]
  const unsigned short* kp;                 // [n][C/8][T][8]
  const unsigned short* vtp;                // [n][T/8][C][8] (keys permuted inside 16-groups)
  const float* bias_v;                      // [C] or null
  unsigned short* out; int out_ld;          // [n][T][out_ld]
  int n, T;
  int qblocks;                              // T / 128
};

constexpr int AT_C = 512, AT_BK = 32;       // channels, keys per tile
constexpr int AT_KBYTES = AT_BK * AT_C * 2; // 32 KiB per operand tile

__global__ __launch_bounds__(256, 1) void flash_attn512_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) char asm_[];
  // LDS: K ring [2][64 c8][32 keys][16 B], V^T ring [2][4 pieces][512 d][16 B]
  const unsigned lds = xlds_addr(asm_);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // workgroup -> (image, query block): the 32 query blocks of an image are 8 apart in block number, i.e. they share an XCD under the
  // round-robin placement (speed only): its L2 then serves the K / V^T stream of ONE image to all of them
  const int b = (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int img = (idx / P.qblocks) * 8 + xcd, qb = idx % P.qblocks;
  if (img >= P.n) return;
  const int T = P.T;
  const int q0 = qb * 128 + wave * 32;

  const unsigned short* kimg = P.kp + (long long)img * T * AT_C;
  const unsigned short* vimg = P.vtp + (long long)img * T * AT_C;
  const unsigned ldsw = xuni(lds + (unsigned)wave * 8192u);      // this wave's 8 KiB share of every tile image

  // DMA of tile t: K pieces e = 8 wave + i cover c8 rows 2e, 2e+1 (lanes 0-31 / 32-63) x 32 keys; V^T pieces are 1 KiB runs
  auto issue_k = [&](int t, int i) {
    const int e = wave * 8 + i;
    const unsigned voff = (unsigned)(((2 * e + lh) * T + t * AT_BK + li) * 16);
    xglds16(voff, kimg, ldsw + (unsigned)((t & 1) * AT_KBYTES + i * 1024));
  };
  auto issue_v = [&](int t, int i) {
    const unsigned voff = (unsigned)(t * AT_KBYTES + (wave * 8 + i) * 1024 + lane * 16);
    xglds16(voff, vimg, ldsw + (unsigned)(2 * AT_KBYTES + (t & 1) * AT_KBYTES + i * 1024));
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_k(0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_v(0, i);

  // Q^T fragments: lane (query li, half lh) holds channels 16 ks + 8 lh .. + 7 of its query for every k-step ks
  bf16x8 qf[32];
  {
    const unsigned short* qp = P.q + ((long long)img * T + q0 + li) * P.q_ld + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  // O^T: 16 accumulator tiles [32 d][32 queries] = 256 registers in the ACCUMULATOR half of the register file, a[0:255], addressed by
  // number from inline asm (attn_agpr.inc): the only place they fit beside Q^T's 128 arch VGPRs.  (As C variables hipcc keeps moving
  // parts of them between the two halves every tile -- 208 v_accvgpr moves + scratch traffic per iteration.)  Nothing else in this
  // kernel may live in AGPRs: every other MFMA result is pinned to arch VGPRs ("+v") and the VGPR pressure (~210) leaves no spills.
  AT_ZERO_ALL();
  float m2 = 0.f, l = 0.f;                  // reference maximum (times log2 e) of this query, sum of this lane's 16-key halves
  constexpr float LOG2E = 1.4426950408889634f;
  // The reference maximum follows the running maximum LAZILY: O and l are rescaled only when some query's tile maximum exceeds its
  // reference by more than 2^AT_THR (p <= 2^24: fp32 sums of 4096 such terms are far from overflow, and bf16 keeps 8 significant bits
  // at any magnitude), so after the first tiles the cold path below is practically never taken.
  constexpr float AT_THR = 24.f;

  const unsigned kfrag = lds + (unsigned)((lh * 32 + li) * 16);            // + (t&1) * 32K + ks * 1024
  const unsigned vfrag = lds + (unsigned)(2 * AT_KBYTES + (lh * 512 + li) * 16);   // + (t&1) * 32K + kk * 16384 + dt * 512

  const int NT = T / AT_BK;
  // ---- software pipeline: iteration t runs  [ QK^T of tile t+1  ||  softmax of tile t ]  then  P.V of tile t .  With ONE wave per SIMD
  // nothing else hides the ~100 VALU operations of the softmax, the wait states behind an MFMA chain, or a DMA round trip (first
  // version, everything in sequence: 53 % of the tile time in MFMAs), so the softmax of tile t is cut into 32 slices placed between the
  // MFMAs of the NEXT tile's score product, and the images are issued early: V^T(t+1) during that phase, K(t+2) during the first half
  // of P.V(t); both are only waited for at the barrier that ends the iteration.
  if (NT > 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_k(1, i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x16 sc, sn;                       // scores of the tile whose softmax is due / of the next tile
  auto qk_step = [&](f32x16& acc, const bf16x8& kfr, int ks) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(kfr), "v"(qf[ks]));      // arch VGPRs: the softmax reads them
  };
  {   // S^T(0)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    bf16x8 kf[2];
    kf[0] = xlds_read16(kfrag);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (ks + 1 < 32) kf[(ks + 1) & 1] = xlds_read16(kfrag + (unsigned)((ks + 1) * 1024));
      qk_step(sc, kf[ks & 1], ks);
    }
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sc));     // inline-asm MFMA -> VALU read of its result: wait states the compiler cannot count
  }

  for (int t = 0; t < NT; ++t) {
    const unsigned kb = kfrag + (unsigned)(((t + 1) & 1) * AT_KBYTES), vb = vfrag + (unsigned)((t & 1) * AT_KBYTES);
    const bool more = t + 1 < NT, more2 = t + 2 < NT;
    // ---- softmax of tile t in 32 slices (lane = query; this lane's 16 keys + the partner half's 16) ----
    float pmax[4], mt = 0.f, p[16];
    unsigned pku[8];
    auto slice = [&](const int i) {                     // i is a compile-time constant at every call site
      if (i < 4) pmax[i] = fmaxf(fmaxf(sc[4 * i], sc[4 * i + 1]), fmaxf(sc[4 * i + 2], sc[4 * i + 3]));
      else if (i == 4) mt = fmaxf(fmaxf(pmax[0], pmax[1]), fmaxf(pmax[2], pmax[3]));
      else if (i == 5) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
        mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * LOG2E;
      } else if (i == 6) {
        if (t == 0) m2 = mt;                            // (O and l are still zero: nothing to rescale)
        else if (__builtin_amdgcn_ballot_w64(mt > m2 + AT_THR) != 0ull) {
          // cold path: some query's maximum outgrew its reference; every lane moves its reference to its running maximum
          const float mn = fmaxf(m2, mt);
          const float alpha = __builtin_amdgcn_exp2f(m2 - mn);
          l *= alpha;
          AT_SCALE_ALL(alpha);
          m2 = mn;
        }
      } else if (i < 23) { const int r = i - 7; p[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], LOG2E, -m2)); l += p[r]; }
      else if (i < 31) { const int j = i - 23; pku[j] = xcvt_pk_bf16(p[2 * j], p[2 * j + 1]); }
    };
    if (more) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[r] = 0.f;
      bf16x8 kf[2];
      kf[0] = xlds_read16(kb);
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        if (ks + 1 < 32) kf[(ks + 1) & 1] = xlds_read16(kb + (unsigned)((ks + 1) * 1024));
        qk_step(sn, kf[ks & 1], ks);
        slice(ks);
        if ((ks & 3) == 3) issue_v(t + 1, ks >> 2);     // next tile's V^T image, one piece per 4 MFMAs
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) slice(i);
    }
    bf16x8 pf[2];
    {
      union { unsigned u[4]; bf16x8 v; } pk0, pk1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { pk0.u[j] = pku[j]; pk1.u[j] = pku[4 + j]; }
      pf[0] = pk0.v; pf[1] = pk1.v;
    }
    // ---- O^T += V^T_tile . P^T : per 32-channel tile dt one fragment per 16-key half ----
    bf16x8 va = xlds_read16(vb), vc;
#define AT_PV_STEP(dt)                                                                         \
    vc = xlds_read16(vb + (unsigned)(16384 + (dt) * 512));                                     \
    AT_PV_##dt(va, pf[0]);                                                                     \
    if ((dt) + 1 < 16) va = xlds_read16(vb + (unsigned)(((dt) + 1) * 512));                    \
    AT_PV_##dt(vc, pf[1]);                                                                     \
    if ((dt) < 8 && more2) issue_k(t + 2, (dt));
    AT_PV_STEP(0) AT_PV_STEP(1) AT_PV_STEP(2) AT_PV_STEP(3) AT_PV_STEP(4) AT_PV_STEP(5) AT_PV_STEP(6) AT_PV_STEP(7)
    AT_PV_STEP(8) AT_PV_STEP(9) AT_PV_STEP(10) AT_PV_STEP(11) AT_PV_STEP(12) AT_PV_STEP(13) AT_PV_STEP(14) AT_PV_STEP(15)
#undef AT_PV_STEP
    // the images issued in this iteration must have landed; everybody is done with the K image of tile t+1's product and with V^T(t)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // sn was written by inline-asm MFMAs: the hazard recognizer does not count wait states for the VALU copy below.  32 P.V MFMAs and a
    // barrier lie in between, so these nops never delay anything -- they make the distance a property of the source, not of scheduling.
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sn));
    sc = sn;
  }

  // ---- epilogue: O / l + v bias -> bf16 rows ----
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.f / l;
  unsigned short* orow = P.out + ((long long)img * T + q0 + li) * P.out_ld;
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");              // the last MFMA's accumulators are read below
#define AT_OUT_TILE(dt) {                                                                                                            \
    float v[16];                                                                                                                     \
    AT_READ_##dt(v);                                                                                                                 \
    _Pragma("unroll") for (int gp = 0; gp < 4; gp += 2) {                                                                            \
      float w8[8];                                                                                                                   \
      /* registers 4 gp + j / 4 (gp + 1) + j hold channels 8 gp + 4 lh + j / 8 (gp + 1) + 4 lh + j: after the half-wave swap lane */  \
      /* (li, lh) owns the 8 consecutive channels 8 (gp + lh) .. + 7 of its query                                                */  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                                \
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * gp + j] * inv), __float_as_uint(v[4 * gp + 4 + j] * inv), false, false); \
        w8[j] = __uint_as_float(sw[0]); w8[4 + j] = __uint_as_float(sw[1]);                                                          \
      }                                                                                                                              \
      const int d0 = 32 * (dt) + 8 * (gp + lh);                                                                                      \
      if (P.bias_v) {                                                                                                                \
        const float4 b0 = *reinterpret_cast<const float4*>(P.bias_v + d0), b1 = *reinterpret_cast<const float4*>(P.bias_v + d0 + 4); \
        w8[0] += b0.x; w8[1] += b0.y; w8[2] += b0.z; w8[3] += b0.w; w8[4] += b1.x; w8[5] += b1.y; w8[6] += b1.z; w8[7] += b1.w;      \
      }                                                                                                                              \
      *reinterpret_cast<uint4*>(orow + d0) = make_uint4(xcvt_pk_bf16(w8[0], w8[1]), xcvt_pk_bf16(w8[2], w8[3]), xcvt_pk_bf16(w8[4], w8[5]), xcvt_pk_bf16(w8[6], w8[7])); \
    } }
  AT_OUT_TILE(0) AT_OUT_TILE(1) AT_OUT_TILE(2) AT_OUT_TILE(3) AT_OUT_TILE(4) AT_OUT_TILE(5) AT_OUT_TILE(6) AT_OUT_TILE(7)
  AT_OUT_TILE(8) AT_OUT_TILE(9) AT_OUT_TILE(10) AT_OUT_TILE(11) AT_OUT_TILE(12) AT_OUT_TILE(13) AT_OUT_TILE(14) AT_OUT_TILE(15)
#undef AT_OUT_TILE
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_flash_attention_bf16(const void* q, int q_ld, const void* kp, const void* vtp, const float* bias_v, int n, int tokens, int channels,
                                           void* out, int out_ld, void* stream) {
  GP_REQUIRE(q && kp && vtp && out && n > 0, "flash_attention_bf16: null pointer / empty batch");
  GP_REQUIRE(channels == AT_C && tokens % 128 == 0 && tokens >= 128 && (long long)tokens * AT_C * 2 < (1ll << 32),
             "flash_attention_bf16: needs channels == 512 and tokens %% 128 == 0 (got %d, %d)", channels, tokens);
  GP_REQUIRE(q_ld % 8 == 0 && out_ld % 8 == 0 && q_ld >= AT_C && out_ld >= AT_C && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(kp) |
             reinterpret_cast<uintptr_t>(vtp) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias_v)) & 15) == 0, "flash_attention_bf16: alignment");
  // the attribute is per device and the call is cheap: set it on every launch (a process may drive more than one GPU, from more than one thread)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return fail(GPEMSR_ELAUNCH, "flash_attention_bf16: cannot raise the dynamic LDS limit");
  AttnParams P{};
  P.q = reinterpret_cast<const unsigned short*>(q); P.q_ld = q_ld;
  P.kp = reinterpret_cast<const unsigned short*>(kp); P.vtp = reinterpret_cast<const unsigned short*>(vtp);
  P.bias_v = bias_v; P.out = reinterpret_cast<unsigned short*>(out); P.out_ld = out_ld;
  P.n = n; P.T = tokens; P.qblocks = tokens / 128;
  const int groups = (n + 7) / 8;
  hipLaunchKernelGGL(flash_attn512_kernel, dim3(8 * groups * P.qblocks), dim3(256), 4 * AT_KBYTES, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("flash_attn512_kernel");
}
