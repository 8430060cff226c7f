// Epilogues of the bf16 implicit-GEMM family, shared by the kernels of conv_bf16.hip and gemm_bf16.hip: they work on D^T accumulator tiles
// (v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the row operand: registers = couts, lanes = pixels) and know nothing about how the tiles were
// produced.  (Moved out of conv_bf16.hip unchanged.)
#pragma once
#include "conv_bf16.h"

namespace gpemsr {

// Epilogue of one wave's MT x NT accumulator tiles, straight from registers.  The accumulators are D^T (registers = 16 couts
// (r&3) + 8 (r>>2) + 4 lh of a 32-cout tile, lanes = 32 pixels): bias (from LDS: a global load here would drain the DMA
// prefetches through the in-order vmcnt) and activation are applied per register, then two v_permlane32_swap per register
// pair give lane (li, lh) the 8 consecutive couts 8 (gp + lh) .. + 7 of its pixel: residual, per-pixel multiplier, optional
// fp32 copy, 16-byte store.  GroupNorm partial sums (conv + bias, valid pixels): butterfly over the 32 pixel lanes.
//
// CODE SIZE is a first-order cost here: the epilogue is instantiated once per accumulator tile, every store map / residual type /
// ragged-channel tail multiplies it, and the kernels had grown to 60-90 KB against a 64 KB instruction cache shared by two CUs --
// in-kernel stamps: the epilogue took 9.4k cycles per tile of the 64-channel kernel and 4.7k once the rarely used paths were
// compiled out (-21 % per tile; -15 % on a 256 -> 256 layer, -17 % on a 512-deep 1x1).  Hence LEAN: the form nearly every layer
// of the network uses (plain NHWC bf16 store, cout % 8 == 0, optional bf16 residual, none / ReLU / LeakyReLU, optional GroupNorm
// partial sums) is compiled without the other paths; the host picks it whenever the descriptor allows (plan_x).  The transposed
// kernels know their store map at compile time as well.
// PRE (lean form only): the layer has a residual and / or a per-pixel multiplier -- they are fetched branch-free before the stores;
// layers without either take the PRE = false copy, which has no load at all (a dummy load would put its latency, and a run-time
// `if` around the loads a vmcnt(0), in front of the stores of every tile).
// PRE: 0 = residual / multiplier loaded where they are used (under a run-time `if`); 1 = prefetched branch-free; 2 = the layer has
// neither (compiled out).  The two-copy form (1 / 2 behind one uniform branch) is used by the weights-resident kernel only: on the
// ring kernel's 168-register budget the second copy spills (measured: slower overall).
template <int MT, int NT, bool GEMM, bool CONVT, bool LEAN, int PRE>
__device__ __forceinline__ void x_epilogue_stores(const XParams& Pfull, const XGeo& g, f32x16 (&acc)[MT][NT], int pix_base, int cout_base,
                                                  const float* bias_lds, int li, int lh) {
  XParams P = Pfull;          // (a by-value view whose fixed fields fold at compile time)
  if (CONVT) P.store_mode = XS_CONVT;
  else if (GEMM && P.store_mode != XS_KPACK) P.store_mode = XS_PLAIN;
  if (LEAN) { if (!CONVT) P.store_mode = XS_PLAIN; P.out32 = nullptr; P.out_f32 = 0; P.res_f32 = 0; }
  if (LEAN && CONVT) { P.residual = nullptr; P.pixmul = nullptr; }
  if (LEAN && PRE == 2) { P.residual = nullptr; P.pixmul = nullptr; }
  const long long img_pix0 = (long long)g.img * P.OH * P.OW;
  const bool up = P.store_mode == XS_PIXSHUF || P.store_mode == XS_CONVT;
  // packed fast path (below): its bias values, ALL requested before the first use (one LDS round trip per wave and tile, not one
  // per four channels; reads are branch-free)
  const bool packed = LEAN && PRE != 1 && P.residual == nullptr && P.pixmul == nullptr && P.act != GPEMSR_ACT_LRELU;
  constexpr bool HOIST_ALL = NT <= 2;          // (four accumulator tiles per wave, the transposed form: 64 registers of bias spill)
  float4 bpre[HOIST_ALL ? NT : 1][4];
  auto bias_of = [&](int nt, int q) -> float4 {
    const int c0 = g.n0 + cout_base + nt * 32 + 8 * q + 4 * lh;
    const int bc = CONVT ? (c0 >> 7) * 32 + (c0 & 31) : c0;
    float4 b = *reinterpret_cast<const float4*>(bias_lds + (c0 < P.cout ? bc : 0));
    if (!P.bias) b = make_float4(0.f, 0.f, 0.f, 0.f);                    // (uniform; a caller may have folded the bias into the accumulators)
    return b;
  };
  if (packed && HOIST_ALL) {
#pragma unroll
    for (int nt = 0; nt < (HOIST_ALL ? NT : 1); ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) bpre[nt][q] = bias_of(nt, q);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = pix_base + mt * 32 + li;
    int oy, ox; bool pok;
    if (GEMM) { oy = 0; ox = g.ox0 + p; pok = ox < P.oh * P.ow; }
    else { oy = g.oy0 + (p >> 5); ox = g.ox0 + (p & 31); pok = oy < P.oh && ox < P.ow; }
    // LEAN: the residual pieces and the per-pixel multiplier of this pixel are fetched BEFORE its first store -- a load issued after
    // a store waits for that store on the in-order vmcnt, which made every conv2 of a residual block pay the store latency 4 times
    // BRANCH-FREE on purpose: loads under a run-time `if` make the compiler wait with vmcnt(0) at every later use -- i.e. behind the
    // stores again.  Lanes / layers without a residual read 16 bytes of the weight array instead (one cached address) and select 0.
    uint4 rpre[NT][2];
    float mpre = 1.f;
    if (LEAN && PRE == 1) {
      const long long opl = img_pix0 + (GEMM ? (long long)ox : (long long)oy * P.OW + ox);
      const unsigned short* rbase = reinterpret_cast<const unsigned short*>(P.residual);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int nidx = g.n0 + cout_base + nt * 32 + 8 * (2 * h2 + lh);
          const bool use = rbase != nullptr && pok && nidx < P.cout;
          const unsigned short* ap = use ? rbase + opl * P.res_ld + nidx : P.weight;
          uint4 u = *reinterpret_cast<const uint4*>(ap);
          if (!use) u = make_uint4(0u, 0u, 0u, 0u);
          rpre[nt][h2] = u;
        }
      const bool usem = P.pixmul != nullptr && pok;
      const float mv = *(usem ? P.pixmul + opl : reinterpret_cast<const float*>(P.weight));
      mpre = usem ? mv : 1.f;
    }
    // Packed fast path (lean layers without residual / per-pixel multiplier / LeakyReLU, i.e. most of the bf16 path): convert to
    // bf16 FIRST, ReLU as a packed signed-integer max (bf16 sign bit = int16 sign bit), half-wave swaps on packed pairs, one output
    // address per pixel -- 16 + 8 + 8 + 4 vector instructions per 32 x 32 accumulator tile instead of ~125 (stamps of the
    // weights-resident kernel: the general form cost as many issue cycles as a 64-channel tile's MFMAs)
    if (packed) {
      // (transposed convolution: an accumulator tile's 32 rows are 32 channels of ONE output phase q = (row >> 5) & 3 of a 128-row block)
      const int opix = GEMM ? ox : (CONVT ? 2 * oy * P.OW + 2 * ox : oy * P.OW + ox);
      unsigned short* const op = reinterpret_cast<unsigned short*>(P.out) + (img_pix0 + opix) * P.out_ld;
      const bool relu = P.act == GPEMSR_ACT_RELU;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cb0 = g.n0 + cout_base + nt * 32;
        unsigned pk[4][2];
        if (!HOIST_ALL) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bpre[0][q] = bias_of(nt, q);        // one LDS round trip per accumulator tile
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = bpre[HOIST_ALL ? nt : 0][q];
          pk[q][0] = xcvt_pk_bf16(acc[mt][nt][4 * q] + b4.x, acc[mt][nt][4 * q + 1] + b4.y);
          pk[q][1] = xcvt_pk_bf16(acc[mt][nt][4 * q + 2] + b4.z, acc[mt][nt][4 * q + 3] + b4.w);
        }
        if (relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[q][j]) : "v"(pk[q][j]));
        }
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 1][1], false, false);
          const int nidx = cb0 + 8 * (gp + lh);
          long long off = nidx;
          if (CONVT) { const int ph = (nidx & 127) >> 5; off = (long long)((ph >> 1) * P.OW + (ph & 1)) * P.out_ld + (nidx >> 7) * 32 + (nidx & 31); }
          if (pok && nidx < P.cout) *reinterpret_cast<uint4*>(op + off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
      continue;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cb0 = g.n0 + cout_base + nt * 32;            // first cout (GEMM column) of this 32-row accumulator tile
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[mt][nt][r];
      if (P.bias) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = cb0 + 8 * gq + 4 * lh;
          int bc = c0;
          if (P.store_mode == XS_CONVT) bc = (c0 >> 7) * 32 + (c0 & 31);     // bias is per true output channel
          if (c0 < P.cout) {                                                    // (the LDS copy is zero-padded to a multiple of 8)
            const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + bc);
            v[4 * gq] += b4.x; v[4 * gq + 1] += b4.y; v[4 * gq + 2] += b4.z; v[4 * gq + 3] += b4.w;
          }
        }
      }
      if (P.act == GPEMSR_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (P.act == GPEMSR_ACT_LRELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.1f * v[r]);
      } else if (!LEAN && P.act != GPEMSR_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], P.act);
      }
      // two register pairs of groups (0,1) and (2,3): after the half-wave swaps lane (li, lh) owns couts 8*(gp + lh) .. +7
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * gp + j]), __float_as_uint(v[4 * gp + 4 + j]), false, false);
          w8[j] = __uint_as_float(sw[0]); w8[4 + j] = __uint_as_float(sw[1]);
        }
        const int nidx = cb0 + 8 * (gp + lh);
        const int nvalid = LEAN ? ((P.cout - nidx) > 0 ? 8 : 0) : ((P.cout - nidx) < 8 ? (P.cout - nidx) : 8);
        if (!pok || nvalid <= 0) continue;
        int ch = nidx, sy = 0, sx = 0;
        if (P.store_mode == XS_PIXSHUF) { const int q = nidx / P.cq; ch = nidx - q * P.cq; sy = q >> 1; sx = q & 1; }
        else if (P.store_mode == XS_CONVT) { const int blk = nidx >> 7, q = (nidx & 127) >> 5; ch = blk * 32 + (nidx & 31); sy = q >> 1; sx = q & 1; }
        const int opix = GEMM ? ox : (up ? (2 * oy + sy) * P.OW + 2 * ox + sx : oy * P.OW + ox);
        const bool full = nvalid == 8;
        if (LEAN && PRE == 1) {
          const uint4 u = rpre[nt][gp >> 1];
          w8[0] += xbf_lo(u.x); w8[1] += xbf_hi(u.x); w8[2] += xbf_lo(u.y); w8[3] += xbf_hi(u.y);
          w8[4] += xbf_lo(u.z); w8[5] += xbf_hi(u.z); w8[6] += xbf_lo(u.w); w8[7] += xbf_hi(u.w);
#pragma unroll
          for (int k = 0; k < 8; ++k) w8[k] *= mpre;
        } else if (P.residual) {
          if (P.res_f32) {
            const float* rp = reinterpret_cast<const float*>(P.residual) + (img_pix0 + opix) * P.res_ld + ch;
            if (full) {
              const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
              w8[0] += r0.x; w8[1] += r0.y; w8[2] += r0.z; w8[3] += r0.w; w8[4] += r1.x; w8[5] += r1.y; w8[6] += r1.z; w8[7] += r1.w;
            } else {
              for (int k = 0; k < nvalid; ++k) w8[k] += rp[k];
            }
          } else {
            const unsigned short* rp = reinterpret_cast<const unsigned short*>(P.residual) + (img_pix0 + opix) * P.res_ld + ch;
            if (full) {
              const uint4 u = *reinterpret_cast<const uint4*>(rp);
              w8[0] += xbf_lo(u.x); w8[1] += xbf_hi(u.x); w8[2] += xbf_lo(u.y); w8[3] += xbf_hi(u.y);
              w8[4] += xbf_lo(u.z); w8[5] += xbf_hi(u.z); w8[6] += xbf_lo(u.w); w8[7] += xbf_hi(u.w);
            } else {
              for (int k = 0; k < nvalid; ++k) w8[k] += __uint_as_float((unsigned)rp[k] << 16);
            }
          }
        }
        if (!(LEAN && PRE == 1) && P.pixmul) {
          const float m = P.pixmul[img_pix0 + opix];
#pragma unroll
          for (int k = 0; k < 8; ++k) w8[k] *= m;
        }
        if (P.out32) {
          float* o32 = P.out32 + (img_pix0 + opix) * P.out32_ld + ch;
          if (full) {
            *reinterpret_cast<float4*>(o32) = make_float4(w8[0], w8[1], w8[2], w8[3]);
            *reinterpret_cast<float4*>(o32 + 4) = make_float4(w8[4], w8[5], w8[6], w8[7]);
          } else {
            for (int k = 0; k < nvalid; ++k) o32[k] = w8[k];
          }
        }
        if (P.out_f32) {
          float* op = reinterpret_cast<float*>(P.out) + (img_pix0 + opix) * P.out_ld + ch;
          if (full) {
            *reinterpret_cast<float4*>(op) = make_float4(w8[0], w8[1], w8[2], w8[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(w8[4], w8[5], w8[6], w8[7]);
          } else {
            for (int k = 0; k < nvalid; ++k) op[k] = w8[k];
          }
        } else {
          const uint4 pk = make_uint4(xcvt_pk_bf16(w8[0], w8[1]), xcvt_pk_bf16(w8[2], w8[3]), xcvt_pk_bf16(w8[4], w8[5]), xcvt_pk_bf16(w8[6], w8[7]));
          unsigned short* op;
          if (P.store_mode == XS_KPACK) op = reinterpret_cast<unsigned short*>(P.out) + (long long)g.img * P.kpack_img_stride + ((long long)(ch >> 3) * (P.OH * P.OW) + opix) * 8;
          else op = reinterpret_cast<unsigned short*>(P.out) + (img_pix0 + opix) * P.out_ld + ch;
          if (full) {
            *reinterpret_cast<uint4*>(op) = pk;
          } else {
            const unsigned wv[4] = {pk.x, pk.y, pk.z, pk.w};
            for (int k = 0; k < nvalid; ++k) op[k] = (unsigned short)((k & 1) ? (wv[k >> 1] >> 16) : (wv[k >> 1] & 0xFFFFu));
          }
        }
      }
      // keep the accumulator tiles' epilogues sequential: hoisting the loads of later tiles costs more registers than a wave
      // with 128 accumulator registers has (the stores are fire-and-forget, the loads are few)
      asm volatile("" ::: "memory");
    }
  }
}

template <int MT, int NT, bool GEMM, bool CONVT = false, bool LEAN = false, bool TWO_COPIES = false>
__device__ __forceinline__ void x_epilogue(const XParams& P, const XGeo& g, f32x16 (&acc)[MT][NT], int pix_base, int cout_base, int gn_part,
                                           const float* bias_lds, int li, int lh) {
  if (LEAN && TWO_COPIES) {
    if (P.residual || P.pixmul) x_epilogue_stores<MT, NT, GEMM, CONVT, LEAN, 1>(P, g, acc, pix_base, cout_base, bias_lds, li, lh);
    else x_epilogue_stores<MT, NT, GEMM, CONVT, LEAN, 2>(P, g, acc, pix_base, cout_base, bias_lds, li, lh);
  } else {
    x_epilogue_stores<MT, NT, GEMM, CONVT, LEAN, 0>(P, g, acc, pix_base, cout_base, bias_lds, li, lh);
  }
  if (P.gn_ws) {
    // Partial sums of (conv + bias) and its square over the tile's valid pixels, per channel -- or per 2 / 4 neighbouring channels
    // when they belong to one GroupNorm group (gn_cpg): registers 4q .. 4q+3 of a lane are 4 consecutive channels, so that sum is
    // free, and every value left costs a 32-lane reduction.  The reduction runs on DPP adds (quad swaps, half-row and row mirrors,
    // one row broadcast: 5 VALU operations per value); the first version's shuffles went through the LDS crossbar and cost the
    // 64-channel VQGAN layers half of their time again.
    const int gs = (P.gn_cpg % 4 == 0 && P.gn_cpg > 0) ? 4 : ((P.gn_cpg % 2 == 0 && P.gn_cpg > 0) ? 2 : 1);
    auto row_sum32 = [](float v) -> float {           // lanes 16-31 (48-63) end up with the sum over lanes 0-31 (32-63)
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));    // row_mirror
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast15 into rows 1, 3
      return v;
    };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cb0 = g.n0 + cout_base + nt * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = cb0 + 8 * q + 4 * lh;             // this lane's channels c0 .. c0+3 (registers 4q .. 4q+3)
        float sm[4], sq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { sm[j] = 0.f; sq[j] = 0.f; }
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.bias && c0 < P.cout) { const float4 t = *reinterpret_cast<const float4*>(bias_lds + c0); b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int p = pix_base + mt * 32 + li;
          bool pok;
          if (GEMM) pok = g.ox0 + p < P.oh * P.ow;
          else pok = (g.oy0 + (p >> 5)) < P.oh && (g.ox0 + (p & 31)) < P.ow;
          if (pok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float v = acc[mt][nt][4 * q + j] + b4[j]; sm[j] += v; sq[j] = fmaf(v, v, sq[j]); }
          }
        }
        if (gs == 4) { sm[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]); sq[0] = (sq[0] + sq[1]) + (sq[2] + sq[3]); sm[1] = sm[2] = sm[3] = 0.f; sq[1] = sq[2] = sq[3] = 0.f; }
        else if (gs == 2) { sm[0] += sm[1]; sq[0] += sq[1]; sm[2] += sm[3]; sq[2] += sq[3]; sm[1] = sm[3] = 0.f; sq[1] = sq[3] = 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j % gs == 0) { sm[j] = row_sum32(sm[j]); sq[j] = row_sum32(sq[j]); }
        if (li == 16 && c0 < P.cout) {                    // (couts are a multiple of 4 wherever partial sums are requested: host)
          float* wsp = P.gn_ws + (((long long)g.img * P.gn_parts + gn_part) * P.cout + c0) * 2;
          *reinterpret_cast<float4*>(wsp) = make_float4(sm[0], sq[0], sm[1], sq[1]);
          *reinterpret_cast<float4*>(wsp + 4) = make_float4(sm[2], sq[2], sm[3], sq[3]);
        }
      }
    }
  }
}

// XEPI == 1: row maxima instead of a stored result (GEMM form; the arg-max of the indexer's logits, R:model/codebook.py:34-43 via
// R:model/indexer.py:100).  The accumulators are D^T: lane = GEMM row (pixel), registers = 16 of the 32 columns of a tile (the other
// 16 sit in the partner half-wave).  Every lane scans its columns in increasing order (strict >, so the lowest column wins a tie),
// one v_permlane32_swap joins the halves, and lanes 0-31 write (value, column) for their row: ws[row][part = column tile * WN + wn].
template <int MT, int NT>
__device__ __forceinline__ void x_epilogue_rowmax(const XParams& P, const XGeo& g, f32x16 (&acc)[MT][NT], int pix_base, int cout_base, int wn, int WN_,
                                                  const float* bias_lds, int li, int lh) {
  const int parts = P.tiles_n * WN_;
  const int part = (g.n0 / (NT * 32 * WN_)) * WN_ + wn;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float best = -3.4e38f;
    int bcol = 0x7fffffff;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = g.n0 + cout_base + nt * 32 + 8 * q + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + (c0 < P.cout ? c0 : 0));
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[mt][nt][4 * q + j] + (P.bias ? bb[j] : 0.f);
          const bool take = (c0 + j < P.cout) && v > best;
          best = take ? v : best; bcol = take ? c0 + j : bcol;
        }
      }
    // join the two half-waves (columns 8q + 4 lh ...: the partner holds the other 4 of every 8)
    const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto sc = __builtin_amdgcn_permlane32_swap((unsigned)bcol, (unsigned)bcol, false, false);
    const float v0 = __uint_as_float(sv[0]), v1 = __uint_as_float(sv[1]);
    const int c0 = (int)sc[0], c1 = (int)sc[1];
    const bool second = v1 > v0 || (v1 == v0 && c1 < c0);
    const float bv = second ? v1 : v0;
    const int bc = second ? c1 : c0;
    const int row = g.ox0 + pix_base + mt * 32 + li;
    if (lh == 0 && row < P.oh * P.ow) {
      float2* wsp = reinterpret_cast<float2*>(P.rowmax) + ((long long)g.img * (P.oh * P.ow) + row) * parts + part;
      *wsp = make_float2(bv, __int_as_float(bc));
    }
  }
}

}  // namespace gpemsr
