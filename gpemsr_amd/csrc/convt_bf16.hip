// Transposed convolutions of the bf16 data path with the weights resident in LDS (dispatched by gpemsr_conv2d_bf16 / plan_x in
// conv_bf16.hip; replaces, for bf16 tensors, nn.ConvTranspose2d(64, 64, 3, 2, 1, 1) of R:model/GPEMSR.py:252-253,340,380-382).
#include "conv_bf16.h"

namespace gpemsr {

// ---------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(k3,s2,p1,op1) with 64 input channels and the WEIGHTS RESIDENT IN LDS (reffea_L{2,3,4}_conv1, R:model/GPEMSR.py:252-253,
// 380-382,340: 64 -> 64 at up to 512 x 512 output).  On the ring kernel above these layers ran at 0.3 PFLOP/s: a stage moved 51 KB by
// LDS-DMA (19 KB halo image + 32 KB of phase-stacked weights, 7 of 16 (tap, phase) blocks zero) for 18 MFMAs of a wave, two stages per
// tile.  The layer is bound by its OUTPUT (4 x the input: 115 FLOP per HBM byte = 0.58 PFLOP/s at 5 TB/s), so the kernel is built
// around streaming: the nine non-zero (tap, phase) weight blocks of a 64-cout slab (73.7 KB, the size of a 3x3 slab) stay in LDS, four
// loader waves keep the halo images of the NEXT 8 x 32-pixel input tile in flight (both 32-channel chunks, 2 x 19 KB, double buffered
// by tile), and each of eight multiplying waves owns one input row x 64 couts and walks the four output phases one after the other:
// phase q = 2 py + px contracts its 1 / 2 / 2 / 4 taps x 64 channels into two accumulator tiles (32 registers) and stores them --
// 16-byte pieces, every output pixel one 128-byte line -- before the next phase starts.  One workgroup barrier per tile; inside a
// tile the waves drift freely (nothing they read changes), so one wave's stores overlap the other waves' MFMAs.
//   out(2i+py, 2j+px) = b + sum_{dy<=py, dx<=px} in(i+dy, j+dx) . W[:, :, py+1-2dy, px+1-2dx]            (gpemsr_amd/packing.py::pack_convT)
// Weight slab in global memory (appended to the staged form by pack_convT_bf16, descriptor.weight_forms bit 0):
//   [cout/64][chunk 2][block 9][piece 4][64 couts][8], blocks in phase order: q0:(0,0) | q1:(0,0),(0,1) | q2:(0,0),(1,0) | q3:(0,0),(0,1),(1,0),(1,1)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768, 3) void convt64_resident_kernel(XParams P) {
  constexpr int HALO_W = 33, HALO_H = 9, HALO_PX = HALO_W * HALO_H, R = 4;
  constexpr int A_BYTES = HALO_PX * R * 16;            // 19,008 per chunk image
  constexpr int W_BYTES = 2 * 9 * 4 * 64 * 16;         // 73,728
  constexpr int NA = (HALO_PX * R + 255) / 256;        // 5 slots per loader thread per chunk image
  constexpr int NW = W_BYTES / 16 / 256;               // 18 slots per loader thread for the weight slab
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  float* const bias_lds = reinterpret_cast<float*>(xsm + W_BYTES + 4 * A_BYTES);
  const unsigned xsm_lds = xlds_addr(xsm);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bid = (int)blockIdx.x;                           // XCD-aware order: logically consecutive workgroups (neighbouring tiles) share an L2
  if ((gridDim.x & 7u) == 0u) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int tn = bid % P.tiles_n, sg = bid / P.tiles_n;
  const int n0 = tn * 64;
  const int gpt = P.gpt, NS = P.ns;
  const int T_me = (NS - sg + gpt - 1) / gpt;          // 8 x 32 input tiles of this workgroup: sg, sg + gpt, ...

  x_stage_bias(P, bias_lds, P.nbias, 768);

  auto tile_geo = [&](int j) -> XGeo {
    int t = sg + j * gpt, tx, ty;
    xdivmod(t, P.tiles_x, P.mg_x, t, tx);
    xdivmod(t, P.tiles_y, P.mg_y, t, ty);
    XGeo g;
    g.img = t; g.n0 = n0; g.oy0 = ty * 8; g.ox0 = tx * 32; g.tile_in_img = ty * P.tiles_x + tx;
    return g;
  };
  auto tile_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 8) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    const int dtid = tid - 512, dwave = wave - 8;
    const unsigned lds0 = xuni(xsm_lds + (unsigned)dwave * 1024u);
    const unsigned pixb = (unsigned)P.ld[0] * 2u;
    {   // this workgroup's slab: 72 consecutive KiB
      const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.weight + (long long)tn * (W_BYTES / 2)));
#pragma unroll
      for (int i = 0; i < NW; ++i) xglds16((unsigned)(dtid + i * 256) * 16u, wp, lds0 + i * 4096u);
    }
    auto issue_tile = [&](int j) {                     // both chunk images of tile j -> buffers (j & 1, 0 / 1)
      const XGeo t = tile_geo(j);
      int a_pix[NA];
      bool pad = false;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = dtid + i * 256;
        int pix = -1;
        if (e < HALO_PX * R) {
          const int hp = e / R;
          const int iy = t.oy0 + hp / HALO_W, ix = t.ox0 + hp % HALO_W;      // taps reach down / right only (dy, dx in {0, 1})
          if (iy < P.h && ix < P.w) pix = iy * P.w + ix;
          pad = pad || pix < 0;
        }
        a_pix[i] = pix;
      }
      const bool any_pad = __ballot(pad) != 0ull;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned short* sp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.src[0] + (long long)t.img * P.img_stride[0] + c * 32));
        const unsigned la = xuni(lds0 + (unsigned)(W_BYTES + (2 * (j & 1) + c) * A_BYTES));
        if (any_pad) {
          char* ab = xsm + W_BYTES + (2 * (j & 1) + c) * A_BYTES;
#pragma unroll
          for (int i = 0; i < NA; ++i)
            if (dtid + i * 256 < HALO_PX * R && a_pix[i] < 0) *reinterpret_cast<float4*>(ab + (dtid + i * 256) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
          if (a_pix[i] >= 0) {
            const int e = dtid + i * 256, hp = e / R;
            const unsigned q = (unsigned)((e % R) ^ ((hp >> 2) & 3));
            xglds16((unsigned)a_pix[i] * pixb + 16u * q, sp, la + i * 4096u);
          }
      }
    };
    issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile_barrier();                                    // weights + tile 0 have landed
    for (int j = 0; j < T_me; ++j) {
      // buffer (j + 1) & 1 was read by tile j - 1, which every multiplying wave left at the previous barrier
      if (j + 1 < T_me) issue_tile(j + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tile_barrier();
    }
    return;
  }

  // ------------------------------------------------ multiplying waves: input row `wave` of the tile, all 64 couts ------------------------------------------------
  unsigned aoff[2][2];                                 // fragment byte offsets of tap (dy, dx), k-step 0 (k-step 1 flips bit 5); tile-invariant
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int hp = (wave + dy) * HALO_W + li + dx;
      aoff[dy][dx] = xsm_lds + (unsigned)(W_BYTES + hp * 64 + ((lh ^ ((hp >> 2) & 3)) * 16));
    }
  const unsigned wfrag = xsm_lds + (unsigned)(li * 16 + lh * 1024);
  const bool lrelu = P.act == GPEMSR_ACT_LRELU, relu = P.act == GPEMSR_ACT_RELU;

  tile_barrier();
  for (int j = 0; j < T_me; ++j) {
    const XGeo g = tile_geo(j);
    const unsigned abuf = (unsigned)((j & 1) * 2 * A_BYTES);
    const int oy = g.oy0 + wave, ox = g.ox0 + li;
    const bool pok = oy < P.oh && ox < P.ow;
    unsigned short* const orow = reinterpret_cast<unsigned short*>(P.out) +
        (((long long)g.img * P.OH + 2 * oy) * P.OW + 2 * ox) * P.out_ld + n0 + 8 * lh;
    // one output phase after the other: BLK0 = first weight block of the phase, NTAP taps
    auto phase = [&](auto QC) {
      constexpr int Q = decltype(QC)::value;
      constexpr int BLK0 = Q == 0 ? 0 : (Q == 1 ? 1 : (Q == 2 ? 3 : 5));
      constexpr int NTAP = Q == 0 ? 1 : (Q == 3 ? 4 : 2);
      f32x16 acc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + n0 + nt * 32 + 8 * q4 + 4 * lh);
          acc[nt][4 * q4] = b4.x; acc[nt][4 * q4 + 1] = b4.y; acc[nt][4 * q4 + 2] = b4.z; acc[nt][4 * q4 + 3] = b4.w;
        }
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        // taps of phase (py, px) in block order: q1: (0,0),(0,1)  q2: (0,0),(1,0)  q3: (0,0),(0,1),(1,0),(1,1)
        constexpr int dyq[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 1}};
        constexpr int dxq[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 0, 0}, {0, 1, 0, 1}};
        const unsigned ao = aoff[dyq[Q][t]][dxq[Q][t]] + abuf;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 fa = xlds_read16((ks ? (ao ^ 32u) : ao) + (unsigned)(c * A_BYTES));
            const unsigned wb = wfrag + (unsigned)(((c * 9 + BLK0 + t) * 4 + 2 * ks) * 1024);
            const bf16x8 f0 = xlds_read16(wb), f1 = xlds_read16(wb + 512u);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, fa, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fa, acc[1], 0, 0, 0);
          }
      }
      unsigned short* const op = orow + ((long long)(Q >> 1) * P.OW + (Q & 1)) * P.out_ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        unsigned pk[4][2];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = acc[nt][4 * q4 + k];
            if (lrelu) v[k] = fmaxf(v[k], 0.1f * v[k]);
          }
          pk[q4][0] = xcvt_pk_bf16(v[0], v[1]);
          pk[q4][1] = xcvt_pk_bf16(v[2], v[3]);
        }
        if (relu) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int k = 0; k < 2; ++k) asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[q4][k]) : "v"(pk[q4][k]));
        }
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 1][1], false, false);
          if (pok) *reinterpret_cast<uint4*>(op + nt * 32 + 8 * gp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    tile_barrier();
  }
}

}  // namespace gpemsr

using namespace gpemsr;

int gpemsr::launch_convt64_resident(const XParams& Pin, size_t lds, hipStream_t st) {
  XParams P = Pin;
  static dev_once_t tattr{0};
  if (dev_once_begin(tattr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(convt64_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d_bf16: cannot raise the dynamic LDS limit");
    dev_once_done(tattr);
  }
  // one workgroup per CU; the workgroups of a cout slab split the spatial tiles between them
  const int cus = device_cus();
  int gpt = cus / P.tiles_n;
  if (gpt < 1) gpt = 1;
  if (gpt > P.ns) gpt = P.ns;
  P.gpt = gpt;
  hipLaunchKernelGGL(convt64_resident_kernel, dim3(gpt * P.tiles_n), dim3(768), lds, st, P);
  return check_launch("convt64_resident_kernel");
}
