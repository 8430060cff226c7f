// Fused mask front end of the prior fusion (R:model/GPEMSR.py:385-395):
//     a = relu1_2(vgg(ref_img x3)),  b = relu1_2(vgg(bilinear_s(LR) x3)),
//     mask[p] = <a_p, b_p> / (max(|a_p|, 1e-12) max(|b_p|, 1e-12))        per 16x16x64 patch p
// in ONE kernel: neither the two [n,64,sH,sW] feature maps (1.3 GB each per 5 slices at 1024^2 in fp32; SURVEY section 2.3 says
// they must never reach HBM) nor the up-sampled LR image are ever materialised.  VGG relu1_2 is 13.7 % of the essential
// FLOPs of the forward and was 25 % of the bf16 step (stem 1->64 write + conv1_2 read/write + patch reduction read).
//
// Structure = conv64_resident_kernel (conv_bf16.hip): a 512-thread workgroup keeps conv1_2's 64x64x3x3 bf16 weights in LDS
// and walks 16x32-pixel tiles (= one patch row x two patches).  Per tile and per image (prior image, then up-sampled LR):
//   1. the 20x36 fp32 image window goes to LDS (the LR window is bilinearly resampled on the fly, torch's align_corners=False
//      rule); zero outside the image = conv1_1's padding;
//   2. conv1_1 (1 -> 64, the three identical input channels folded into the weights) on the matrix pipe: im2col K = 9 taps as
//      hi + lo bf16 halves of the fp32 pixel (18 of 32 k slots, so the INPUT keeps fp32 accuracy), D^T = W1 . im2col^T per
//      32 pixels x 32 couts, + bias, ReLU, -> bf16, written straight into the swizzled A images of conv1_2 (both 32-channel
//      chunks); halo pixels outside the image are conv1_2's zero padding;
//   3. conv1_2: 2 chunks x 9 taps x 2 k-steps x 4 MFMAs per wave from LDS only, accumulators D^T (lane = pixel);
//   4. after both images: relu(acc + bias) products reduced per patch (lanes of a 16-column half, registers, waves) in a fixed
//      order -> deterministic.
// No HBM traffic besides the two 1-channel images and 4 bytes per patch.
#include "bf16_common.h"
#include <stdlib.h>

namespace gpemsr {

#ifdef GPVGG_STAMP
// diagnostic build (scripts/attic/vgg_stamp_probe.py): cycles per bucket of every wave of the first 256 workgroups, summed over the workgroup's life
__device__ unsigned long long g_vstamps[256 * 16 * 8];
#define VSEG_DECL unsigned long long vs_t = __builtin_amdgcn_s_memtime(), vs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define VSEG(k) do { const unsigned long long vs_n = __builtin_amdgcn_s_memtime(); vs_acc[k] += vs_n - vs_t; vs_t = vs_n; } while (0)
#define VSEG_FLUSH do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) { for (int k = 0; k < 8; ++k) g_vstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + k] = vs_acc[k]; } } while (0)
#else
#define VSEG_DECL
#define VSEG(k)
#define VSEG_FLUSH
#endif

// timing hooks exist only in a probe build (-DGPEMSR_VGG_PROBE; either of them makes the masks wrong on purpose)
#ifdef GPEMSR_VGG_PROBE
#define VGG_DBG(P) ((P).dbg)
#else
#define VGG_DBG(P) 0
#endif

struct VggParams {
  const float* ref; const float* lr;       // [n][H][W] prior image, [n][h][w] LR slice
  int n, H, W, h, w;
  float sh, sw;                              // h / H, w / W as torch computes them
  const float* w1; const float* b1;          // conv1_1: [64][9] (cin-summed), [64]
  const unsigned short* w2; const float* b2; // conv1_2: staged bf16 [2][9][4][64][8], [64]
  float* out;                                // [n][H/16][W/16]
  int tiles_x, tiles_y, ns;
  unsigned mg_x, mg_y;                       // floor((2^32 - 1) / tiles_*) for xdivmod
  int dbg;                                   // timing experiments only, compiled in with -DGPEMSR_VGG_PROBE (GPEMSR_VGG_DBG): 1 = producers idle, 2 = no conv1_2
  int hr2;                                   // `lr` is already at the HR size (scale == 1): both images are read the same way
};

__device__ __forceinline__ void vsrc_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);            // aten area_pixel_compute_source_index, align_corners=False
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - i0;
}

__global__ __launch_bounds__(512, 2) void vgg_mask_kernel(VggParams P) {
  constexpr int HALO_W = 34, HALO_H = 18, HALO_PX = HALO_W * HALO_H;      // conv1_2 input window of a 16x32 tile
  constexpr int PW = 36, PH = 20;                                        // conv1_1 input window
  constexpr int A_BYTES = HALO_PX * 64;                                   // one 32-channel chunk image, 64-B pixel rows
  constexpr int W_BYTES = 2 * 9 * 4 * 64 * 16;
  constexpr int NW = W_BYTES / 16 / 512;
  extern __shared__ __attribute__((aligned(16))) char vsm[];
  char* const w_base = vsm;
  char* const a_base = vsm + W_BYTES;                                     // chunk 0, chunk 1
  float* const patch = reinterpret_cast<float*>(vsm + W_BYTES + 2 * A_BYTES);          // [PH][PW]
  float* const cst = patch + PH * PW;                                     // b1[64], b2[64]
  float* const red = cst + 128;                                           // [8 waves][2 patches][3]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)vsm + (unsigned)wave * 1024u);

  // ---- once per workgroup: conv1_2 weights -> LDS, biases -> LDS, conv1_1 weights -> MFMA row fragments in registers ----
  {
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.w2));
#pragma unroll
    for (int i = 0; i < NW; ++i) xglds16((unsigned)(tid + i * 512) * 16u, wp, lds0 + i * 8192u);
  }
  if (tid < 64) { cst[tid] = P.b1[tid]; cst[64 + tid] = P.b2[tid]; }
  // W1 fragments: row operand of D^T = W1 . im2col^T.  k slots: 0..8 = taps (for the hi half of the pixel), 9..17 = the same
  // taps again (lo half), 18..31 = 0.  Lane (li = cout within the 32-row tile ct, lh) holds k = 16 s + 8 lh + j.
  bf16x8 w1f[2][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * lh + j;
        const int tap = k < 9 ? k : k - 9;
        const float v = k < 18 ? P.w1[(ct * 32 + li) * 9 + tap] : 0.f;
        w1f[ct][s][j] = (short)(xcvt_pk_bf16(v, 0.f) & 0xFFFFu);
      }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // tile-invariant fragment offsets, formed once (see conv64_resident_kernel): k-step 1 = offset ^ 32
  int aoff[2][9];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hp = (2 * wave + mt) * HALO_W + li + (tap / 3) * HALO_W + tap % 3;
      aoff[mt][tap] = hp * 64 + ((lh ^ ((hp >> 2) & 3)) * 16);
    }
  const int b_frag = li * 16 + lh * 1024;

  for (int t = (int)blockIdx.x; t < P.ns; t += (int)gridDim.x) {
    int q = t;
    const int tx = q % P.tiles_x; q /= P.tiles_x;
    const int ty = q % P.tiles_y; q /= P.tiles_y;
    const int img = q;
    const int oy0 = ty * 16, ox0 = tx * 32;
    f32x16 acc[2][2];                 // [pixel tile][cout tile] of the image being convolved
    unsigned pa[2][2][8];             // relu1_2 features of the prior image, packed bf16 pairs (registers 2k, 2k+1)
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
      // ---- 1. image window -> LDS (zero outside the image) ----
      for (int e = tid; e < PH * PW; e += 512) {
        const int py = e / PW, px = e % PW;
        const int Y = oy0 - 2 + py, X = ox0 - 2 + px;
        float v = 0.f;
        if (Y >= 0 && Y < P.H && X >= 0 && X < P.W) {
          if (which == 0) v = P.ref[((long long)img * P.H + Y) * P.W + X];
          else {
            int y0, y1, x0, x1; float ly, lx;
            vsrc_index(Y, P.sh, P.h, y0, y1, ly);
            vsrc_index(X, P.sw, P.w, x0, x1, lx);
            const float* b = P.lr + (long long)img * P.h * P.w;
            const float hy = 1.f - ly, hx = 1.f - lx;
            v = hy * (hx * b[y0 * P.w + x0] + lx * b[y0 * P.w + x1]) + ly * (hx * b[y1 * P.w + x0] + lx * b[y1 * P.w + x1]);
          }
        }
        patch[e] = v;
      }
      __syncthreads();                 // also: everybody is done reading the A images of the previous phase
      // ---- 2. conv1_1 on the matrix pipe -> A images of conv1_2 ----
      for (int grp = wave; grp < (HALO_PX + 31) / 32; grp += 8) {
        const int hp = grp * 32 + li;
        const int hy = hp / HALO_W, hx = hp % HALO_W;           // halo pixel; its 3x3 window starts at patch (hy, hx)
        const bool exists = hp < HALO_PX;
        float pv[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) pv[tp] = exists ? patch[(hy + tp / 3) * PW + hx + tp % 3] : 0.f;
        unsigned hi[9], lo[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          hi[tp] = xcvt_pk_bf16(pv[tp], 0.f) & 0xFFFFu;
          lo[tp] = xcvt_pk_bf16(pv[tp] - xbf_lo(hi[tp]), 0.f) & 0xFFFFu;
        }
        // column operand: k = 16 s + 8 lh + j  ->  s = 0: lh = 0: hi0..hi7 | lh = 1: hi8, lo0..lo6 ;  s = 1: lh = 0: lo7, lo8, 0.. | lh = 1: 0
        bf16x8 f0, f1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned v0 = lh == 0 ? hi[j] : (j == 0 ? hi[8] : lo[j - 1]);
          const unsigned v1 = lh == 0 ? (j == 0 ? lo[7] : (j == 1 ? lo[8] : 0u)) : 0u;
          f0[j] = (short)v0; f1[j] = (short)v1;
        }
        const int Y = oy0 - 1 + hy, X = ox0 - 1 + hx;
        const bool inside = exists && Y >= 0 && Y < P.H && X >= 0 && X < P.W;        // else: conv1_2's zero padding
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ct][0], f0, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ct][1], f1, d, 0, 0, 0);
          if (exists) {
            char* arow = a_base + ct * A_BYTES + hp * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {                       // registers 4g..4g+3 = couts 32 ct + 8 g + 4 lh + (0..3)
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = inside ? fmaxf(d[4 * g + j] + cst[ct * 32 + 8 * g + 4 * lh + j], 0.f) : 0.f;
              *reinterpret_cast<uint2*>(arow + ((g ^ ((hp >> 2) & 3)) * 16) + lh * 8) = make_uint2(xcvt_pk_bf16(v[0], v[1]), xcvt_pk_bf16(v[2], v[3]));
            }
          }
        }
      }
      __syncthreads();
      // ---- 3. conv1_2 from LDS ----
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
#pragma unroll 1
      for (int chunk = 0; chunk < 2; ++chunk) {
        const char* A = a_base + chunk * A_BYTES;
        const char* B = w_base + chunk * (9 * 4 * 1024) + b_frag;
        bf16x8 fa[2][2], fb[2][2];
        auto load_step = [&](int set, int st) {
          const int tap = st >> 1, ks = st & 1;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) fa[set][mt] = *reinterpret_cast<const bf16x8*>(A + (ks ? (aoff[mt][tap] ^ 32) : aoff[mt][tap]));
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) fb[set][nt] = *reinterpret_cast<const bf16x8*>(B + (tap * 4 + 2 * ks) * 1024 + nt * 512);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
          if (st + 1 < 18) load_step((st + 1) & 1, st + 1);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st & 1][nt], fa[st & 1][mt], acc[mt][nt], 0, 0, 0);
        }
      }
      if (which == 0) {                // keep a = relu(conv1_2 + bias) as bf16 (what the layer-by-layer path stores in HBM)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float b0 = cst[64 + nt * 32 + ((2 * k) & 3) + 8 * ((2 * k) >> 2) + 4 * lh];
              const float b1v = cst[64 + nt * 32 + ((2 * k + 1) & 3) + 8 * ((2 * k + 1) >> 2) + 4 * lh];
              pa[mt][nt][k] = xcvt_pk_bf16(fmaxf(acc[mt][nt][2 * k] + b0, 0.f), fmaxf(acc[mt][nt][2 * k + 1] + b1v, 0.f));
            }
      }
    }
    // ---- 4. patch cosine: lane = pixel (column li -> patch li >> 4), registers = couts ----
    // (pixels beyond the image -- ragged last tile -- carry relu(bias) of zero features: masked out)
    float dot = 0.f, na = 0.f, nb = 0.f;
    {
      const bool colok = ox0 + li < P.W;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bias = cst[64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const float okf = (colok && oy0 + 2 * wave + mt < P.H) ? 1.f : 0.f;
            const float a = okf * ((r & 1) ? xbf_hi(pa[mt][nt][r >> 1]) : xbf_lo(pa[mt][nt][r >> 1]));
            const float b = okf * fmaxf(acc[mt][nt][r] + bias, 0.f);
            dot = fmaf(a, b, dot); na = fmaf(a, a, na); nb = fmaf(b, b, nb);
          }
        }
    }
#pragma unroll
    for (int o = 1; o <= 8; o <<= 1) { dot += __shfl_xor(dot, o); na += __shfl_xor(na, o); nb += __shfl_xor(nb, o); }
    dot += __shfl_xor(dot, 32); na += __shfl_xor(na, 32); nb += __shfl_xor(nb, 32);
    if (lane == 0 || lane == 16) {
      float* rp = red + (wave * 2 + (lane >> 4)) * 3;
      rp[0] = dot; rp[1] = na; rp[2] = nb;
    }
    __syncthreads();
    if (tid < 2) {
      float d = 0.f, x = 0.f, y = 0.f;
      for (int wv = 0; wv < 8; ++wv) { d += red[(wv * 2 + tid) * 3]; x += red[(wv * 2 + tid) * 3 + 1]; y += red[(wv * 2 + tid) * 3 + 2]; }
      const int pxx = tx * 2 + tid;
      if (pxx < P.W / 16)
        P.out[((long long)img * (P.H / 16) + ty) * (P.W / 16) + pxx] = d / (fmaxf(sqrtf(x), 1e-12f) * fmaxf(sqrtf(y), 1e-12f));   // F.normalize eps
    }
    // (the next tile's first __syncthreads orders the reuse of `red`, `patch` and the A images)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same computation with its phases OVERLAPPED (the kernel above runs window load -> conv1_1 -> conv1_2 -> reduction one after
// the other on all 8 waves: the matrix pipe is busy 31 % of the time).  768 threads = 8 multiplying waves + 4 producer waves, a
// stream of work items (8 x 32-pixel half-tile, image) and two A-image buffers:
//   interval i:  multiplying waves: conv1_2 of item i from buffer i & 1 (wave w = pixel row w, 32 pixels x 64 couts, 72 MFMAs);
//                producer waves: conv1_1 of item i + 1 into buffer (i + 1) & 1 -- each lane fetches the 3x3 window of its halo
//                pixel straight from the 1-channel image (the LR image is resampled on the fly), no window staging in LDS;
//   one workgroup-wide barrier per interval.  A 16 x 32 super-tile = 4 items (upper half: prior image, up-sampled LR; lower half:
//   the same); the per-lane patch sums live in registers across its two halves, the cross-wave reduction of super-tile k is
//   finished in the interval after its last item.
// LDS: W2 73,728 + 4 x 21,760 (two buffers x two 32-channel chunks of a 10 x 34 halo) + 704 bytes of constants / partial sums.
template <int NPROD>
__global__ __launch_bounds__((8 + NPROD) * 64, NPROD == 8 ? 4 : 3) void vgg_mask2_kernel(VggParams P) {
  constexpr int HALO_W = 34, HALO_H = 10, HALO_PX = HALO_W * HALO_H;      // conv1_2 input window of an 8 x 32 half-tile
  constexpr int A_BYTES = HALO_PX * 64;                                   // 21,760
  constexpr int W_BYTES = 2 * 9 * 4 * 64 * 16;
  constexpr int NW = W_BYTES / 16 / 768;                                  // 6
  constexpr int NGRP = (HALO_PX + 31) / 32;                               // 11 groups of 32 halo pixels
  extern __shared__ __attribute__((aligned(16))) char vsm[];
  float* const cst = reinterpret_cast<float*>(vsm + W_BYTES + 4 * A_BYTES);   // b1[64], b2[64]
  float* const red = cst + 128;                                               // [8 waves][2 patches][3]
  const unsigned vsm_lds = xlds_addr(vsm);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  {
    const unsigned lds0 = xuni(vsm_lds + (unsigned)wave * 1024u);
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(xuni_ptr(P.w2));
    if (tid < 768) {
#pragma unroll
      for (int i = 0; i < NW; ++i) xglds16((unsigned)(tid + i * 768) * 16u, wp, lds0 + i * 12288u);
    }
  }
  if (tid < 64) { cst[tid] = P.b1[tid]; cst[64 + tid] = P.b2[tid]; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();           // the biases are read from LDS by both roles (registers are scarce in the 16-wave form)
  asm volatile("" ::: "memory");
  const int T_me = (P.ns - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // 16 x 32 super-tiles of this workgroup
  const int NWI = 4 * T_me;
  auto item_geo = [&](int wi, int& img, int& oy0, int& ox0, int& tx, int& ty) {
    int q = (int)blockIdx.x + (wi >> 2) * (int)gridDim.x;
    xdivmod(q, P.tiles_x, P.mg_x, q, tx);
    xdivmod(q, P.tiles_y, P.mg_y, q, ty);
    img = q; oy0 = ty * 16 + ((wi >> 1) & 1) * 8; ox0 = tx * 32;
  };
  auto end_interval = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= 8) {
    // ------------------------------------------------ producer waves: conv1_1 ------------------------------------------------
    const int pw = wave - 8;
    VSEG_DECL;
    // Measured and NOT kept (build switches for A/B: -DVGG_PRIO=n, -DVGG_BAL=1; kernel alone, 80 slices of 128^2 x8, ms): the producers'
    // few MFMAs lose the arbitration for the matrix pipe to the older multiplying waves (in-kernel stamps: every producer wave sat ~5k
    // cycles in its conversion phase), but raising their priority only moves the wait to the multiplying waves (12.65 vs 12.35), and
    // dealing the 22 (group, cout tile) units evenly (three instead of four on the longest wave) does not pay either (12.76): the
    // interval is set by the SIMDs' total issue + LDS load, not by one role.
#ifndef VGG_PRIO
#define VGG_PRIO 0
#endif
    if (VGG_PRIO) __builtin_amdgcn_s_setprio(VGG_PRIO);
    // W1 fragments: row operand of D^T = W1 . im2col^T.  k slots: 0..8 = taps (hi half of the pixel), 9..17 = the same taps (lo
    // half), 18..31 = 0.  Lane (li = cout within the 32-row tile ct, lh) holds k = 16 s + 8 lh + j.
    bf16x8 w1f[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * lh + j;
          const int tap = k < 9 ? k : k - 9;
          // k slots 18 / 19: conv1_1's bias as a bf16 pair (hi + lo) against a constant 1 in the pixel operand -- the matrix pipe adds it
          const float bv = P.b1[ct * 32 + li];
          const float bh = xbf_lo(xcvt_pk_bf16(bv, 0.f) & 0xFFFFu);
          const float v = k < 18 ? P.w1[(ct * 32 + li) * 9 + tap] : (k == 18 ? bh : (k == 19 ? bv - bh : 0.f));
          w1f[ct][s][j] = (short)(xcvt_pk_bf16(v, 0.f) & 0xFFFFu);
        }
    // One producer wave owns halo-pixel groups pw, pw + NPROD (11 groups of 32 over NPROD waves: one or two per item).  The 3x3 windows of
    // ALL its groups are fetched before the first is converted: with one group after the other (round 2) the second group's loads were
    // issued only after the first group's arithmetic, i.e. two dependent memory round trips (~2 us each under load) per item -- the
    // producers alone took 11.6 ms of the kernel's 13.9.
    constexpr int MAXG = (NGRP + NPROD - 1) / NPROD;
    // FETCH of item i + 2 is issued before the barrier that ends interval i, its values are consumed (masked, converted) in interval
    // i + 1: the memory round trip (3-4k cycles under load, in-kernel stamps) lies under a whole interval instead of in front of the
    // conversion.  `raw` / `rmask` are live across the barrier (20 registers).
    float raw[MAXG][9];
    unsigned rmask[MAXG];
    // Work split: groups pw, pw + NPROD (VGG_BAL = 1: group pw whole + one cout tile of groups 8-10 on waves 0-5; see above)
#ifndef VGG_BAL
#define VGG_BAL 0
#endif
    auto slot_group = [&](int gi) -> int {
      if (NPROD != 8 || !VGG_BAL) return pw + gi * NPROD;
      return gi == 0 ? pw : (pw < 6 ? NPROD + (pw >> 1) : NGRP);
    };
    auto slot_does = [&](int gi, int ct) -> bool { return NPROD != 8 || !VGG_BAL || gi == 0 || ct == (pw & 1); };
    auto fetch = [&](int wi) {
      int img, oy0, ox0, tx, ty;
      item_geo(wi, img, oy0, ox0, tx, ty);
      const int which = wi & 1;
      const float* rimg = P.ref + (long long)img * P.H * P.W;
      const float* limg = P.lr + (long long)img * P.h * P.w;
      VSEG(3);
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi) {
        const int grp = slot_group(gi);
        const int hp = grp * 32 + li;
        const bool exists = grp < NGRP && hp < HALO_PX;
        const int hpc = exists ? hp : 0;
        const int hy = hpc / HALO_W, hx = hpc - hy * HALO_W;
        const int Y = oy0 - 1 + hy, X = ox0 - 1 + hx;                     // this halo pixel; its 3x3 window is centred on it
        bool rin[3], cin[3];
        int yc[3], xc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int yy = Y - 1 + d, xx = X - 1 + d;
          rin[d] = yy >= 0 && yy < P.H; cin[d] = xx >= 0 && xx < P.W;
          yc[d] = yy < 0 ? 0 : (yy > P.H - 1 ? P.H - 1 : yy);
          xc[d] = xx < 0 ? 0 : (xx > P.W - 1 ? P.W - 1 : xx);
        }
        unsigned m = 0u;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) m |= ((exists && rin[tp / 3] && cin[tp % 3]) ? 1u : 0u) << tp;      // zero outside the image = conv1_1's padding
        rmask[gi] = m;
        if (which == 0 || P.hr2) {
          // (the second image already up-sampled by gpemsr_bilinear, `scale` == 1: the on-the-fly resampling below -- 36 loads and
          // ~250 vector operations per halo pixel group -- kept the producer waves busy for 10k cycles per LR item, in-kernel stamps:
          // scripts/attic/vgg_stamp_probe.py; the up-sampled image costs 0.34 GB of traffic per step)
          const float* im = which == 0 ? rimg : P.lr + (long long)img * P.H * P.W;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) raw[gi][tp] = im[(long long)yc[tp / 3] * P.W + xc[tp % 3]];      // clamped address, masked at use
        } else {
          int y0[3], y1[3], x0[3], x1[3]; float ly[3], lx[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) { vsrc_index(yc[d], P.sh, P.h, y0[d], y1[d], ly[d]); vsrc_index(xc[d], P.sw, P.w, x0[d], x1[d], lx[d]); }
          float q[6][6];                  // LR pixels at rows (y0, y1) x cols (x0, x1) of the three window rows / columns (L1 / L2 resident)
#pragma unroll
          for (int a2 = 0; a2 < 6; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2)
              q[a2][b2] = limg[((a2 & 1) ? y1[a2 >> 1] : y0[a2 >> 1]) * P.w + ((b2 & 1) ? x1[b2 >> 1] : x0[b2 >> 1])];
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int r = tp / 3, c = tp % 3;
            const float hy2 = 1.f - ly[r], hx2 = 1.f - lx[c];
            raw[gi][tp] = hy2 * (hx2 * q[2 * r][2 * c] + lx[c] * q[2 * r][2 * c + 1]) + ly[r] * (hx2 * q[2 * r + 1][2 * c] + lx[c] * q[2 * r + 1][2 * c + 1]);
          }
        }
      }
      VSEG(0);
    };
    auto convert = [&](int wi) {
      int img, oy0, ox0, tx, ty;
      item_geo(wi, img, oy0, ox0, tx, ty);
      char* const dst = vsm + W_BYTES + (2 * (wi & 1)) * A_BYTES;          // buffer = item parity
      float pvs[MAXG][9];
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) pvs[gi][tp] = ((rmask[gi] >> tp) & 1u) ? raw[gi][tp] : 0.f;
#ifdef GPVGG_STAMP
      VSEG(4);
#endif
      // ---- phase 2: conv1_1 on the matrix pipe, + bias, ReLU, -> bf16 rows of conv1_2's A images ----
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi) {
        const int grp = slot_group(gi);
        if (grp >= NGRP) break;                                             // (wave-uniform)
        const int hp = grp * 32 + li;
        const bool exists = hp < HALO_PX;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int Y = oy0 - 1 + hy, X = ox0 - 1 + hx;
        unsigned hi[9], lo[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          hi[tp] = xcvt_pk_bf16(pvs[gi][tp], 0.f) & 0xFFFFu;
          lo[tp] = xcvt_pk_bf16(pvs[gi][tp] - xbf_lo(hi[tp]), 0.f) & 0xFFFFu;
        }
        bf16x8 f0, f1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned v0 = lh == 0 ? hi[j] : (j == 0 ? hi[8] : lo[j - 1]);
          const unsigned v1 = lh == 0 ? (j == 0 ? lo[7] : (j == 1 ? lo[8] : (j < 4 ? 0x3F80u : 0u))) : 0u;     // slots 18, 19 = 1.0 (bias)
          f0[j] = (short)v0; f1[j] = (short)v1;
        }
        const bool inside = exists && Y >= 0 && Y < P.H && X >= 0 && X < P.W;        // else: conv1_2's zero padding
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (!slot_does(gi, ct)) continue;                                 // (wave-uniform: the second slot is one cout tile of a shared group)
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ct][0], f0, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ct][1], f1, d, 0, 0, 0);
          if (exists) {
            char* arow = dst + ct * A_BYTES + hp * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = inside ? fmaxf(d[4 * g + j], 0.f) : 0.f;
              *reinterpret_cast<uint2*>(arow + ((g ^ ((hp >> 2) & 3)) * 16) + lh * 8) = make_uint2(xcvt_pk_bf16(v[0], v[1]), xcvt_pk_bf16(v[2], v[3]));
            }
          }
        }
      }
    };
    if (NWI > 0) { fetch(0); convert(0); }
    if (NWI > 1) fetch(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    end_interval();                                                       // weights, constants and item 0 are in LDS
    VSEG(3);
    for (int i = 0; i <= NWI; ++i) {
      if (i + 1 < NWI && !(VGG_DBG(P) == 1 && i >= 2)) {
        convert(i + 1);                                                   // windows fetched during the previous interval
        VSEG(1);
        if (i + 2 < NWI) fetch(i + 2);
      }
      end_interval();
      VSEG(2);
    }
    VSEG_FLUSH;
    return;
  }

  // ------------------------------------------------ multiplying waves: conv1_2 + patch sums ------------------------------------------------
  unsigned aoff[3][3];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int hp = (wave + rr) * HALO_W + li + kx;
      aoff[rr][kx] = (unsigned)(W_BYTES + hp * 64 + ((lh ^ ((hp >> 2) & 3)) * 16));
    }
  const unsigned b_frag = (unsigned)(li * 16 + lh * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  end_interval();
  unsigned pa[2][8];                      // relu1_2 features of the prior image at this lane's pixel, packed bf16 pairs
  float dot = 0.f, na = 0.f, nb = 0.f;
  VSEG_DECL;
  for (int i = 0; i <= NWI; ++i) {
    VSEG(3);
    if (i < NWI) {
      f32x16 acc[2];                      // start from conv1_2's bias (the epilogue below then has no adds and no LDS reads)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(cst + 64 + nt * 32 + 8 * q + 4 * lh);
          acc[nt][4 * q] = t.x; acc[nt][4 * q + 1] = t.y; acc[nt][4 * q + 2] = t.z; acc[nt][4 * q + 3] = t.w;
        }
      const unsigned bufo = (unsigned)((i & 1) * 2 * A_BYTES);
#pragma unroll 1
      for (int chunk = 0; chunk < (VGG_DBG(P) == 2 ? 0 : 2); ++chunk) {
        const unsigned A = vsm_lds + bufo + (unsigned)(chunk * A_BYTES);
        const unsigned B = vsm_lds + b_frag + (unsigned)(chunk * (9 * 4 * 1024));
        bf16x8 fa[2], fb[2][2];
        auto load_step = [&](int set, int st) {
          const int tap = st >> 1, ks = st & 1;
          const unsigned o = aoff[tap / 3][tap % 3];
          fa[set] = xlds_read16(A + (ks ? (o ^ 32u) : o));
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) fb[set][nt] = xlds_read16(B + (unsigned)((tap * 4 + 2 * ks) * 1024 + nt * 512));
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
          if (st + 1 < 18) load_step((st + 1) & 1, st + 1);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st & 1][nt], fa[st & 1], acc[nt], 0, 0, 0);
        }
      }
      VSEG(0);
      if ((i & 1) == 0) {                 // prior image: keep a = relu(conv1_2 + bias) as bf16 (what the layered path stores in HBM)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 8; ++k)
            pa[nt][k] = xcvt_pk_bf16(fmaxf(acc[nt][2 * k], 0.f), fmaxf(acc[nt][2 * k + 1], 0.f));
      } else {                            // up-sampled LR: products with the prior image's features, summed over the 64 couts
        int img, oy0, ox0, tx, ty;
        item_geo(i, img, oy0, ox0, tx, ty);
        const float okf = (ox0 + li < P.W && oy0 + wave < P.H) ? 1.f : 0.f;        // (pixels beyond a ragged last tile: masked out)
        float d1 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a = (r & 1) ? xbf_hi(pa[nt][r >> 1]) : xbf_lo(pa[nt][r >> 1]);
            const float b = fmaxf(acc[nt][r], 0.f);
            d1 = fmaf(a, b, d1); a1 = fmaf(a, a, a1); b1 = fmaf(b, b, b1);
          }
        dot = fmaf(okf, d1, dot); na = fmaf(okf, a1, na); nb = fmaf(okf, b1, nb);
        if ((i & 3) == 3) {               // both halves of the super-tile are in: per-wave patch sums -> LDS
#pragma unroll
          for (int o = 1; o <= 8; o <<= 1) { dot += __shfl_xor(dot, o); na += __shfl_xor(na, o); nb += __shfl_xor(nb, o); }
          dot += __shfl_xor(dot, 32); na += __shfl_xor(na, 32); nb += __shfl_xor(nb, 32);
          if (lane == 0 || lane == 16) {
            float* rp = red + (wave * 2 + (lane >> 4)) * 3;
            rp[0] = dot; rp[1] = na; rp[2] = nb;
          }
          dot = 0.f; na = 0.f; nb = 0.f;
        }
      }
    }
    VSEG(1);
    if (i >= 4 && (i & 3) == 0 && tid < 2) {          // finish the super-tile whose last item ran in the previous interval
      int img, oy0, ox0, tx, ty;
      item_geo(i - 4, img, oy0, ox0, tx, ty);
      float d = 0.f, x = 0.f, y = 0.f;
      for (int wv = 0; wv < 8; ++wv) { d += red[(wv * 2 + tid) * 3]; x += red[(wv * 2 + tid) * 3 + 1]; y += red[(wv * 2 + tid) * 3 + 2]; }
      const int pxx = tx * 2 + tid;
      if (pxx < P.W / 16)
        P.out[((long long)img * (P.H / 16) + ty) * (P.W / 16) + pxx] = d / (fmaxf(sqrtf(x), 1e-12f) * fmaxf(sqrtf(y), 1e-12f));   // F.normalize eps
    }
    end_interval();
    VSEG(2);
  }
  VSEG_FLUSH;
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_vgg_mask_bf16(const float* ref_img, const float* lr, int n, int h, int w, int scale, const float* w1, const float* b1,
                                    const void* w2_bf16, const float* b2, float* out, void* stream) {
  GP_REQUIRE(ref_img && lr && w1 && b1 && w2_bf16 && b2 && out && n > 0 && h > 0 && w > 0 && scale > 0, "vgg_mask_bf16: bad args");
  const int H = h * scale, W = w * scale;
  GP_REQUIRE(H % 16 == 0 && W % 16 == 0, "vgg_mask_bf16: the HR size must be a multiple of 16 (the reflect 'same' padding of "
             "model/GPEMSR.py:14-30 is not implemented; the forward asserts LR sizes that never need it)");
  GP_REQUIRE((reinterpret_cast<uintptr_t>(w2_bf16) & 15) == 0, "vgg_mask_bf16: weight alignment");
  VggParams P{};
  P.ref = ref_img; P.lr = lr; P.n = n; P.H = H; P.W = W; P.h = h; P.w = w;
  P.sh = (float)((double)h / (double)H); P.sw = (float)((double)w / (double)W);
  P.w1 = w1; P.b1 = b1; P.w2 = reinterpret_cast<const unsigned short*>(w2_bf16); P.b2 = b2; P.out = out;
  P.hr2 = scale == 1 ? 1 : 0;
  P.tiles_x = cdiv(W, 32); P.tiles_y = H / 16;
  P.mg_x = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.mg_y = 0xFFFFFFFFu / (unsigned)P.tiles_y;
  const long long ns = (long long)n * P.tiles_x * P.tiles_y;
  GP_REQUIRE(ns < (1ll << 31), "vgg_mask_bf16: grid too large");
  P.ns = (int)ns;
  const size_t lds = 73728 + 2 * (18 * 34 * 64) + (20 * 36 + 128 + 48) * 4;
  const size_t lds2 = 73728 + 4 * (10 * 34 * 64) + (128 + 48) * 4;
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(vgg_mask_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(vgg_mask2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(vgg_mask2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "vgg_mask_bf16: cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
#ifdef GPEMSR_VGG_PROBE
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GPEMSR_VGG_DBG"); dbg = e ? atoi(e) : 0; } P.dbg = dbg; }
#else
  P.dbg = 0;                            // the shipped library cannot idle the producers or skip conv1_2 (either makes the masks wrong on purpose)
#endif
  static int form = -1;                 // GPEMSR_VGG_FORM=1 selects the lockstep kernel (A/B measurements)
  if (form < 0) { const char* e = getenv("GPEMSR_VGG_FORM"); form = (e && e[0] == '1') ? 1 : ((e && e[0] == '2') ? 2 : 3); }      // 3: 8 producer waves
  const int cus = device_cus();
  const int grid = P.ns < cus ? P.ns : cus;
  if (form == 2) {
    hipLaunchKernelGGL(vgg_mask2_kernel<4>, dim3(grid), dim3(768), lds2, reinterpret_cast<hipStream_t>(stream), P);
    return check_launch("vgg_mask2_kernel");
  }
  if (form == 3) {
    hipLaunchKernelGGL(vgg_mask2_kernel<8>, dim3(grid), dim3(1024), lds2, reinterpret_cast<hipStream_t>(stream), P);
    return check_launch("vgg_mask2_kernel");
  }
  hipLaunchKernelGGL(vgg_mask_kernel, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("vgg_mask_kernel");
}

#ifdef GPVGG_STAMP
extern "C" int gpemsr_debug_read_vstamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gpemsr::g_vstamps), sizeof(unsigned long long) * 256 * 16 * 8);
}
#endif
