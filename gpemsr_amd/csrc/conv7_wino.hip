// 7x7 stride-1 convolution on the exact-fp32 matrix pipe in the 1-D Winograd form F(2, 7) along the image rows.
//
// SpyNet's 32 -> 64 and 64 -> 32 convolutions (basicsr spynet_arch.BasicModule, called from POD, R:model/GPEMSR.py:67,98-100) are 113 ms of
// the 630 ms fp32 step at 0.76-0.82 of the fp32 matrix peak in the direct form: nothing left to tune, only arithmetic to remove.  F(2, 7)
// computes two neighbouring outputs of a row from eight inputs with 8 multiplies per (cin, cout, filter row) instead of 14:
//     [y(2j), y(2j+1)] = A^T [ sum_ky (G g_ky) (.) (B^T d_{y+ky}) ]        d = columns 2j-3 .. 2j+4 of input row y + ky - 3
// with the interpolation points 0, +-1, +-2, +-1/2, infinity -- eight independent GEMMs  M_nu[pair][cout] = sum_(ky, cin) V_nu U_nu  on half
// of the pixels, K = 7 cin.  Executed MFMA FLOPs = 8/14 of the algorithmic ones.  All arithmetic is fp32 (U = G g folded on the host in
// float64); measured error ~3e-6 of the result (the direct fp32 kernel: 2e-6; the bar on the flows is 2e-3).
//
//   * workgroup = 8 waves, output tile 4 rows x 64 pixels (32 pairs = one MFMA row tile) x 32 NT couts; wave (yw = w & 3, vh = w >> 2)
//     owns output row yw and the positions nu = 4 vh .. 4 vh + 3: 4 NT accumulator tiles;
//   * per chunk of 8 input channels: (T) all threads transform the 10 raw halo rows into V[row][nu][quad][pair] in LDS (rows 8 and 9 cut
//     into position pairs so that every thread has work in both rounds) -- the transform of an
//     input row is shared by the 7 filter rows and all couts that use it (30 vector operations per 8 outputs and channel: rows 1/2, 3/4, 5/6
//     of B^T are even +- odd parts); (M) seven steps ky = 0..6: wave yw multiplies V[yw + ky] with the weight slab U[ky] -- one float4 of V
//     and one of U per four MFMAs;
//   * raw halo image [quad][column parity][row][column / 2] (conflict-free 16-byte reads for consecutive pairs), double buffered, the next
//     chunk's image arrives by LDS-DMA in six pieces spread over the seven steps (waves 0-3); weight slabs [nu][quad][cout][4] in a ring
//     (4 slabs of 8 KB, or 2 of 16 KB with 64 couts), one barrier per step, counted vmcnt;
//   * epilogue: A^T over the wave's own four positions in registers, the two position halves joined through LDS (same lane, same register:
//     no transposition), bias / activation, 128-byte stores (32 couts of one pixel).
//
// Replaces gpemsr_conv2d's direct form (descriptor.transposed = 4; weight = packing.pack_winograd7) for 7x7 stride-1 layers with one fp32
// source of c % 8 == 0 channels, cout % 32 == 0, no residual / multiplier, images at least 64 pixels wide.
#include "common.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct W7Params {
  const float* src; long long img_stride; int ld, cin;
  int n, h, w, cout;
  const float* weight;            // U [cin/8][7 ky][8 nu][2 quads][cout][4]
  const float* bias; int act;
  float* out; int out_ld;
  int tiles_x, tiles_y, tiles_n, nblocks;
};

constexpr int W7_TH = 4, W7_ROWS = W7_TH + 6, W7_C2 = 35;              // halo rows; columns per parity ((64 + 6) / 2)
constexpr int W7_RAW_SLOTS = 2 * 2 * W7_ROWS * W7_C2;                    // 1400 16-byte slots: [quad][parity][row][col / 2]
constexpr int W7_RAW_BYTES = W7_RAW_SLOTS * 16;                          // 22,400
constexpr int W7_V_BYTES = W7_ROWS * 8 * 2 * 32 * 16;                    // [row][nu][quad][pair]: 81,920
constexpr int W7_U_BYTES = 32768;                                        // slab ring
constexpr int W7_RAW_OFF = W7_V_BYTES, W7_U_OFF = W7_V_BYTES + 2 * W7_RAW_BYTES;
constexpr int W7_LDS = W7_U_OFF + W7_U_BYTES;                            // 159,488
constexpr int W7_PIECES = (W7_RAW_SLOTS + 255) / 256;                    // 6 raw pieces of 256 slots (issued by waves 0-3)

__device__ __forceinline__ void w7_glds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void w7_wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
  }
}

#define W7_F4(op, a, b) make_float4((a).x op (b).x, (a).y op (b).y, (a).z op (b).z, (a).w op (b).w)
__device__ __forceinline__ float4 w7_fma(float s, const float4& a, const float4& b) {      // s a + b
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 w7_mul(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

template <int NT>
__global__ __launch_bounds__(512, 1) void conv7_wino_f32_kernel(W7Params P) {
  constexpr int CO = 32 * NT;
  constexpr int SLAB = 8 * 2 * CO * 16;                    // bytes of one weight slab (one filter row of one chunk)
  constexpr int RING = W7_U_BYTES / SLAB;                  // 4 : 2
  constexpr int LA = RING - 1;                             // slabs issued ahead
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int yw = wave & 3, vh = wave >> 2;
  const bool issuer = wave < 4;                            // raw-image DMA (wave-uniform)

  int bid = blockIdx.x;
  {   // XCD-aware (bijective): consecutive logical blocks -- the cout blocks and neighbouring pixel tiles of one image -- share an L2
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = bid;
  const int tn = t % P.tiles_n; t /= P.tiles_n;
  const int tx0 = t % P.tiles_x; t /= P.tiles_x;
  const int ty0 = t % P.tiles_y; t /= P.tiles_y;
  const int img = t;
  const int oy0 = ty0 * W7_TH, ox0 = tx0 * 64, n0 = tn * CO;

  // ---- raw-image DMA slots of this thread (waves 0-3): piece p covers slots 256 p + tid; 2 * pixel + quad, or -1 outside the image ----
  int a_pk[W7_PIECES];
  int piece_live = 0;                                       // bit p: this WAVE issues an instruction for piece p (wave-uniform)
#pragma unroll
  for (int p = 0; p < W7_PIECES; ++p) {
    const int s = p * 256 + tid;
    a_pk[p] = -1;
    if (issuer && s < W7_RAW_SLOTS) {
      const int c2 = s % W7_C2, r1 = s / W7_C2;
      const int row = r1 % W7_ROWS, r2 = r1 / W7_ROWS;
      const int par = r2 & 1, qd = r2 >> 1;
      const int iy = oy0 - 3 + row, ix = ox0 - 3 + 2 * c2 + par;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pk[p] = 2 * (iy * P.w + ix) + qd;
      else {                                                // zero padding: written once, both buffers; never touched by the DMA
        *reinterpret_cast<float4*>(wsm + W7_RAW_OFF + s * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(wsm + W7_RAW_OFF + W7_RAW_BYTES + s * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (__ballot(a_pk[p] >= 0) != 0ull) piece_live |= 1 << p;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wsm;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  const int nchunks = P.cin / 8;
  const int total_steps = nchunks * 7;
  const float* src_img = P.src + (long long)img * P.img_stride;
  const unsigned pixb = (unsigned)P.ld * 4u;

  auto issue_raw = [&](int chunk, int p) -> int {           // piece p of chunk's image -> raw buffer chunk & 1; returns the DMA instructions issued
    if (!((piece_live >> p) & 1)) return 0;
    const float* sp = src_img + chunk * 8;
    const unsigned la = lds0 + (unsigned)(W7_RAW_OFF + (chunk & 1) * W7_RAW_BYTES + p * 4096);
#pragma unroll
    for (int pp = 0; pp < W7_PIECES; ++pp)
      if (pp == p && a_pk[pp] >= 0) w7_glds16((unsigned)(a_pk[pp] >> 1) * pixb + 16u * (unsigned)(a_pk[pp] & 1), sp, la);
    return 1;
  };
  // weight slab of global step g = chunk * 7 + ky -> ring slot g % RING; slot e = i * 512 + tid of [nu][quad][CO]: NT instructions per wave
  const unsigned u_off0 = (unsigned)(((tid / CO) * P.cout + n0 + (tid % CO)) * 4) * 4u;
  const unsigned u_step = (unsigned)((512 / CO) * P.cout * 4) * 4u;
  auto issue_slab = [&](int g) {
    const float* wp = P.weight + (long long)g * (16 * P.cout * 4);
    const unsigned la = lds0 + (unsigned)(W7_U_OFF + (g % RING) * SLAB);
    unsigned bo = u_off0;
#pragma unroll
    for (int i = 0; i < NT; ++i) { w7_glds16(bo, wp, la + i * 8192u); bo += u_step; }
  };

  f32x16 acc[4][NT];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;

  // ---- prologue: the whole raw image of chunk 0, slabs 0 .. LA-1 ----
#pragma unroll
  for (int p = 0; p < W7_PIECES; ++p) issue_raw(0, p);
  for (int g = 0; g < LA && g < total_steps; ++g) issue_slab(g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // fragment addresses: V[(row * 8 + nu) * 2 + lh][li], U slab [(nu * 2 + lh) * CO + nt * 32 + li]
  const unsigned v_frag = (unsigned)((((yw * 8 + 4 * vh) * 2 + lh) * 32 + li) * 16);
  const unsigned u_frag = (unsigned)(W7_U_OFF + (((4 * vh) * 2 + lh) * CO + li) * 16);

  for (int c = 0; c < nchunks; ++c) {
    // ---- (T) input transform of chunk c.  Round 1: item (row 0..7, quad, pair) per thread, all eight positions.  Round 2: rows 8 and 9 are
    // only 128 items -- each is cut into four position pairs (0,7), (1,2), (3,4), (5,6) so that all 512 threads work (a second full round
    // for a quarter of the threads cost as much as the first) ----
    {
      const char* rawb = wsm + W7_RAW_OFF + (c & 1) * W7_RAW_BYTES;
      const int j = tid & 31, q = (tid >> 5) & 1;
      auto load8 = [&](int row, float4 (&d)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          d[i] = *reinterpret_cast<const float4*>(rawb + ((((q * 2 + (i & 1)) * W7_ROWS + row) * W7_C2) + j + (i >> 1)) * 16);
      };
      // B^T d  (points 0, +-1, +-2, +-1/2, infinity; rows 1/2, 3/4, 5/6 = even part +- odd part); sel: -1 all, else the pair 0..3
      auto transform = [&](const float4 (&d)[8], char* vb, const int sel) {
        if (sel < 0 || sel == 0) {
          *reinterpret_cast<float4*>(vb + 0 * 1024) = W7_F4(-, w7_fma(5.25f, W7_F4(-, d[4], d[2]), d[0]), d[6]);
          *reinterpret_cast<float4*>(vb + 7 * 1024) = W7_F4(-, w7_fma(5.25f, W7_F4(-, d[3], d[5]), d[7]), d[1]);
        }
        if (sel < 0 || sel == 1) {
          const float4 e1 = w7_fma(17.f / 18.f, d[4], w7_mul(-2.f / 9.f, W7_F4(+, d[2], d[6])));
          const float4 o1 = w7_fma(17.f / 18.f, d[3], w7_mul(-2.f / 9.f, W7_F4(+, d[1], d[5])));
          *reinterpret_cast<float4*>(vb + 1 * 1024) = W7_F4(+, e1, o1);
          *reinterpret_cast<float4*>(vb + 2 * 1024) = W7_F4(-, e1, o1);
        }
        if (sel < 0 || sel == 2) {
          const float4 e3 = w7_fma(1.f / 360.f, d[2], w7_fma(-1.f / 72.f, d[4], w7_mul(1.f / 90.f, d[6])));
          const float4 o3 = w7_fma(1.f / 180.f, d[1], w7_fma(-1.f / 36.f, d[3], w7_mul(1.f / 45.f, d[5])));
          *reinterpret_cast<float4*>(vb + 3 * 1024) = W7_F4(+, e3, o3);
          *reinterpret_cast<float4*>(vb + 4 * 1024) = W7_F4(-, e3, o3);
        }
        if (sel < 0 || sel == 3) {
          const float4 e5 = w7_fma(128.f / 45.f, d[2], w7_fma(-32.f / 9.f, d[4], w7_mul(32.f / 45.f, d[6])));
          const float4 o5 = w7_fma(64.f / 45.f, d[1], w7_fma(-16.f / 9.f, d[3], w7_mul(16.f / 45.f, d[5])));
          *reinterpret_cast<float4*>(vb + 5 * 1024) = W7_F4(+, e5, o5);
          *reinterpret_cast<float4*>(vb + 6 * 1024) = W7_F4(-, e5, o5);
        }
      };
      {
        const int row = tid >> 6;                              // 0..7
        float4 d[8];
        load8(row, d);
        transform(d, wsm + ((row * 8 * 2 + q) * 32 + j) * 16, -1);
      }
      {
        const int row = 8 + ((tid >> 6) & 1), sub = wave >> 1;  // (wave-uniform)
        float4 d[8];
        load8(row, d);
        char* vb = wsm + ((row * 8 * 2 + q) * 32 + j) * 16;
        if (sub == 0) transform(d, vb, 0);
        else if (sub == 1) transform(d, vb, 1);
        else if (sub == 2) transform(d, vb, 2);
        else transform(d, vb, 3);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- (M) seven filter rows ----
    int prev_ops = 0;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      const int g = c * 7 + ky;
      int ops = 0;
      if (c + 1 < nchunks && ky < W7_PIECES) ops += issue_raw(c + 1, ky);      // the next chunk's image, a piece per step
      if (g + LA < total_steps) { issue_slab(g + LA); ops += NT; }            // -> the slot the last barrier freed
      const char* vrow = wsm + v_frag + ky * (8 * 2 * 32 * 16);
      const char* us = wsm + u_frag + (g % RING) * SLAB;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const float4 vf = *reinterpret_cast<const float4*>(vrow + nu * (2 * 32 * 16));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float4 uf = *reinterpret_cast<const float4*>(us + nu * (2 * CO * 16) + nt * 512);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, uf.x, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, uf.y, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, uf.z, acc[nu][nt], 0, 0, 0);
          acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, uf.w, acc[nu][nt], 0, 0, 0);
        }
      }
      // slab g + 1 (issued LA steps ago) must have landed: what was issued after it may stay in flight -- the issues of the last LA - 1
      // steps.  At the end of a chunk everything lands (the next transform reads the raw image issued during this chunk).
      if (ky == 6) w7_wait_vmcnt(0);
      else w7_wait_vmcnt(LA >= 3 ? prev_ops + ops : (LA == 2 ? ops : 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      prev_ops = ops;
    }
  }

  // ---- epilogue: A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 1] over this wave's four positions, halves joined through LDS ----
  float* E = reinterpret_cast<float*>(wsm);                  // [yw][a][nt][16 registers][64 lanes]: the V buffer is free now
  if (vh == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z0 = (acc[0][nt][r] + acc[1][nt][r]) + acc[2][nt][r];
        const float z1 = fmaf(-2.f, acc[0][nt][r], acc[3][nt][r]) + 0.5f * (acc[1][nt][r] - acc[2][nt][r]);
        E[(((yw * 2 + 0) * NT + nt) * 16 + r) * 64 + lane] = z0;
        E[(((yw * 2 + 1) * NT + nt) * 16 + r) * 64 + lane] = z1;
      }
  }
  __syncthreads();
  if (vh == 0) {
    const int oy = oy0 + yw;
    float* out_img = P.out + (long long)img * P.h * P.w * P.out_ld;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + nt * 32 + li;
      const float b = P.bias ? P.bias[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pair = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float z0 = ((acc[0][nt][r] + acc[1][nt][r]) + (acc[2][nt][r] + acc[3][nt][r])) + E[(((yw * 2 + 0) * NT + nt) * 16 + r) * 64 + lane];
        float z1 = (fmaf(2.f, acc[3][nt][r], acc[1][nt][r]) - acc[2][nt][r]) + E[(((yw * 2 + 1) * NT + nt) * 16 + r) * 64 + lane];
        z0 = apply_act(z0 + b, P.act); z1 = apply_act(z1 + b, P.act);
        const int ox = ox0 + 2 * pair;
        if (oy < P.h && ox < P.w) out_img[((long long)oy * P.w + ox) * P.out_ld + co] = z0;
        if (oy < P.h && ox + 1 < P.w) out_img[((long long)oy * P.w + ox + 1) * P.out_ld + co] = z1;
      }
    }
  }
}

// descriptor.transposed == 4: called from gpemsr_conv2d (conv_mfma.hip)
int conv2d_winograd7(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap) {
  GP_REQUIRE(d->ksize == 7 && d->stride == 1 && d->weight_image_stride == 0 && d->nsrc == 1, "conv2d (F(2,7) form): 7x7, stride 1, one source, one weight set");
  GP_REQUIRE(!d->residual && !d->pixmul && !d->pixel_shuffle && !d->gn_partials && !d->cos_partials, "conv2d (F(2,7) form): plain store only");
  GP_REQUIRE(d->cout % 32 == 0 && d->src[0].c % 8 == 0 && d->src[0].ld % 4 == 0 && d->src[0].ld >= d->src[0].c &&
             (reinterpret_cast<uintptr_t>(d->src[0].ptr) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->weight) & 15) == 0,
             "conv2d (F(2,7) form): cout %% 32 == 0, source c %% 8 == 0 with 16-byte aligned rows");
  const bool wide = d->cout % 64 == 0;
  if (name_buf) { snprintf(name_buf, (size_t)name_cap, wide ? "conv7_wino_f32_kernel<64>" : "conv7_wino_f32_kernel<32>"); return GPEMSR_OK; }
  W7Params P{};
  P.src = d->src[0].ptr; P.ld = d->src[0].ld; P.cin = d->src[0].c;
  P.img_stride = d->src_image_stride[0] < 0 ? (long long)d->h * d->w * d->src[0].ld : d->src_image_stride[0];
  GP_REQUIRE(P.img_stride % 4 == 0 && (long long)d->h * d->w * P.ld * 4 < (1ll << 32), "conv2d (F(2,7) form): source too large / misaligned");
  GP_REQUIRE((long long)7 * 16 * d->cout * 4 * 4 < (1ll << 32), "conv2d (F(2,7) form): weight slab offsets exceed 32 bits");
  P.n = d->n; P.h = d->h; P.w = d->w; P.cout = d->cout;
  P.weight = d->weight; P.bias = d->bias; P.act = d->act; P.out = d->out; P.out_ld = d->out_ld;
  P.tiles_x = cdiv(d->w, 64); P.tiles_y = cdiv(d->h, W7_TH); P.tiles_n = d->cout / (wide ? 64 : 32);
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d (F(2,7) form): grid too large");
  P.nblocks = (int)nb;
  static dev_once_t done{0};
  if (dev_once_begin(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_wino_f32_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, W7_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_wino_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, W7_LDS) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d (F(2,7) form): cannot raise the dynamic LDS limit to %d bytes", W7_LDS);
    dev_once_done(done);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (wide) hipLaunchKernelGGL(conv7_wino_f32_kernel<2>, dim3(P.nblocks), dim3(512), W7_LDS, st, P);
  else hipLaunchKernelGGL(conv7_wino_f32_kernel<1>, dim3(P.nblocks), dim3(512), W7_LDS, st, P);
  return check_launch("conv7_wino_f32_kernel");
}

}  // namespace gpemsr
