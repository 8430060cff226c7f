// Winograd F(4x4, 3x3) form of the 3x3 stride-1 convolution on the exact-fp32 matrix pipe, for layers with MANY input channels.
//
// F(2x2, 3x3) (conv_wino.hip) does 16 multiplies per 2x2 outputs = 4 per output; F(4x4, 3x3) does 36 per 4x4 outputs = 2.25 per output
// (direct: 9): executed MFMA FLOPs = 1/4 of the algorithmic ones.
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d = 6x6 input patch (pad 1: it starts one pixel up / left of the 4x4 output block)
// with the interpolation points 0, +-1, +-2, infinity (Lavin & Gray's scaling: B^T is integer, G carries the fractions and is folded into U on
// the host in float64).  Thirty-six independent GEMMs  M_p[block][cout] = sum_cin V_p[block][cin] U_p[cin][cout]  on 1/16 of the pixels.  All
// arithmetic is fp32; simulated and measured error ~2e-5 of the result at 512 channels (F(2x2): ~5e-6; direct: 3e-6; the path's bar: 1e-3).
//
// The price is transform work per MULTIPLY four times that of F(2x2) and an exchange of 36 x 32 x 64 sums per tile in the epilogue, so this
// kernel is for the layers where a tile's main loop is long: >= 16 chunks of 8 input channels (the 128-512-channel layers of the VQGAN prior and
// the first prior-fusion convolutions, R:model/blocks.py:5-29, R:model/GPEMSR.py:255-262); 64-channel layers keep F(2x2).
//
//   * workgroup = 12 waves (three per SIMD, 168 registers), output tile 16 x 32 pixels = 32 blocks of 4x4 (one MFMA row tile) x 64 couts;
//     wave w owns the positions p = 3 w .. 3 w + 2 (p = 6 xi + nu): 3 x 2 accumulator tiles = 96 registers;
//   * NO weights in LDS: a position is multiplied by exactly one wave, so its U rows go global -> registers (U is packed
//     [cin / 8][36][quad][cout][4]: a wave's fragment is 512 consecutive bytes per quad), one chunk ahead, into the registers the
//     previous chunk's fragments have just left; the raw halo image goes global -> registers -> LDS one chunk ahead too.  There is no LDS-DMA
//     in this kernel and no branch around a load, so every wait is the compiler's own and exact;
//   * per chunk: (T1) the row transform B^T along x of the 18 x 34 halo image into X[row][nu][quad][block column] (576 items of one channel
//     pair), (T2) the column transform into V[xi][nu][quad][block] (768 items: every thread one), (M) 24 MFMAs per wave from V and the
//     register-held U; three barriers.  Pair-sized items keep the transform's transient registers at 24 beside the 96 accumulators;
//   * raw image in LDS as [quad][column mod 4][row][column / 4]: the six columns an item reads for consecutive block columns are consecutive
//     16-byte slots (a pixel-major image would be read at a 64-byte stride);
//   * epilogue: the 36 sums of every (block, cout) meet through LDS in four passes of 16 couts (92 KB each), A^T . A in registers (10 + 10
//     additions per row / column pass), bias, GroupNorm partial sums (conv + bias, per tile and channel, fixed order), activation, store.
//
// Replaces gpemsr_conv2d's direct form (descriptor.transposed = 5; weight = packing.pack_winograd4) for 3x3 stride-1 layers whose sources are
// multiples of 8 channels (>= 128 in all), cout % 64 == 0, no residual / multiplier / PixelShuffle.
#include "common.h"
#include "conv_wino.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// timing experiments only (scripts/build_wino_probe_lib.sh compiles variants with -DW4_SKIP=mask; results are wrong on purpose):
// 1 no T1, 2 no T2, 4 no MFMAs, 8 no U loads after the prologue, 16 no image loads after the prologue.  Compile-time: a run-time branch around a
// load would make the compiler's vmcnt counts conservative and change what is being timed.
#ifndef W4_SKIP
#define W4_SKIP 0
#endif

struct W4Params {
  const float* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w, cout;
  const float* weight;            // U [cin / 8][36][2 quads][cout][4]
  const float* bias; int act;
  float* out; int out_ld;
  float* gn_ws; int gn_parts;     // GroupNorm partial sums of (conv + bias): [n][gn_parts = tiles per image][cout][2]
  int tiles_x, tiles_y, tiles_n, nblocks;
};

constexpr int W4_NT = 768;                                               // threads
constexpr int W4_RAW_SLOTS = 2 * 4 * 18 * 9;                             // [quad][col & 3][row 18][col >> 2 (9)] = 1296 16-byte slots
constexpr int W4_RAW_BYTES = W4_RAW_SLOTS * 16;                          // 20,736
constexpr int W4_XROW = 6 * 2 * 8 * 16 + 32;                             // X row stride: 1,536 + 32 bytes -- the four block rows a T2 wave reads (4 rows apart)
                                                                         // then fall on alternating 128-byte bank halves instead of one (4-way conflict)
constexpr int W4_X_BYTES = 18 * W4_XROW;                                 // X[row][nu][quad][block column]: 28,224
constexpr int W4_V_BYTES = 36 * 2 * 32 * 16;                             // V[p][quad][block]: 36,864
constexpr int W4_X_OFF = 2 * W4_RAW_BYTES, W4_V_OFF = W4_X_OFF + W4_X_BYTES;
constexpr int W4_MAIN = W4_V_OFF + W4_V_BYTES;                           // 106,560
constexpr int W4_EPIX = 20;                                              // floats per (position, block) row of the exchange buffer: 16 couts + 4
constexpr int W4_E_BYTES = 36 * 32 * W4_EPIX * 4;                        // 92,160 (overlays raw / X / V)
constexpr int W4_RED_OFF = W4_MAIN;                                      // GroupNorm sums [32 blocks][64 couts][2]: 16,384
constexpr int W4_LDS = W4_RED_OFF + 16384;                               // 122,944

// B^T of F(4, 3) (integer form): one 6-vector, component-wise on a channel pair
__device__ __forceinline__ void w4_bt(const float2 (&d)[6], float2 (&t)[6]) {
#define W4_LIN2(a, sa, b) make_float2(fmaf(sa, (a).x, (b).x), fmaf(sa, (a).y, (b).y))
#define W4_ADD(a, b) make_float2((a).x + (b).x, (a).y + (b).y)
#define W4_SUB(a, b) make_float2((a).x - (b).x, (a).y - (b).y)
  const float2 a = W4_LIN2(d[2], -4.f, d[4]);            // d4 - 4 d2
  const float2 b = W4_LIN2(d[1], -4.f, d[3]);            // d3 - 4 d1
  const float2 c = W4_SUB(d[4], d[2]);
  const float2 e = W4_SUB(d[3], d[1]);
  t[0] = W4_LIN2(d[0], 4.f, W4_LIN2(d[2], -5.f, d[4]));  // 4 d0 - 5 d2 + d4
  t[1] = W4_ADD(a, b);
  t[2] = W4_SUB(a, b);
  t[3] = W4_LIN2(e, 2.f, c);
  t[4] = W4_LIN2(e, -2.f, c);
  t[5] = W4_LIN2(d[1], 4.f, W4_LIN2(d[3], -5.f, d[5]));  // 4 d1 - 5 d3 + d5
}
// A^T of F(4, 3): six sums -> four outputs
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&y)[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y[0] = (m[0] + s12) + s34;
  y[1] = fmaf(2.f, d34, d12);
  y[2] = fmaf(4.f, s34, s12);
  y[3] = fmaf(8.f, d34, d12) + m[5];
}

__global__ __launch_bounds__(W4_NT, 3) void conv_wino4_f32_kernel(W4Params P) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  {   // XCD-aware (bijective): consecutive logical blocks -- neighbouring pixel tiles of one cout block -- share an L2 (the U slice, halo rows)
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = bid;
  const int tx0 = t % P.tiles_x; t /= P.tiles_x;
  const int ty0 = t % P.tiles_y; t /= P.tiles_y;
  const int img = t % P.n; t /= P.n;
  const int tn = t;                                           // cout block slowest: the workgroups running together read ONE U slice
  const int oy0 = ty0 * 16, ox0 = tx0 * 32, n0 = tn * 64;

  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += P.c[s] / 8;

  // ---- raw halo image: two 16-byte slots per thread, slot s = tid + 768 i of [quad][col & 3][row][col >> 2]; pixel index or -1 ----
  //      global -> registers at the start of a chunk's MFMA phase -> LDS at its end (the phase is >= 4,600 matrix clocks per SIMD: longer than an
  //      HBM round trip).  Every lane loads in every chunk -- a slot outside the image reads pixel 0 and is zeroed on its way to LDS, the last
  //      chunk re-reads itself -- so no branch surrounds a load and the compiler's vmcnt counts are exact: the waits for the U fragments
  //      leave the two younger image loads in flight.  (A first version fetched the image by LDS-DMA at the top of the chunk: the DMA is
  //      invisible to those counts, the fragment waits became vmcnt(0) and every chunk stalled for the image's round trip -- 8.0 k instead
  //      of 5.4 k clocks per chunk.)
  int r_pix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = tid + i * W4_NT;
    r_pix[i] = -1;
    if (s < W4_RAW_SLOTS) {
      const int c4 = s % 9, r1 = s / 9;
      const int row = r1 % 18, ph = (r1 / 18) & 3;
      const int col = 4 * c4 + ph;
      const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
      if (col < 34 && iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) r_pix[i] = iy * P.w + ix;
    }
  }
  const int r_q0 = tid >= W4_RAW_SLOTS / 2 ? 4 : 0;           // channel offset of slot 0's quad (slot 1 = tid + 768 is always quad 1)
  int f_src = 0, f_c0 = 0;                                    // source cursor of the chunk whose raw image is loaded next
  auto load_raw = [&](float4 (&rr)[2]) {
    const float* sp = P.src[f_src] + (long long)img * P.img_stride[f_src] + f_c0;
    const int ldp = P.ld[f_src];
    rr[0] = *reinterpret_cast<const float4*>(sp + (long long)max(r_pix[0], 0) * ldp + r_q0);
    rr[1] = *reinterpret_cast<const float4*>(sp + (long long)max(r_pix[1], 0) * ldp + 4);
    asm volatile("" ::: "memory");                            // the image loads stay OLDER than the U loads that follow (vmcnt is in order)
    f_c0 += 8;
    if (f_c0 >= P.c[f_src]) {
      if (f_src + 1 < P.nsrc) { f_c0 = 0; ++f_src; } else f_c0 -= 8;
    }
  };
  auto store_raw = [&](int buf, const float4 (&rr)[2]) {      // (the zeroing is a bit mask: a select of two float4 made the compiler go through scratch)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = ~(r_pix[i] >> 31);
      asm volatile("" : "+v"(m) :: "memory");                 // the masking (= the wait for the image loads) stays BEHIND the U loads of this chunk
      const float4 v = make_float4(__int_as_float(__float_as_int(rr[i].x) & m), __int_as_float(__float_as_int(rr[i].y) & m),
                                   __int_as_float(__float_as_int(rr[i].z) & m), __int_as_float(__float_as_int(rr[i].w) & m));
      if (i == 0 || tid + W4_NT < W4_RAW_SLOTS) *reinterpret_cast<float4*>(wsm + buf * W4_RAW_BYTES + (tid + i * W4_NT) * 16) = v;
    }
  };
  // ---- U fragments of this wave's three positions: lane (li = cout, lh = quad), [chunk][p][quad][cout][4] ----
  const float* u_lane = P.weight + ((long long)(3 * wave * 2 + lh) * P.cout + n0 + li) * 4;
  const long long u_chunk = (long long)36 * 2 * P.cout * 4;
  auto load_u = [&](int chunk, int j, float4 (&U)[2]) {
    const float* up = u_lane + (long long)chunk * u_chunk + (long long)j * (2 * P.cout * 4);
    U[0] = *reinterpret_cast<const float4*>(up);
    U[1] = *reinterpret_cast<const float4*>(up + 32 * 4);
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;

  // ---- prologue: raw image and U fragments of chunk 0 ----
  float4 U[3][2], rr[2];
  load_raw(rr);
#pragma unroll
  for (int j = 0; j < 3; ++j) load_u(0, j, U[j]);
  store_raw(0, rr);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int cn = c + 1 < nchunks ? c + 1 : c;              // (the last chunk re-reads its own U and image: no branch around the loads)
    int tl = tid;                                            // opaque copy: the item addresses are recomputed per chunk (a few integer operations)
    asm volatile("" : "+v"(tl));                             // rather than hoisted out of the loop into registers the accumulators need
    const int hb = (tl & 1) * 8, t_bc = (tl >> 1) & 7;       // transform items: one channel PAIR of a quad (8 bytes) -- all 12 waves take part
    // ---- (T1) row transform along x: item (row, quad, block column, pair) ----
    if (wave < 9 && !(W4_SKIP & 1)) {
      const int q = (tl >> 4) & 1, row = tl >> 5;
      const char* rb = wsm + (c & 1) * W4_RAW_BYTES + hb;
      float2 d[6], tt[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = *reinterpret_cast<const float2*>(rb + ((((q * 4 + (i & 3)) * 18 + row) * 9) + t_bc + (i >> 2)) * 16);
      w4_bt(d, tt);
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<float2*>(wsm + W4_X_OFF + hb + row * W4_XROW + (((nu * 2 + q) * 8) + t_bc) * 16) = tt[nu];
    }
    __syncthreads();
    // ---- (T2) column transform along y: item (nu, quad, block row, block column, pair); (nu, quad) is wave-uniform ----
    if (!(W4_SKIP & 2)) {
      const int br = (tl >> 4) & 3, q = wave & 1, nu = wave >> 1;
      float2 d[6], tt[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = *reinterpret_cast<const float2*>(wsm + W4_X_OFF + hb + (4 * br + i) * W4_XROW + (((nu * 2 + q) * 8) + t_bc) * 16);
      w4_bt(d, tt);
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) *reinterpret_cast<float2*>(wsm + W4_V_OFF + hb + ((((xi * 6 + nu) * 2 + q) * 32) + br * 8 + t_bc) * 16) = tt[xi];
    }
    __syncthreads();
    // ---- (M) this wave's three positions; the next chunk's raw image and U fragments are fetched underneath ----
    if (!(W4_SKIP & 16)) load_raw(rr);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 vf = *reinterpret_cast<const float4*>(wsm + W4_V_OFF + (3 * wave + j) * 1024 + (tl & 63) * 16);   // [p][lh][li]
      if (!(W4_SKIP & 4))
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, U[j][nt].x, acc[j][nt], 0, 0, 0);
        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, U[j][nt].y, acc[j][nt], 0, 0, 0);
        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, U[j][nt].z, acc[j][nt], 0, 0, 0);
        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, U[j][nt].w, acc[j][nt], 0, 0, 0);
      }
      if (!(W4_SKIP & 8)) load_u(cn, j, U[j]);
    }
    store_raw((c + 1) & 1, rr);                              // (waits for the two image loads only: the six U loads behind them stay in flight)
    __syncthreads();                                         // V may be rewritten; the next raw image is visible
  }

  // ---- epilogue: four passes of 16 couts through the exchange buffer ----
  float* E = reinterpret_cast<float*>(wsm);
  float* red = reinterpret_cast<float*>(wsm + W4_RED_OFF);
  float* out_img = P.out + (long long)img * P.h * P.w * P.out_ld;
  const int act = P.act;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int nt = k >> 1, half = k & 1;
    if ((li >> 4) == half) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // block of this register
          const float v = nt == 0 ? acc[j][0][r] : acc[j][1][r];
          E[((3 * wave + j) * 32 + row) * W4_EPIX + (li & 15)] = v;
        }
    }
    __syncthreads();
    if (tid < 512) {
      const int cc = tid & 15, b = tid >> 4;                  // (cout of this pass, block)
      const int br = b >> 3, bc = b & 7;
      const int co = n0 + nt * 32 + 16 * half + cc;
      const float bias = P.bias ? P.bias[co] : 0.f;
      float z[6][4];                                          // A^T over xi for every nu: z[nu][i]
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        float m[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) m[xi] = E[((xi * 6 + nu) * 32 + b) * W4_EPIX + cc];
        w4_at(m, z[nu]);
      }
      float gs = 0.f, gq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m[6] = {z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i]};
        float y[4];
        w4_at(m, y);
        const int oy = oy0 + 4 * br + i;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
          const int ox = ox0 + 4 * bc + jx;
          if (oy < P.h && ox < P.w) {
            float v = y[jx] + bias;
            gs += v; gq = fmaf(v, v, gq);
            v = apply_act(v, act);
            out_img[((long long)oy * P.w + ox) * P.out_ld + co] = v;
          }
        }
      }
      if (P.gn_ws) { red[(b * 64 + nt * 32 + 16 * half + cc) * 2] = gs; red[(b * 64 + nt * 32 + 16 * half + cc) * 2 + 1] = gq; }
    }
    __syncthreads();
  }
  if (P.gn_ws && tid < 64) {                                   // per (tile, channel): the 32 blocks in fixed order
    float s = 0.f, q = 0.f;
    for (int b = 0; b < 32; ++b) { s += red[(b * 64 + tid) * 2]; q += red[(b * 64 + tid) * 2 + 1]; }
    const int part = ty0 * P.tiles_x + tx0;
    float* wsp = P.gn_ws + (((long long)img * P.gn_parts + part) * P.cout + n0 + tid) * 2;
    wsp[0] = s; wsp[1] = q;
  }
}

// descriptor.transposed == 5: called from gpemsr_conv2d (conv_mfma.hip); parts_only != NULL: only report the GroupNorm records per image
int conv2d_winograd4(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap, int* parts_only) {
  GP_REQUIRE(d->ksize == 3 && d->stride == 1 && d->weight_image_stride == 0, "conv2d (F(4x4,3x3) form): 3x3, stride 1, one weight set");
  GP_REQUIRE(!d->residual && !d->pixmul && !d->pixel_shuffle && !d->cos_partials, "conv2d (F(4x4,3x3) form): plain store only");
  GP_REQUIRE(d->cout % 64 == 0, "conv2d (F(4x4,3x3) form): cout %% 64 == 0 (got %d)", d->cout);
  if (d->gn_partials || parts_only) GP_REQUIRE(d->act == GPEMSR_ACT_NONE, "conv2d (F(4x4,3x3) form): GroupNorm partial sums need act NONE");
  if (parts_only) { *parts_only = cdiv(d->h, 16) * cdiv(d->w, 32); return GPEMSR_OK; }
  if (name_buf) { snprintf(name_buf, (size_t)name_cap, "conv_wino4_f32_kernel"); return GPEMSR_OK; }
  W4Params P{};
  int cin = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].c % 8 == 0 && d->src[s].ld % 4 == 0 && d->src[s].ld >= d->src[s].c &&
               (reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0, "conv2d (F(4x4,3x3) form): source %d needs c %% 8 == 0, 16-byte aligned rows", s);
    P.src[s] = d->src[s].ptr; P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    GP_REQUIRE(P.img_stride[s] % 4 == 0 && (long long)d->h * d->w * d->src[s].ld * 4 < (1ll << 32),
               "conv2d (F(4x4,3x3) form): source %d misaligned, or an image beyond the 32-bit byte offsets of the LDS-DMA", s);
    cin += d->src[s].c;
  }
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->out) & 3) == 0 && (long long)d->h * d->w < (1ll << 31),
             "conv2d (F(4x4,3x3) form): weight alignment / image size");
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w; P.cout = d->cout;
  P.weight = d->weight; P.bias = d->bias; P.act = d->act; P.out = d->out; P.out_ld = d->out_ld;
  P.gn_ws = d->gn_partials; P.gn_parts = cdiv(d->h, 16) * cdiv(d->w, 32);
  P.tiles_x = cdiv(d->w, 32); P.tiles_y = cdiv(d->h, 16); P.tiles_n = d->cout / 64;
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d (F(4x4,3x3) form): grid too large");
  P.nblocks = (int)nb;
  static dev_once_t done{0};
  if (dev_once_begin(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d (F(4x4,3x3) form): cannot raise the dynamic LDS limit to %d bytes", W4_LDS);
    dev_once_done(done);
  }
  hipLaunchKernelGGL(conv_wino4_f32_kernel, dim3(P.nblocks), dim3(W4_NT), W4_LDS, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("conv_wino4_f32_kernel");
}

}  // namespace gpemsr
