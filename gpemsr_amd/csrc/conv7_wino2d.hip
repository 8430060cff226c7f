// 7x7 stride-1 convolution on the exact-fp32 matrix pipe in the 2-D Winograd form F(2x2, 7x7).
//
// SpyNet's 32 -> 64 and 64 -> 32 convolutions (basicsr spynet_arch.BasicModule, called from POD, R:model/GPEMSR.py:67,98-100) are 62 ms of the
// 458 ms fp32 step in the 1-D form F(2, 7) (conv7_wino.hip: 8 multiplies per output pair and FILTER ROW = 112 per 2x2 outputs; direct: 196).
// The 2-D form needs 64:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d = the 8x8 input patch that starts three pixels up / left of the 2x2 output block
// with the same interpolation points 0, +-1, +-2, +-1/2, infinity in both directions -- sixty-four independent GEMMs
// M_p[block][cout] = sum_cin V_p[block][cin] U_p[cin][cout], p = 8 xi + nu, on a quarter of the pixels.  All arithmetic is fp32 (U folded on
// the host in float64); error ~4-5e-6 of the result on a 32 / 64-channel layer (1-D form 9e-7, direct 7e-7; the bar on the flows is 2e-3).
//
//   * workgroup = 8 waves (two per SIMD, up to 256 registers), output tile 8 x 16 pixels = 32 blocks of 2x2 (one MFMA row tile) x 32 couts;
//     wave w owns ROW xi = w of the transformed tile: the eight positions nu = 0..7 = 8 accumulator tiles = 128 registers -- so the first output
//     transform (A^T over nu) runs on the wave's own accumulators in registers, as in the F(4x4) kernel (conv_wino4.hip);
//   * NO weights in LDS: a position is multiplied by exactly one wave, its U rows go global -> registers one chunk ahead (U is packed
//     [cin / 8][64][quad][cout][4]); the raw halo image (14 x 22 pixels x 8 channels) goes global -> LDS by DMA three chunks ahead (ring of 3);
//   * per chunk of 8 input channels: (M) 32 MFMAs per wave from V and the register-held U, with (T1) of the NEXT chunk -- the row transform B^T
//     along x of the 14 halo rows into X[row][nu][quad][block column], 448 items of one channel pair -- between them; barrier; (T2) the column
//     transform of the next chunk into V[xi][nu][quad][block], two pair-sized items per thread; barrier.  ONE V buffer (64 KB: two do not fit), so
//     T2 is not overlapped with MFMAs;
//   * raw image in LDS as [quad][column parity][row][column / 2]: the eight columns an item reads for consecutive block columns are
//     consecutive 16-byte slots;
//   * epilogue: A^T over nu in registers (8 -> 2 columns), one exchange E[xi][column][block][cout] through LDS, an item = one block x one cout
//     PAIR: A^T over xi (8 -> 2 rows), bias, activation, 8-byte stores;
//   * persistent workgroups (one per CU), the next tile's first raw images / U fragments / bias requested during the last chunk of this one.
//
// Measured (profiles/r06_ab_winograd77.log; 80 frames): 64 -> 32 at 512^2 24.7 -> 18.2 ms, 32 -> 64 at 512^2 21.9 -> 20.0 ms (two cout blocks =
// two workgroups, each with its own transforms), the fp32 step 458.3 -> 446.6 ms.  Per tile 2.8 us fixed (4.4 before the persistent form) + 3.2 us
// per 8-channel chunk against 2.0 us of pure MFMA time: ~150 vector instructions per wave and chunk (the 8-point transforms are 30 operations
// per 8 outputs, twice) beside 32 MFMAs, and on this chip they are matrix time.  k-steps of two positions interleaved:
// measured 1.4 % slower, removed.  <16>: SpyNet's 32 -> 16 layers (the direct row-pair form before: 99 TFLOP/s) on v_mfma_f32_16x16x4_f32 --
// 16 couts per workgroup, two 16-block tiles per position, the same V: the step 443.4 -> 440.6 ms.
//
// Replaces gpemsr_conv2d's direct form (descriptor.transposed = 6; weight = packing.pack_winograd77) for 7x7 stride-1 layers with one fp32
// source of c % 8 == 0 channels, cout % 16 == 0, plain store (8-byte aligned rows), activation NONE / RELU / LRELU.
#include "common.h"
#include "conv_wino.h"
#include <type_traits>

namespace gpemsr {

struct W77Params {
  const float* src; long long img_stride; int ld, cin;
  int n, h, w, cout;
  const float* weight;            // U [cin/8][64 = 8 xi + nu][2 quads][cout][4]
  const float* bias; int act;
  float* out; int out_ld;
  int tiles_x, tiles_y, tiles_n, nblocks;
};

constexpr int W77_NT = 512;
constexpr int W77_ROWS = 14, W77_C2 = 11;                                  // halo rows; columns per parity ((16 + 6) / 2)
constexpr int W77_RAW_SLOTS = 2 * 2 * W77_ROWS * W77_C2;                   // 616 16-byte slots: [quad][parity][row][col / 2]
constexpr int W77_RAW_BYTES = W77_RAW_SLOTS * 16;                          // 9,856
constexpr int W77_X_BYTES = W77_ROWS * 8 * 2 * 8 * 16;                     // X[row][nu][quad][block column]: 28,672
constexpr int W77_V_BYTES = 64 * 2 * 32 * 16;                              // V[p][quad][block]: 65,536
constexpr int W77_V_OFF = 0, W77_X_OFF = W77_V_BYTES, W77_RAW_OFF = W77_X_OFF + W77_X_BYTES;
constexpr int W77_LDS = W77_RAW_OFF + 3 * W77_RAW_BYTES;                   // 123,776
constexpr int W77_E_BYTES = 8 * 2 * 32 * 32 * 4;                           // E[xi][column][block][cout]: 65,536, over V
static_assert(W77_E_BYTES <= W77_V_BYTES && W77_LDS <= 160 * 1024, "LDS map");

#define W77_2(op, a, b) make_float2((a).x op (b).x, (a).y op (b).y)
__device__ __forceinline__ float2 w77_fma(float s, const float2& a, const float2& b) { return make_float2(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y)); }
__device__ __forceinline__ float2 w77_mul(float s, const float2& a) { return make_float2(s * a.x, s * a.y); }
#define W77_4(op, a, b) make_float4((a).x op (b).x, (a).y op (b).y, (a).z op (b).z, (a).w op (b).w)
__device__ __forceinline__ float4 w77_fma(float s, const float4& a, const float4& b) {
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 w77_mul(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float2 w77_add(const float2& a, const float2& b) { return W77_2(+, a, b); }
__device__ __forceinline__ float2 w77_sub(const float2& a, const float2& b) { return W77_2(-, a, b); }
__device__ __forceinline__ float4 w77_add(const float4& a, const float4& b) { return W77_4(+, a, b); }
__device__ __forceinline__ float4 w77_sub(const float4& a, const float4& b) { return W77_4(-, a, b); }

// B^T of F(2, 7) for the points 0, +-1, +-2, +-1/2, infinity (packing.wino7_bt): rows 1/2, 3/4, 5/6 are even part +- odd part
template <typename T>
__device__ __forceinline__ void w77_bt(const T (&d)[8], T (&t)[8]) {
  t[0] = w77_sub(w77_fma(5.25f, w77_sub(d[4], d[2]), d[0]), d[6]);
  t[7] = w77_sub(w77_fma(5.25f, w77_sub(d[3], d[5]), d[7]), d[1]);
  const T e1 = w77_fma(17.f / 18.f, d[4], w77_mul(-2.f / 9.f, w77_add(d[2], d[6])));
  const T o1 = w77_fma(17.f / 18.f, d[3], w77_mul(-2.f / 9.f, w77_add(d[1], d[5])));
  t[1] = w77_add(e1, o1); t[2] = w77_sub(e1, o1);
  const T e3 = w77_fma(1.f / 360.f, d[2], w77_fma(-1.f / 72.f, d[4], w77_mul(1.f / 90.f, d[6])));
  const T o3 = w77_fma(1.f / 180.f, d[1], w77_fma(-1.f / 36.f, d[3], w77_mul(1.f / 45.f, d[5])));
  t[3] = w77_add(e3, o3); t[4] = w77_sub(e3, o3);
  const T e5 = w77_fma(128.f / 45.f, d[2], w77_fma(-32.f / 9.f, d[4], w77_mul(32.f / 45.f, d[6])));
  const T o5 = w77_fma(64.f / 45.f, d[1], w77_fma(-16.f / 9.f, d[3], w77_mul(16.f / 45.f, d[5])));
  t[5] = w77_add(e5, o5); t[6] = w77_sub(e5, o5);
}
// A^T of F(2, 7) = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 1]: eight sums -> two outputs, on a pair
__device__ __forceinline__ void w77_at(const float2 (&m)[8], float2& y0, float2& y1) {
  y0 = w77_add(w77_add(w77_add(m[0], m[1]), w77_add(m[2], m[3])), w77_add(w77_add(m[4], m[5]), m[6]));
  y1 = w77_add(w77_add(w77_fma(2.f, w77_sub(m[3], m[4]), w77_sub(m[1], m[2])), w77_mul(0.5f, w77_sub(m[5], m[6]))), m[7]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// CO = couts per workgroup: 32 (v_mfma_f32_32x32x2_f32, one 32 x 32 tile per position) or 16 (SpyNet's 32 -> 16 layers: v_mfma_f32_16x16x4_f32,
// two 16-block x 16-cout tiles per position -- a 32-cout tile would be half empty; same V, same transforms, half the multiplies)
template <int CO>
__global__ __launch_bounds__(W77_NT, 1) void conv7_wino2d_f32_kernel(W77Params P) {
  constexpr bool C16 = CO == 16;
  static_assert(CO == 32 || CO == 16, "couts per workgroup");
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // PERSISTENT workgroups (one per CU: 121 KB of LDS): a tile's fixed cost -- the launch gap, the latency of its first raw images and U
  // fragments -- was 4.4 us of the 16.6 us a 4-chunk tile took; the NEXT tile's raw images and U fragments are requested during this tile's
  // last chunk.  Round k gives the 32 workgroups of XCD x the logical tiles [(8 k + x) 32, + 32): the two cout blocks and the neighbouring
  // pixel tiles of an image share that XCD's L2 (workgroup b runs on XCD b % 8; the grid is a multiple of 8).
  const int per_xcd = gridDim.x >> 3;
  const int tile_step = per_xcd * 8;
  int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  int img = 0, oy0 = 0, ox0 = 0, n0 = 0;
  const int nchunks = P.cin / 8;
  // ---- raw halo image: slots tid and tid + 512 of [quad][parity][row][col / 2]; byte offset of the slot's 16 bytes in the image, or ~0u outside.
  //      Global -> LDS by DMA (inline asm: invisible to the compiler's vmcnt counts, see conv_wino4.hip for the wait discipline copied here) ----
  int s_desc[2];                                              // this thread's two slots: halo row | halo column << 8 | quad << 16 (tile-independent)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = tid + i * W77_NT;
    const int c2 = s % W77_C2, r1 = s / W77_C2;
    const int row = r1 % W77_ROWS, r2 = r1 / W77_ROWS;
    s_desc[i] = row | ((2 * c2 + (r2 & 1)) << 8) | (((r2 >> 1) & 1) << 16);
  }
  unsigned r_off[2];
  const float* src_img = nullptr;
  int f_chunk = 0;                                            // chunk whose raw image is issued next
  auto set_tile = [&](int tl) {                               // decode logical tile tl: cout block fastest, then pixel tiles of one image
    int t = tl;
    const int tn = t % P.tiles_n; t /= P.tiles_n;
    const int tx0 = t % P.tiles_x; t /= P.tiles_x;
    const int ty0 = t % P.tiles_y; t /= P.tiles_y;
    img = t; oy0 = ty0 * 8; ox0 = tx0 * 16; n0 = tn * CO;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = oy0 - 3 + (s_desc[i] & 255), ix = ox0 - 3 + ((s_desc[i] >> 8) & 255);
      r_off[i] = ~0u;
      if (tid + i * W77_NT < W77_RAW_SLOTS && iy >= 0 && iy < P.h && ix >= 0 && ix < P.w)
        r_off[i] = (unsigned)(iy * P.w + ix) * ((unsigned)P.ld * 4u) + 16u * (unsigned)(s_desc[i] >> 16);
    }
    src_img = P.src + (long long)img * P.img_stride;
    f_chunk = 0;
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wsm + (unsigned)wave * 1024u);
  auto issue_raw = [&](int buf) {
    const float* sp = src_img + f_chunk * 8;
    const unsigned la = lds0 + (unsigned)(W77_RAW_OFF + buf * W77_RAW_BYTES);
    float z = 0.f;
    asm volatile("" : "+v"(z));
    const float4 zero = make_float4(z, z, z, z);
    if (r_off[0] != ~0u) wn_glds16(r_off[0], sp, la);
    else *reinterpret_cast<float4*>(wsm + W77_RAW_OFF + buf * W77_RAW_BYTES + tid * 16) = zero;
    if (tid + W77_NT < W77_RAW_SLOTS) {
      if (r_off[1] != ~0u) wn_glds16(r_off[1], sp, la + W77_NT * 16u);
      else *reinterpret_cast<float4*>(wsm + W77_RAW_OFF + buf * W77_RAW_BYTES + (tid + W77_NT) * 16) = zero;
    }
    ++f_chunk;
  };
  // ---- U fragments of this wave's eight positions (row xi = wave): a uniform base + ONE per-lane offset.  CO = 32: lane (li = cout, lh = quad),
  //      a float4 = the quad's four channels = four k-steps.  CO = 16: lane (cout = lane % 16, channel = lane / 16 of a quad), one float per quad ----
  const unsigned u_off = C16 ? (unsigned)((lane & 15) * 4 + (lane >> 4)) : (unsigned)(lh * P.cout + li) * 4u;      // floats
  typedef typename std::conditional<C16, float2, float4>::type UT;
  auto load_u = [&](int chunk, int j, UT& U) {
    const float* ub = P.weight + ((long long)(chunk * 64 + 8 * wave + j) * 2 * P.cout + n0) * 4;   // wave-uniform
    if constexpr (C16) { U.x = ub[u_off]; U.y = ub[u_off + 4 * P.cout]; }
    else U = *reinterpret_cast<const float4*>(ub + u_off);
  };

  typedef typename std::conditional<C16, f32x4, f32x16>::type AccT;
  constexpr int NACC = C16 ? 16 : 8;                           // CO = 16: acc[2 j + half], half = blocks 16 half .. + 15
  constexpr int AR = C16 ? 4 : 16;
  AccT acc[NACC];

  // ---- transform items.  (T1) row transform along x: item (row, quad, block column, channel pair), threads 0-447 (waves 0-6);
  //      (T2) column transform along y: item (nu = wave, quad, block row, block column, channel pair), two per thread (block rows br, br + 2) ----
  const int t1_hb = (tid & 1) * 8, t1_bc = (tid >> 1) & 7, t1_q = (tid >> 4) & 1, t1_row = tid >> 5;
  const int a_t1l = ((t1_q * 2) * W77_ROWS + t1_row) * W77_C2 * 16 + t1_bc * 16 + t1_hb;        // + ((i & 1) * 14 * 11 + (i >> 1)) * 16 + raw image
  const int a_t1s = W77_X_OFF + (((t1_row * 8) * 2 + t1_q) * 8 + t1_bc) * 16 + t1_hb;           // + nu * 256
  const int t2_br = lane >> 5;                                                                   // (and + 2: two items per thread)
  const int a_t2l = W77_X_OFF + ((((2 * t2_br) * 8 + wave) * 2 + t1_q) * 8 + t1_bc) * 16 + t1_hb;  // + i * 2048 (rows 2 br + i), + 8192 for br + 2
  const int a_t2s = W77_V_OFF + (((wave) * 2 + t1_q) * 32 + t2_br * 8 + t1_bc) * 16 + t1_hb;       // + xi * 8192, + 256 for br + 2
  auto t1_load = [&](int buf, float2 (&d)[8]) {
    const char* rb = wsm + W77_RAW_OFF + buf * W77_RAW_BYTES + a_t1l;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = *reinterpret_cast<const float2*>(rb + ((i & 1) * (W77_ROWS * W77_C2) + (i >> 1)) * 16);
  };
  auto t1_store = [&](const float2 (&d)[8]) {
    float2 tt[8];
    w77_bt(d, tt);
#pragma unroll
    for (int nu = 0; nu < 8; ++nu) *reinterpret_cast<float2*>(wsm + a_t1s + nu * 256) = tt[nu];
  };
  auto t2_run = [&]() {                                       // two pair-sized items per thread (block rows br, br + 2): a float4 item's 64 transient
#pragma unroll                                                // registers did not fit beside the accumulators and the prefetched U
    for (int it = 0; it < 2; ++it) {
      float2 d[8], tt[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = *reinterpret_cast<const float2*>(wsm + a_t2l + it * 8192 + i * 2048);
      w77_bt(d, tt);
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) *reinterpret_cast<float2*>(wsm + a_t2s + it * 256 + xi * 8192) = tt[xi];
    }
  };

  // ---- a tile's first requests: raw images of chunks 0-2 (DMA), then the U fragments of chunk 0 (younger than the DMA: the compiler's wait
  //      for them is also the wait for the images) ----
  UT U[8];
  float2 bias_req = make_float2(0.f, 0.f);                    // the bias pair of this thread's epilogue item, requested WITH the tile (a load inside
  auto request_tile = [&]() {                                  // the epilogue would be younger than the next tile's requests: its wait would drain them)
    issue_raw(0);
    if (nchunks > 1) issue_raw(1);
    if (nchunks > 2) issue_raw(2);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) load_u(0, j, U[j]);
    if (P.bias) bias_req = *reinterpret_cast<const float2*>(P.bias + n0 + 2 * (tid & (CO / 2 - 1)));
  };
  if (tile >= P.nblocks) return;
  set_tile(tile);
  request_tile();

  for (;;) {
  // ---- prologue of this tile: its requests were issued before the previous tile's epilogue (or just above); X and V of chunk 0 ----
#pragma unroll
  for (int j = 0; j < 8; ++j) {                                // (the compiler's wait for the U loads lands here)
    if constexpr (C16) asm volatile("" :: "v"(U[j].x), "v"(U[j].y));
    else asm volatile("" :: "v"(U[j].x), "v"(U[j].y), "v"(U[j].z), "v"(U[j].w));
  }
  const float2 bias = bias_req;
  asm volatile("" :: "v"(bias.x), "v"(bias.y) : "memory");
  __syncthreads();
  {
    float2 d[8];
    if (wave < 7) { t1_load(0, d); t1_store(d); }
    __syncthreads();
    t2_run();
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < AR; ++r) acc[j][r] = 0.f;

  // ---- main loop: iteration c multiplies chunk c (V, U in registers), runs T1 of chunk c + 1 between the MFMAs (raw image (c + 1) % 3 -> X),
  //      fetches the U fragments of chunk c + 1 and starts the DMA of raw image c + 3; barrier; T2 of chunk c + 1 (X -> V); barrier ----
  const int a_v = W77_V_OFF + (8 * wave) * 1024 + lane * 16;  // V[p][lh][li]
  int rb1 = 1;                                               // ring slot of raw image c + 1; image c + 3 goes to the slot of image c = (rb1 + 2) % 3
  int e_img = 0, e_oy0 = 0, e_ox0 = 0, e_n0 = 0;              // this tile's coordinates, for its epilogue
  bool have_next = false;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    int cn = c + 1;
    if (!more) {
      // The LAST chunk's iteration belongs to the next tile as far as memory is concerned: the U prefetch "of chunk c + 1" fetches the next
      // tile's chunk 0 and the DMA window starts its raw images 0-2 (the ring is free: the last T1 ran an iteration ago) -- nothing of the
      // next tile waits behind this tile's epilogue, and no stale U re-read stands between the compiler's waits and the DMA.
      e_img = img; e_oy0 = oy0; e_ox0 = ox0; e_n0 = n0;
      tile += tile_step;
      have_next = tile < P.nblocks;                           // (uniform)
      if (have_next) set_tile(tile);                          // img / oy0 / ox0 / n0 / r_off / src_img describe the NEXT tile from here on
      cn = have_next ? 0 : c;                                 // (no next tile: the loads re-read this chunk's U -- no branch around them)
    }
    const char* vb = wsm + a_v;
    // position j, k-steps k0 .. k1 - 1 of 4.  CO = 32: the V fragment (a float4: lane = (block, quad)) is read here, element k is k-step k.
    // CO = 16: k-steps 2 q, 2 q + 1 stand for quad q: one 16x16x4 MFMA per block half, the lane's A value is channel lane / 16 of the quad.
    auto mma = [&](int j, int k0, int k1) {
      if constexpr (C16) {
        const char* v16 = wsm + W77_V_OFF + (8 * wave + j) * 1024 + (lane & 15) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int q = k0 / 2; q < k1 / 2; ++q) {
          const float a0 = *reinterpret_cast<const float*>(v16 + q * 512), a1 = *reinterpret_cast<const float*>(v16 + q * 512 + 256);
          const float u = q == 0 ? U[j].x : U[j].y;
          acc[2 * j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, u, acc[2 * j], 0, 0, 0);
          acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, u, acc[2 * j + 1], 0, 0, 0);
        }
      } else {
        const float4 vf = *reinterpret_cast<const float4*>(vb + j * 1024);
        const float v[4] = {vf.x, vf.y, vf.z, vf.w};
        const float u[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
#pragma unroll
        for (int k = k0; k < k1; ++k) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[k], u[k], acc[j], 0, 0, 0);
      }
    };
    float2 d[8];
    const bool do_t1 = more && wave < 7;
    if (do_t1) t1_load(rb1, d);
    mma(0, 0, 4);
    load_u(cn, 0, U[0]);
    mma(1, 0, 4);
    load_u(cn, 1, U[1]);
    if (do_t1) t1_store(d);
    mma(2, 0, 4);
    load_u(cn, 2, U[2]);
    mma(3, 0, 4);
    load_u(cn, 3, U[3]);
    mma(4, 0, 4);
    load_u(cn, 4, U[4]);
    mma(5, 0, 4);
    load_u(cn, 5, U[5]);
    mma(6, 0, 4);
    load_u(cn, 6, U[6]);
    mma(7, 0, 2);
    // the DMA goes out behind the last wait for this chunk's U and ahead of the load of U[7] (the compiler's wait for that load, one iteration
    // later, is then also the wait for the DMA); the two waves of a SIMD issue it at different points of the window
    const bool dma = c + 3 < nchunks;
    const bool roll = !more && have_next;                     // (uniform) the next tile's raw images 0-2
    const int role = wave >> 2, rbi = rb1 == 0 ? 2 : rb1 - 1;
    __builtin_amdgcn_sched_barrier(0);
    if (role == 0) { if (dma) issue_raw(rbi); else if (roll) { issue_raw(0); if (nchunks > 1) issue_raw(1); if (nchunks > 2) issue_raw(2); } }
    __builtin_amdgcn_sched_barrier(0);
    mma(7, 2, 4);
    __builtin_amdgcn_sched_barrier(0);
    if (role == 1) { if (dma) issue_raw(rbi); else if (roll) { issue_raw(0); if (nchunks > 1) issue_raw(1); if (nchunks > 2) issue_raw(2); } }
    asm volatile("" ::: "memory");
    load_u(cn, 7, U[7]);
    if (roll && P.bias) bias_req = *reinterpret_cast<const float2*>(P.bias + n0 + 2 * (tid & (CO / 2 - 1)));
    rb1 = rb1 == 2 ? 0 : rb1 + 1;
    __syncthreads();
    if (more) t2_run();
    __syncthreads();
  }

  // ---- epilogue, first part: A^T over nu on the wave's own accumulators (registers, packed on register pairs): the two output columns land in
  //      the accumulators of nu = 0, 1 ----
  constexpr int NH = C16 ? 2 : 1;                              // accumulator tiles per position
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int r = 0; r < AR; r += 2) {
      float2 m[8];
#pragma unroll
      for (int nu = 0; nu < 8; ++nu) m[nu] = make_float2(acc[NH * nu + hf][r], acc[NH * nu + hf][r + 1]);
      float2 y0, y1;
      w77_at(m, y0, y1);
      acc[hf][r] = y0.x; acc[hf][r + 1] = y0.y; acc[NH + hf][r] = y1.x; acc[NH + hf][r + 1] = y1.y;
    }
  // ---- exchange E[xi][column j][block][cout] (CO floats per line), then an item = one block x one cout PAIR ----
  float* E = reinterpret_cast<float*>(wsm + W77_V_OFF);       // (every wave is past the loop's last barrier: V is free)
  if constexpr (C16) {
    float* ew = E + ((2 * wave) * 32 + 4 * (lane >> 4)) * CO + (lane & 15);      // register r of half hf = block 16 hf + 4 (lane / 16) + r, lane % 16 = cout
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) ew[(j * 32 + 16 * hf + r) * CO] = acc[NH * j + hf][r];
  } else {
    float* ew = E + ((2 * wave) * 32 + 4 * lh) * CO + li;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(j * 32 + (r & 3) + 8 * (r >> 2)) * CO] = acc[j][r];    // register r = block (r & 3) + 8 (r >> 2) + 4 lh
  }
  __syncthreads();
  if (tid < 32 * (CO / 2)) {
    const int e_cp = tid & (CO / 2 - 1), e_b = tid / (CO / 2);  // item: (cout pair, block)
    const int e_br = e_b >> 3, e_bc = e_b & 7;
    const int ch = e_n0 + 2 * e_cp;
    const float slope = P.act == GPEMSR_ACT_LRELU ? 0.1f : 0.f;
    const bool has_act = P.act != GPEMSR_ACT_NONE;
    const float* er = E + e_b * CO + 2 * e_cp;
    const int oy = e_oy0 + 2 * e_br, ox = e_ox0 + 2 * e_bc;
    float* op = P.out + (((long long)e_img * P.h + oy) * P.w + ox) * P.out_ld + ch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float2 m[8];
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) m[xi] = *reinterpret_cast<const float2*>(er + ((xi * 2 + j) * 32) * CO);
      m[1] = w77_add(m[1], bias);                              // the bias rides the transform: A^T e_1 = (1, 1)
      float2 y[2];
      w77_at(m, y[0], y[1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float2 v = y[i];
        if (has_act) v = make_float2(fmaxf(v.x, slope * v.x) + 0.f, fmaxf(v.y, slope * v.y) + 0.f);
        if (oy + i < P.h && ox + j < P.w) *reinterpret_cast<float2*>(op + ((long long)i * P.w + j) * P.out_ld) = v;
      }
    }
  }
  if (!have_next) break;
  }   // persistent tile loop (the barrier at the top of the next tile's prologue also separates the items' E reads from its T2)
}

// descriptor.transposed == 6: called from gpemsr_conv2d (conv_mfma.hip)
int conv2d_winograd77(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap) {
  GP_REQUIRE(d->ksize == 7 && d->stride == 1 && d->weight_image_stride == 0 && d->nsrc == 1, "conv2d (F(2x2,7x7) form): 7x7, stride 1, one source, one weight set");
  GP_REQUIRE(!d->residual && !d->pixmul && !d->pixel_shuffle && !d->gn_partials && !d->cos_partials && !d->a_scale, "conv2d (F(2x2,7x7) form): plain store only");
  GP_REQUIRE(d->act == GPEMSR_ACT_NONE || d->act == GPEMSR_ACT_RELU || d->act == GPEMSR_ACT_LRELU, "conv2d (F(2x2,7x7) form): act NONE / RELU / LRELU (got %d)", d->act);
  GP_REQUIRE(d->cout % 16 == 0 && d->src[0].c % 8 == 0 && d->src[0].ld % 4 == 0 && d->src[0].ld >= d->src[0].c &&
             (reinterpret_cast<uintptr_t>(d->src[0].ptr) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->weight) & 15) == 0,
             "conv2d (F(2x2,7x7) form): cout %% 16 == 0, source c %% 8 == 0 with 16-byte aligned rows");
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->out) & 7) == 0 && d->out_ld % 2 == 0 && (!d->bias || (reinterpret_cast<uintptr_t>(d->bias) & 7) == 0),
             "conv2d (F(2x2,7x7) form): 8-byte aligned output rows and bias");
  const int co = d->cout % 32 == 0 ? 32 : 16;                  // couts per workgroup (16: the 16x16x4 MFMA form, no half-empty 32-cout tile)
  if (name_buf) { snprintf(name_buf, (size_t)name_cap, co == 32 ? "conv7_wino2d_f32_kernel<32>" : "conv7_wino2d_f32_kernel<16>"); return GPEMSR_OK; }
  W77Params P{};
  P.src = d->src[0].ptr; P.ld = d->src[0].ld; P.cin = d->src[0].c;
  P.img_stride = d->src_image_stride[0] < 0 ? (long long)d->h * d->w * d->src[0].ld : d->src_image_stride[0];
  GP_REQUIRE(P.img_stride % 4 == 0 && (long long)d->h * d->w * P.ld * 4 < (1ll << 32), "conv2d (F(2x2,7x7) form): source too large / misaligned");
  P.n = d->n; P.h = d->h; P.w = d->w; P.cout = d->cout;
  P.weight = d->weight; P.bias = d->bias; P.act = d->act; P.out = d->out; P.out_ld = d->out_ld;
  P.tiles_x = cdiv(d->w, 16); P.tiles_y = cdiv(d->h, 8); P.tiles_n = d->cout / co;
  const long long nb = (long long)d->n * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d (F(2x2,7x7) form): grid too large");
  P.nblocks = (int)nb;
  static dev_once_t done{0};
  if (dev_once_begin(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_wino2d_f32_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, W77_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv7_wino2d_f32_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, W77_LDS) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "conv2d (F(2x2,7x7) form): cannot raise the dynamic LDS limit to %d bytes", W77_LDS);
    dev_once_done(done);
  }
  int grid = device_cus();                                     // persistent: one workgroup per CU, a multiple of 8 (one share per XCD)
  if ((long long)grid > nb) grid = (int)nb;
  grid = grid < 8 ? 8 : grid & ~7;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (co == 32) hipLaunchKernelGGL(conv7_wino2d_f32_kernel<32>, dim3(grid), dim3(W77_NT), W77_LDS, st, P);
  else hipLaunchKernelGGL(conv7_wino2d_f32_kernel<16>, dim3(grid), dim3(W77_NT), W77_LDS, st, P);
  return check_launch("conv7_wino2d_f32_kernel");
}

}  // namespace gpemsr
