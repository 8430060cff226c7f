// Weight gradient of the convolution family on the f32 matrix pipe (v_mfma_f32_32x32x2_f32):
//   dW[co][ci][ky][kx] = sum_{n,oy,ox} dZ[n][oy][ox][co] * X[n][oy*s + ky - p][ox*s + kx - p][ci]
// i.e. a GEMM whose K dimension is the pixel axis.  NHWC makes both operands "k-major": for one pixel the 32 couts
// (A) and the 32 cins (B) of an MFMA operand are contiguous, so a lane's operand register is ONE float read at
// [pixel(lane>>5)][channel(lane&31)] -- no transposes anywhere.
//
// One 256-thread workgroup owns a 64(co) x 64(ci) block of dW for ONE FILTER ROW (k taps; wave w: co half w&1, ci half
// w>>1; k accumulators of 32x32 = 48 VGPRs) and walks a slab of output tiles (TH rows x 32 pixels).  Per tile it stages
// the dZ tile and the X rows that filter row touches (TH rows x 31s+k cols) in LDS once; the k taps read them at shifted
// offsets.  Splitting the filter rows over workgroups instead of the pixel axis multiplies the parallelism of the small
// training-crop layers by k without multiplying the partial-sum traffic.  Slabs write partial dW images
// [slab][tap][co][ci]; a second kernel adds them in slab order into the OIHW gradient (+=): deterministic.
//
// Two flavours of the tile loop:
//   wgrad_dma_kernel  full 64x64 channel blocks with 16-B aligned rows (the large layers): LDS-DMA into a double buffer,
//                     tile t+1 in flight while tile t is multiplied (see its comment);
//   wgrad_kernel      everything else (1-, 2-, 16-, 34-channel ends, unaligned slices): register-staged; LDS rows are only
//                     as wide as the channel block really is (1..64 floats), a lane that reads past its row picks up
//                     finite-or-not garbage that lands in rows/cols of D which are never written back (D[i][j] depends
//                     on A row i and B col j only).
//
// ConvTranspose2d(k3,s2,p1,op1) weights [Cin][Cout][3][3] use the same routine with the roles swapped
// (x := dOut at 2h x 2w, dz := the layer input at h x w, stride 2): out(2iy-1+ky) <- in(iy) * W[ci][co][ky][kx].
#include "common.h"
#include <stdlib.h>

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
  const float* x; int x_ld; int cin;
  const float* dz; int dz_ld; int cout;
  int n, h, w, oh, ow, stride, pad, th;
  int tiles_x, tiles_y, total_tiles, tiles_per_slab;
  int cwx_max, cwz_max, hr, hc;     // LDS row widths (register flavour); rows / columns of the staged X image
  int x_vec, dz_vec;                 // float4 loads allowed (ld % 4 == 0, base 16-B aligned)
  float* part;
};

__device__ __forceinline__ float4 wg_load4(const float* row, int ch, int cmax, int vec) {
  if (vec && ch + 3 < cmax) return *reinterpret_cast<const float4*>(row + ch);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ch < cmax) v.x = row[ch];
  if (ch + 1 < cmax) v.y = row[ch + 1];
  if (ch + 2 < cmax) v.z = row[ch + 2];
  if (ch + 3 < cmax) v.w = row[ch + 3];
  return v;
}

template <int KS>
__global__ __launch_bounds__(256, 4) void wgrad_kernel(WgradParams P) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int cob = (wv & 1) * 32, cib = (wv >> 1) * 32;
  const int con = min(64, P.cout - co0), cin_b = min(64, P.cin - ci0);     // valid channels of this block
  const int cwz = (con + 3) & ~3, cwx = (cin_b + 3) & ~3;                  // LDS row widths of this block
  float* dzs = smem;
  float* xs = smem + P.th * 32 * P.cwz_max + 64;
  const bool active = cob < con && cib < cin_b;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int slab = blockIdx.x / KS, ky = blockIdx.x % KS;                  // this workgroup's filter row
  const int t0 = slab * P.tiles_per_slab;
  const int t1 = min(t0 + P.tiles_per_slab, P.total_tiles);
  const int cz4 = cwz >> 2, cx4 = cwx >> 2;
  const int npx = P.th * 32, nhp = P.th * P.hc;
  for (int t = t0; t < t1; ++t) {
    const int tx = t % P.tiles_x, ty = (t / P.tiles_x) % P.tiles_y, img = t / (P.tiles_x * P.tiles_y);
    const int oy0 = ty * P.th, ox0 = tx * 32;
    __syncthreads();                       // the previous tile's reads are done
    for (int e = tid; e < npx * cz4; e += 256) {
      const int c4 = e % cz4, px = e / cz4;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < P.oh && ox < P.ow)
        v = wg_load4(P.dz + (((long long)img * P.oh + oy) * P.ow + ox) * P.dz_ld, co0 + 4 * c4, P.cout, P.dz_vec);
      *reinterpret_cast<float4*>(dzs + px * cwz + 4 * c4) = v;
    }
    const int ix0 = ox0 * P.stride - P.pad;
    for (int e = tid; e < nhp * cx4; e += 256) {
      const int c4 = e % cx4, hp = e / cx4;
      const int iy = (oy0 + hp / P.hc) * P.stride - P.pad + ky, ix = ix0 + hp % P.hc;      // LDS row r <-> output row oy0 + r
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w)
        v = wg_load4(P.x + (((long long)img * P.h + iy) * P.w + ix) * P.x_ld, ci0 + 4 * c4, P.cin, P.x_vec);
      *reinterpret_cast<float4*>(xs + hp * cwx + 4 * c4) = v;
    }
    __syncthreads();
    if (active) {
      const float* ap = dzs + lh * cwz + cob + l31;
      const float* bp = xs + lh * P.stride * cwx + cib + l31;
      for (int r = 0; r < P.th; ++r) {
        const float* ar = ap + r * 32 * cwz;
        const float* br = bp + r * P.hc * cwx;
#pragma unroll 4
        for (int cp = 0; cp < 16; ++cp) {
          const float a = ar[2 * cp * cwz];
          const float* bq = br + 2 * cp * P.stride * cwx;
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[kx * cwx], acc[kx], 0, 0, 0);
        }
      }
    }
  }
  if (active) {
    const int ci = ci0 + cib + l31;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cob + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < P.cout && ci < P.cin)
          P.part[(((long long)slab * KS * KS + ky * KS + kx) * P.cout + co) * P.cin + ci] = acc[kx][r];
      }
  }
}


// ---- LDS-DMA flavour for full 64x64 channel blocks (the large layers) ---------------------------------------------------
// Same tiling, but the two images of a tile go global -> LDS directly (global_load_lds_dwordx4: 64 lanes x 16 B = 4 pixel
// rows of 256 B per wave-instruction, no VGPR round trip, no ds_write) into a double buffer: tile t+1 is in flight while
// tile t is multiplied; one s_waitcnt vmcnt(0) + barrier per tile.  Out-of-image slots are zero-filled with ds_write by
// the lane that would have fetched them.  The asm form is needed because hipcc drains a builtin LDS-DMA before the next
// ds_read (same finding as conv_mfma.hip).
__device__ __forceinline__ void wg_glds16(unsigned voff, const void* base, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned wg_lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgradParams P) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform (SGPR): LDS piece addresses are scalar operands
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int cob = (wv & 1) * 32, cib = (wv >> 1) * 32;
  const int nzp = P.th * 32;                              // dz pixels per tile
  const int nxv = P.th * P.hc;                            // x pixels per tile ...
  const int nxp = (nxv + 3) & ~3;                         // ... padded to whole 1-KiB pieces
  const int buf_floats = (nzp + nxp) * 64 + 64;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int slab = blockIdx.x / KS, ky = blockIdx.x % KS;
  const int t0 = slab * P.tiles_per_slab;
  const int t1 = min(t0 + P.tiles_per_slab, P.total_tiles);
  const int sub = lane >> 4, c4 = lane & 15;              // pixel within a 4-pixel piece, float4 column
  // Per-lane piece geometry is tile-invariant: precompute it once (f32 MFMA shares the vector ALUs, so per-tile address
  // arithmetic costs matrix-pipe time).  This wave owns pieces wv, wv+4, ... of both images: <= 4 of dZ, <= 5 of X.
  constexpr int MZ = 4, MX = 5;
  const int nzpc = nzp / 4, nxpc = nxp / 4;
  int z_dy[MZ], z_dx[MZ], x_r[MX], x_c[MX];
  unsigned z_lds[MZ], x_lds[MX];                          // byte offsets of the piece inside a buffer (wave-uniform)
#pragma unroll
  for (int i = 0; i < MZ; ++i) {
    const int px = (wv + 4 * i) * 4 + sub;
    z_dy[i] = px >> 5; z_dx[i] = px & 31;
    z_lds[i] = (unsigned)(wv + 4 * i) * 1024u;
  }
#pragma unroll
  for (int i = 0; i < MX; ++i) {
    const int hp = (wv + 4 * i) * 4 + sub;
    x_r[i] = hp < nxv ? hp / P.hc : -(1 << 20);           // padded tail of the last piece: never inside the image
    x_c[i] = hp % P.hc;
    x_lds[i] = (unsigned)(nzp * 256) + (unsigned)(wv + 4 * i) * 1024u;
  }
  const unsigned zcol = (unsigned)(co0 + 4 * c4) * 4u, xcol = (unsigned)(ci0 + 4 * c4) * 4u;

  auto issue = [&](int t, float* buf) {
    const int tx = t % P.tiles_x, ty = (t / P.tiles_x) % P.tiles_y, img = t / (P.tiles_x * P.tiles_y);
    const int oy0 = ty * P.th, ox0 = tx * 32;
    const float* zbase = P.dz + (long long)img * P.oh * P.ow * P.dz_ld;
    const float* xbase = P.x + (long long)img * P.h * P.w * P.x_ld;
    const unsigned lbase = wg_lds_addr(buf);
#pragma unroll
    for (int i = 0; i < MZ; ++i) {
      if (wv + 4 * i < nzpc) {
        const int oy = oy0 + z_dy[i], ox = ox0 + z_dx[i];
        if (oy < P.oh && ox < P.ow)
          wg_glds16((unsigned)((oy * P.ow + ox) * P.dz_ld) * 4u + zcol, zbase, __builtin_amdgcn_readfirstlane(lbase + z_lds[i]));
        else
          *reinterpret_cast<float4*>(buf + (z_lds[i] >> 2) + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int iyb = oy0 * P.stride - P.pad + ky, ix0 = ox0 * P.stride - P.pad;
#pragma unroll
    for (int i = 0; i < MX; ++i) {
      if (wv + 4 * i < nxpc) {
        const int iy = iyb + x_r[i] * P.stride, ix = ix0 + x_c[i];
        if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w)
          wg_glds16((unsigned)((iy * P.w + ix) * P.x_ld) * 4u + xcol, xbase, __builtin_amdgcn_readfirstlane(lbase + x_lds[i]));
        else
          *reinterpret_cast<float4*>(buf + (x_lds[i] >> 2) + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  if (t0 < t1) issue(t0, smem);
  for (int t = t0; t < t1; ++t) {
    float* cur = smem + ((t - t0) & 1) * buf_floats;
    float* nxt = smem + (((t - t0) & 1) ^ 1) * buf_floats;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // ... and its zero-fill ds_writes
    __builtin_amdgcn_s_barrier();                         // everybody's pieces are visible; everybody left the other buffer
    asm volatile("" ::: "memory");
    if (t + 1 < t1) issue(t + 1, nxt);                    // in flight while tile t is multiplied
    const float* ap = cur + lh * 64 + cob + l31;
    const float* bp = cur + nzp * 64 + lh * P.stride * 64 + cib + l31;
    for (int r = 0; r < P.th; ++r) {
      const float* ar = ap + r * 32 * 64;
      const float* br = bp + r * P.hc * 64;
#pragma unroll 4
      for (int cp = 0; cp < 16; ++cp) {
        const float a = ar[2 * cp * 64];
        const float* bq = br + 2 * cp * P.stride * 64;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[kx * 64], acc[kx], 0, 0, 0);
      }
    }
  }
  const int ci = ci0 + cib + l31;
#pragma unroll
  for (int kx = 0; kx < KS; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cob + (r & 3) + 8 * (r >> 2) + 4 * lh;
      P.part[(((long long)slab * KS * KS + ky * KS + kx) * P.cout + co) * P.cin + ci] = acc[kx][r];
    }
}

// dw[(co*cin_total + cin_off + ci)*taps + tap] += sum_slab part[slab][tap][co][ci].  64 consecutive elements per block, four
// slab lanes each (lane q sums slabs q, q+4, ...), folded in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, int slabs, int taps, int cout, int cin, float* dw,
                                                           int cin_total, int cin_off) {
  const long long total = (long long)taps * cout * cin;
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  __shared__ float red[4][64];
  for (long long e0 = (long long)blockIdx.x * 64; e0 < total; e0 += (long long)gridDim.x * 64) {
    const long long e = e0 + el;
    float s = 0.f;
    if (e < total)
      for (int sl = q; sl < slabs; sl += 4) s += part[(long long)sl * total + e];
    red[q][el] = s;
    __syncthreads();
    if (q == 0 && e < total) {
      const int ci = (int)(e % cin);
      const int co = (int)((e / cin) % cout);
      const int tap = (int)(e / ((long long)cin * cout));
      dw[((long long)co * cin_total + cin_off + ci) * taps + tap] += (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    }
    __syncthreads();
  }
}

}  // namespace gpemsr

using namespace gpemsr;

// Slabs of the pixel axis.  `slots` = workgroups the chip holds at once (CUs x workgroups per CU).  A launch of slots+16
// workgroups takes two rounds (measured: 74 vs 107 TFLOP/s on 256->256 @128^2), so the workgroup count (slabs x filter rows
// x channel blocks) is fitted to one or two FULL rounds, whichever wastes less; more slabs only multiply the partial sums.
static long long wgrad_slabs(long long tiles, int cin, int cout, int ksize, long long* tps_out, long long slots = 1024) {
  static const long long forced = getenv("GPEMSR_WGRAD_WGS") ? atoll(getenv("GPEMSR_WGRAD_WGS")) : 0;
  if (forced > 0) slots = forced;
  const long long blocks = (long long)((cout + 63) / 64) * ((cin + 63) / 64) * ksize;
  long long want = slots / blocks;                              // one round
  const long long want2 = 2 * slots / blocks;                   // two rounds
  if (want < 1 || (want2 >= 1 && want2 * blocks * 1.0 / (2 * slots) > want * blocks * 1.0 / slots + 0.05)) want = want2;
  if (want < 1) want = 1;
  long long tps = (tiles + want - 1) / want;
  if (tps < 2) tps = 2;
  *tps_out = tps;
  return (tiles + tps - 1) / tps;
}

extern "C" int64_t gpemsr_conv2d_wgrad_workspace(int cin, int cout, int ksize, int n, int oh, int ow) {
  const int th = 2;
  const long long tiles = (long long)n * ((oh + th - 1) / th) * ((ow + 31) / 32);
  long long tps;
  return wgrad_slabs(tiles, cin, cout, ksize, &tps) * ksize * ksize * cout * cin;
}

extern "C" int gpemsr_conv2d_wgrad(const float* x, int x_ld, int cin, const float* dz, int dz_ld, int cout, int n, int h, int w,
                                   int oh, int ow, int ksize, int stride, float* ws, int64_t ws_floats, float* dw, int cin_total,
                                   int cin_off, void* stream) {
  GP_REQUIRE(x && dz && ws && dw && n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv2d_wgrad: bad args");
  GP_REQUIRE(ksize == 1 || ksize == 3, "conv2d_wgrad: ksize %d unsupported (1 or 3: every trainable convolution of the stage-3 network)", ksize);
  GP_REQUIRE(stride == 1 || stride == 2 || stride == 4, "conv2d_wgrad: stride %d unsupported", stride);
  const int pad = ksize / 2;
  GP_REQUIRE(oh == (h + 2 * pad - ksize) / stride + 1 && ow == (w + 2 * pad - ksize) / stride + 1,
             "conv2d_wgrad: dz geometry %dx%d does not match x %dx%d (k%d s%d)", oh, ow, h, w, ksize, stride);
  GP_REQUIRE(cin_off >= 0 && cin_off + cin <= cin_total && x_ld >= cin && dz_ld >= cout, "conv2d_wgrad: channel ranges");
  WgradParams P;
  P.x = x; P.x_ld = x_ld; P.cin = cin; P.dz = dz; P.dz_ld = dz_ld; P.cout = cout;
  P.n = n; P.h = h; P.w = w; P.oh = oh; P.ow = ow; P.stride = stride; P.pad = pad;
  P.cwx_max = cin >= 64 ? 64 : (cin + 3) & ~3;
  P.cwz_max = cout >= 64 ? 64 : (cout + 3) & ~3;
  P.th = 2;
  P.hc = 31 * stride + ksize;
  auto lds_bytes = [&](int th) { return (size_t)(th * 32 * P.cwz_max + 64 + th * P.hc * P.cwx_max + 64) * 4; };
  if (lds_bytes(P.th) > 160 * 1024) P.th = 1;
  GP_REQUIRE(lds_bytes(P.th) <= 160 * 1024, "conv2d_wgrad: tile does not fit LDS");
  P.hr = P.th;
  P.tiles_x = (ow + 31) / 32; P.tiles_y = (oh + P.th - 1) / P.th;
  const long long tiles = (long long)n * P.tiles_x * P.tiles_y;
  GP_REQUIRE(tiles < (1ll << 31), "conv2d_wgrad: too many tiles");
  P.total_tiles = (int)tiles;
  const long long per_slab = (long long)ksize * ksize * cout * cin;
  long long tps;
  long long slabs = wgrad_slabs(tiles, cin, cout, ksize, &tps);
  if (slabs * per_slab > ws_floats) {                        // fewer, longer slabs if the workspace is small
    slabs = ws_floats / per_slab;
    GP_REQUIRE(slabs >= 1, "conv2d_wgrad: workspace too small (need >= %lld floats)", per_slab);
    tps = (tiles + slabs - 1) / slabs;
    slabs = (tiles + tps - 1) / tps;
  }
  P.tiles_per_slab = (int)tps;
  P.x_vec = (x_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  P.dz_vec = (dz_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0);
  P.part = ws;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const bool wgrad_no_dma = getenv("GPEMSR_WGRAD_NO_DMA") != nullptr;       // read once per process
  // LDS-DMA flavour: full 64x64 channel blocks, 16-B aligned rows, 32-bit byte offsets inside one image
  const bool dma = (cout % 64 == 0) && (cin % 64 == 0) && P.x_vec && P.dz_vec && stride <= 2 &&
                   ((long long)h * w * x_ld * 4 < (1ll << 32)) && ((long long)oh * ow * dz_ld * 4 < (1ll << 32)) &&
                   ((((stride == 1 ? 2 : 1) * (31 * stride + ksize) + 3) / 4 + 3) / 4 <= 5) &&   // <= 5 X pieces per wave
                   !wgrad_no_dma;
  if (dma) {
    P.th = stride == 1 ? 2 : 1;
    P.hr = P.th;
    P.tiles_y = (oh + P.th - 1) / P.th;
    const long long tl = (long long)n * P.tiles_x * P.tiles_y;
    GP_REQUIRE(tl < (1ll << 31), "conv2d_wgrad: too many tiles");
    P.total_tiles = (int)tl;
    slabs = wgrad_slabs(tl, cin, cout, ksize, &tps, 512);          // 67.6 KB of LDS per workgroup: 2 per CU
    if (slabs * per_slab > ws_floats) { slabs = ws_floats / per_slab; tps = (tl + slabs - 1) / slabs; slabs = (tl + tps - 1) / tps; }
    P.tiles_per_slab = (int)tps;
    const size_t ldsd = 2 * (size_t)((P.th * 32 + ((P.th * P.hc + 3) & ~3)) * 64 + 64) * 4;
    const dim3 gridd((unsigned)(slabs * ksize), (unsigned)(cout / 64), (unsigned)(cin / 64));
    if (ksize == 3) {
      static dev_once_t a3{0};
      if (dev_once_begin(a3)) { hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); dev_once_done(a3); }
      hipLaunchKernelGGL(wgrad_dma_kernel<3>, gridd, dim3(256), ldsd, st, P);
    } else {
      static dev_once_t a1{0};
      if (dev_once_begin(a1)) { hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); dev_once_done(a1); }
      hipLaunchKernelGGL(wgrad_dma_kernel<1>, gridd, dim3(256), ldsd, st, P);
    }
  } else {
  const size_t lds = lds_bytes(P.th);
  const dim3 grid((unsigned)(slabs * ksize), (unsigned)((cout + 63) / 64), (unsigned)((cin + 63) / 64));
  if (ksize == 3) {
    static dev_once_t attr3{0};
    if (dev_once_begin(attr3)) { hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); dev_once_done(attr3); }
    hipLaunchKernelGGL(wgrad_kernel<3>, grid, dim3(256), lds, st, P);
  } else {
    static dev_once_t attr1{0};
    if (dev_once_begin(attr1)) { hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); dev_once_done(attr1); }
    hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), lds, st, P);
  }
  }
  const long long total = per_slab;
  const long long rb = (total + 63) / 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(rb < 32768 ? rb : 32768)), dim3(256), 0, st, ws, (int)slabs, ksize * ksize, cout, cin,
                     dw, cin_total, cin_off);
  return check_launch("conv2d_wgrad");
}
