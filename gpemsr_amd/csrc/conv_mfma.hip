// Implicit-GEMM convolution on the CDNA4 f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// One workgroup (4 waves) produces an 8x16-pixel x BN-channel output tile:
//   D[pixel][cout] = sum_{tap} sum_{cin} In[pixel*stride + tap][cin] * W[tap][cout][cin]
// The K loop runs over (cin chunk of CK channels) x (tap group).  Per chunk the
// input halo tile ((8-1)*s+k) x ((16-1)*s+k) x CK is staged ONCE in LDS and every
// tap of the filter reads it at a shifted offset (k*k-fold reuse out of LDS, no
// im2col in HBM).  The weight slice [taps][BN][CK] of the chunk is staged beside
// it.  Global loads of stage s+1 are issued into registers before the MFMAs of
// stage s, and written to LDS after them (register-staged pipeline).
//
// MFMA operand maps (guide section 3): A[i = lane&31][k = lane>>5], B[k][j = lane&31],
// D col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel).  A lane
// reads one float4 of A and B per tap: element e is k-step e, in which lane half h
// supplies channel 4h+e of the chunk -- A and B agree, so the k order is free.
//
// The same kernel evaluates ConvTranspose2d(k3,s2,p1,op1) as four sub-pixel phases
// (blockIdx carries the phase; phase (py,px) has (1+py)*(1+px) taps), fuses
// PixelShuffle(2) into the store, reads a virtual channel-concat of up to four
// sources, and serves torch.bmm / nn.Linear as 1x1 convolutions with per-image
// weights.  Replaces the ATen conv/bmm calls under model/GPEMSR.py:323-456.
#include "common.h"

namespace gpemsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE_H = 8, TILE_W = 16;
constexpr int MAX_TAPS = 49;
constexpr int A_LOADS = 5;   // float4 per thread per A stage (561 halo px * 2 / 256 -> 5)

struct ConvParams {
  const float* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int vec[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w;            // input geometry
  int oh, ow;             // conv grid (per phase for transposed)
  int OH, OW;             // stored output geometry
  int cin_pad, cout;
  int ksize, stride, pad, transposed;
  const float* weight; long long w_img_stride;
  const float* bias; int act;
  const float* residual; int res_ld;
  const float* pixmul;
  int pixel_shuffle;
  float* out; int out_ld;
  int tiles_x, tiles_y, tiles_n, nphase;
  int halo_h, halo_w;
  int ntaps, taps_per_group, ngroups;
  int nblocks;
};

template <int CK, int BN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams P) {
  constexpr int WN = (BN >= 64) ? 2 : 1;
  constexpr int WM = 4 / WN;
  constexpr int PM = 128 / WM;        // pixels per wave
  constexpr int MT = PM / 32;
  constexpr int WNT = BN / WN;        // couts per wave
  constexpr int NT = WNT / 32;
  constexpr int APIX = (CK == 8) ? 8 : CK + 4;   // floats per halo pixel in LDS
  constexpr int BPIX = APIX;
  constexpr int V4 = CK / 4;          // float4 per pixel per chunk
  constexpr int KJ = CK / 8;          // 8-channel groups per chunk
  constexpr int B_LOADS = (CK == 8) ? (9 * BN * V4 + 255) / 256 : (BN * V4 + 255) / 256;  // CK=32 is 1x1 only

  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* tap_dy = reinterpret_cast<int*>(smem);
  int* tap_dx = tap_dy + 64;
  int* tap_w = tap_dx + 64;
  float* As = smem + 192;
  float* Bs = As + ((P.halo_h * P.halo_w * APIX + 3) & ~3);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- XCD-aware block id remap (bijective; consecutive logical tiles share an XCD/L2) ----
  int bid = blockIdx.x;
  {
    const int nwg = P.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = bid;
  const int tn = t % P.tiles_n; t /= P.tiles_n;
  const int tx = t % P.tiles_x; t /= P.tiles_x;
  const int ty = t % P.tiles_y; t /= P.tiles_y;
  const int phase = t % P.nphase; t /= P.nphase;
  const int img = t;
  const int py = phase >> 1, px = phase & 1;
  const int oy0 = ty * TILE_H, ox0 = tx * TILE_W, n0 = tn * BN;

  // ---- tap table: (dy,dx) in halo coordinates, weight tap index ----
  int ntaps = P.ntaps;
  if (P.transposed) ntaps = (1 + py) * (1 + px);
  if (tid < MAX_TAPS) {
    int dy = 0, dx = 0, wt = 0;
    if (!P.transposed) {
      dy = tid / P.ksize; dx = tid % P.ksize; wt = tid;
    } else {
      // out(2i+py, 2j+px) = sum_{ky,kx} in(i+dy, j+dx) W[ky][kx] with 2*dy = py+1-ky
      const int ay = tid / (1 + px), ax = tid % (1 + px);
      const int ky = py ? (ay ? 2 : 0) : 1, kx = px ? (ax ? 2 : 0) : 1;
      dy = py ? (ay ? 0 : 1) : 0; dx = px ? (ax ? 0 : 1) : 0;
      wt = ky * 3 + kx;
    }
    tap_dy[tid] = dy; tap_dx[tid] = dx; tap_w[tid] = wt;
  }
  const int S = P.transposed ? 1 : P.stride;
  const int iy0 = P.transposed ? oy0 : oy0 * S - P.pad;
  const int ix0 = P.transposed ? ox0 : ox0 * S - P.pad;
  const int halo_px = P.halo_h * P.halo_w;

  // ---- per-thread A staging slots (independent of the stage) ----
  int a_pix[A_LOADS];      // pixel index inside the image, or -1
  int a_lds[A_LOADS];      // LDS float offset, or -1
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int e = tid + i * 256;
    a_pix[i] = -1; a_lds[i] = -1;
    if (e < halo_px * V4) {
      const int hp = e / V4, j = e % V4;
      const int hy = hp / P.halo_w, hx = hp % P.halo_w;
      const int iy = iy0 + hy, ix = ix0 + hx;
      a_lds[i] = hp * APIX + 4 * j;
      if (iy >= 0 && iy < P.h && ix >= 0 && ix < P.w) a_pix[i] = iy * P.w + ix;
    }
  }
  // per-lane A fragment pixel offsets (floats) for each M tile
  int a_frag[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = wm * PM + mt * 32 + li;
    a_frag[mt] = (((p >> 4) * S) * P.halo_w + (p & 15) * S) * APIX + 4 * lh;
  }
  const int b_frag = (wn * WNT + li) * BPIX + 4 * lh;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- stage enumeration: chunk-major, tap-group-minor ----
  int nchunks = 0;
  for (int s = 0; s < P.nsrc; ++s) nchunks += (P.c[s] + CK - 1) / CK;
  const int ngroups = P.transposed ? 1 : P.ngroups;
  const int tpg = P.transposed ? ntaps : P.taps_per_group;
  const int nstages = nchunks * ngroups;
  const float* wbase = P.weight + (long long)img * P.w_img_stride;

  float4 ra[A_LOADS];
  float4 rb[B_LOADS];

  auto prefetch = [&](int stage) {
    const int chunk = stage / ngroups, grp = stage % ngroups;
    int s = 0, c0 = chunk * CK, cpad = 0;
    while (s < P.nsrc - 1 && c0 >= ((P.c[s] + CK - 1) / CK) * CK) {
      const int cp = ((P.c[s] + CK - 1) / CK) * CK;
      c0 -= cp; cpad += cp; ++s;
    }
    if (grp == 0) {
      const float* sp = P.src[s] + (long long)img * P.img_stride[s];
      const int ld = P.ld[s], cs = P.c[s];
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_pix[i] >= 0) {
          const int j = (a_lds[i] % APIX) >> 2;
          const int cc = c0 + 4 * j;
          const float* gp = sp + (long long)a_pix[i] * ld + cc;
          if (P.vec[s] && cc + 3 < cs) {
            v = *reinterpret_cast<const float4*>(gp);
          } else {
            if (cc + 0 < cs) v.x = gp[0];
            if (cc + 1 < cs) v.y = gp[1];
            if (cc + 2 < cs) v.z = gp[2];
            if (cc + 3 < cs) v.w = gp[3];
          }
        }
        ra[i] = v;
      }
    }
    const int nb4 = tpg * BN * V4;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int e = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < nb4) {
        const int tt = e / (BN * V4), rem = e % (BN * V4);
        const int nn = rem / V4, j = rem % V4;
        if (n0 + nn < P.cout) {
          const int wt = tap_w[grp * tpg + tt];
          v = *reinterpret_cast<const float4*>(wbase + ((long long)wt * P.cout + n0 + nn) * P.cin_pad + cpad + c0 + 4 * j);
        }
      }
      rb[i] = v;
    }
  };
  auto commit = [&](int stage) {
    const int grp = stage % ngroups;
    if (grp == 0) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        if (a_lds[i] >= 0) *reinterpret_cast<float4*>(As + a_lds[i]) = ra[i];
    }
    const int nb4 = tpg * BN * V4;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int e = tid + i * 256;
      if (e < nb4) {
        const int tt = e / (BN * V4), rem = e % (BN * V4);
        const int nn = rem / V4, j = rem % V4;
        *reinterpret_cast<float4*>(Bs + (tt * BN + nn) * BPIX + 4 * j) = rb[i];
      }
    }
  };

  __syncthreads();   // tap table visible
  prefetch(0);
  for (int stage = 0; stage < nstages; ++stage) {
#ifdef GP_EXP_NO_RESTAGE
    if (stage == 0) {
#endif
    __syncthreads();           // everyone done reading the previous stage
    commit(stage);
    __syncthreads();
    if (stage + 1 < nstages) prefetch(stage + 1);
#ifdef GP_EXP_NO_RESTAGE
    }
#endif
    const int grp = stage % ngroups;
    for (int tt = 0; tt < tpg; ++tt) {
      const int tg = grp * tpg + tt;
      const int aoff = (tap_dy[tg] * P.halo_w + tap_dx[tg]) * APIX;
      const float* bp = Bs + tt * BN * BPIX + b_frag;
#pragma unroll
      for (int kj = 0; kj < KJ; ++kj) {
        float4 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(As + a_frag[mt] + aoff + 8 * kj);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(bp + nt * 32 * BPIX + 8 * kj);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: bias, activation, residual, pixel multiplier, (shuffled) store ----
  const int cq = P.cout >> 2;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int nidx = n0 + wn * WNT + nt * 32 + li;
    if (nidx >= P.cout) continue;
    const float bv = P.bias ? P.bias[nidx] : 0.f;
    int ch = nidx, sy = 0, sx = 0;
    if (P.pixel_shuffle) { const int q = nidx / cq; ch = nidx - q * cq; sy = q >> 1; sx = q & 1; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef GP_EXP_NO_EPILOGUE
        if (r != 0) { asm volatile("" ::"v"(acc[mt][nt][r])); continue; }
#endif
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int p = wm * PM + mt * 32 + row;
        const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
        if (oy >= P.oh || ox >= P.ow) continue;
        int Y = oy, X = ox;
        if (P.transposed) { Y = 2 * oy + py; X = 2 * ox + px; }
        else if (P.pixel_shuffle) { Y = 2 * oy + sy; X = 2 * ox + sx; }
        const long long opix = ((long long)img * P.OH + Y) * P.OW + X;
        float v = apply_act(acc[mt][nt][r] + bv, P.act);
        if (P.residual) v += P.residual[opix * P.res_ld + ch];
        if (P.pixmul) v *= P.pixmul[opix];
        P.out[opix * P.out_ld + ch] = v;
      }
    }
  }
}

template <int CK, int BN>
static int launch(const ConvParams& P, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL((conv_mfma_kernel<CK, BN>), dim3(P.nblocks), dim3(256), lds_bytes, st, P);
  return check_launch("conv_mfma_kernel");
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_conv2d(const gpemsr_conv_desc* d, void* stream) {
  GP_REQUIRE(d != nullptr, "conv2d: null descriptor");
  GP_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "conv2d: bad geometry n=%d h=%d w=%d cout=%d", d->n, d->h, d->w, d->cout);
  GP_REQUIRE(d->nsrc >= 1 && d->nsrc <= GPEMSR_MAX_SRC, "conv2d: nsrc=%d", d->nsrc);
  GP_REQUIRE(d->ksize == 1 || d->ksize == 3 || d->ksize == 7, "conv2d: ksize=%d unsupported", d->ksize);
  GP_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
  ConvParams P{};
  const bool tr = d->transposed != 0;
  if (tr) GP_REQUIRE(d->ksize == 3 && !d->pixel_shuffle, "conv2d: transposed needs k=3, no pixel_shuffle");
  else GP_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride=%d unsupported (use conv2d_direct)", d->stride);
  if (d->pixel_shuffle) GP_REQUIRE(d->cout % 4 == 0 && d->stride == 1, "conv2d: pixel_shuffle needs cout%%4==0, stride 1");
  const int CK = (d->ksize == 1) ? 32 : 8;
  int cin_pad = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    GP_REQUIRE(d->src[s].ptr && d->src[s].c > 0 && d->src[s].ld >= d->src[s].c, "conv2d: bad source %d", s);
    P.src[s] = d->src[s].ptr; P.ld[s] = d->src[s].ld; P.c[s] = d->src[s].c;
    P.img_stride[s] = d->src_image_stride[s] < 0 ? (long long)d->h * d->w * d->src[s].ld : d->src_image_stride[s];
    P.vec[s] = (d->src[s].ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) == 0) && (P.img_stride[s] % 4 == 0);
    cin_pad += ((d->src[s].c + CK - 1) / CK) * CK;
  }
  GP_REQUIRE((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0 && d->weight_image_stride % 4 == 0, "conv2d: weight must be 16B aligned");
  P.nsrc = d->nsrc; P.n = d->n; P.h = d->h; P.w = d->w;
  P.cin_pad = cin_pad; P.cout = d->cout; P.ksize = d->ksize;
  P.stride = tr ? 1 : d->stride; P.pad = d->ksize / 2; P.transposed = tr;
  P.weight = d->weight; P.w_img_stride = d->weight_image_stride; P.bias = d->bias; P.act = d->act;
  P.residual = d->residual; P.res_ld = d->res_ld; P.pixmul = d->pixmul; P.pixel_shuffle = d->pixel_shuffle;
  P.out = d->out; P.out_ld = d->out_ld;
  if (tr) { P.oh = d->h; P.ow = d->w; P.OH = 2 * d->h; P.OW = 2 * d->w; P.nphase = 4;
            P.halo_h = TILE_H + 1; P.halo_w = TILE_W + 1; P.ntaps = 4; P.taps_per_group = 4; P.ngroups = 1; }
  else {
    P.oh = (d->h + 2 * P.pad - d->ksize) / P.stride + 1;
    P.ow = (d->w + 2 * P.pad - d->ksize) / P.stride + 1;
    P.OH = d->pixel_shuffle ? 2 * P.oh : P.oh; P.OW = d->pixel_shuffle ? 2 * P.ow : P.ow; P.nphase = 1;
    P.halo_h = (TILE_H - 1) * P.stride + d->ksize; P.halo_w = (TILE_W - 1) * P.stride + d->ksize;
    P.ntaps = d->ksize * d->ksize;
    P.taps_per_group = d->ksize == 7 ? 7 : P.ntaps; P.ngroups = P.ntaps / P.taps_per_group;
  }
  const int BN = d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128);
  P.tiles_x = cdiv(P.ow, TILE_W); P.tiles_y = cdiv(P.oh, TILE_H); P.tiles_n = cdiv(d->cout, BN);
  const long long nb = (long long)d->n * P.nphase * P.tiles_y * P.tiles_x * P.tiles_n;
  GP_REQUIRE(nb > 0 && nb < (1ll << 31), "conv2d: grid too large");
  P.nblocks = (int)nb;
  const int APIX = CK == 8 ? 8 : CK + 4;
  GP_REQUIRE(P.halo_h * P.halo_w * (CK / 4) <= A_LOADS * 256, "conv2d: halo too large");
  const size_t a_floats = ((size_t)P.halo_h * P.halo_w * APIX + 3) & ~(size_t)3;
  const size_t b_floats = (size_t)P.taps_per_group * BN * APIX;
  const size_t lds = (192 + a_floats + b_floats) * sizeof(float);
  GP_REQUIRE(lds <= 160 * 1024 - 256, "conv2d: LDS %zu too large", lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (CK == 8) {
    if (BN == 32) return launch<8, 32>(P, lds, st);
    if (BN == 64) return launch<8, 64>(P, lds, st);
    return launch<8, 128>(P, lds, st);
  } else {
    if (BN == 32) return launch<32, 32>(P, lds, st);
    if (BN == 64) return launch<32, 64>(P, lds, st);
    return launch<32, 128>(P, lds, st);
  }
}

// The Python binding (gpemsr_amd/_abi.py) mirrors this struct field by field.
static_assert(sizeof(gpemsr_conv_desc) == 208, "gpemsr_conv_desc layout changed: update gpemsr_amd/_abi.py");
