// Modulated deformable convolution (DCNv2) of the bf16 data path in ONE kernel: the deformable sampling writes its columns into LDS and
// the 64 x 576 contraction reads them from there -- the [pixels][9 x 64] column tensor (1.5 GB per 128^2 level at batch 16) never exists in
// HBM, and neither does its round trip through a separate 1x1 product.
//
// Replaces, for bf16 tensors, basicsr `DCNv2Pack.forward` -> torchvision `deform_conv2d` as called from POD (R:model/GPEMSR.py:79-94,
// 112-138) AFTER its `conv_offset` convolution: x [n][h][w][64] bf16, om [n][h][w][216] fp32 (the raw conv_offset output: 144 sampling
// offsets, 72 mask logits; sampling coordinates keep fp32), weight rows [64 couts][9 taps x 64 channels] bf16, fp32 bias -> out bf16.
// Semantics: deformable group g = channels 8g..8g+7, tap k = (ky, kx): sample (y - 1 + ky + om[18g + 2k],
// x - 1 + kx + om[18g + 2k + 1]) bilinearly with zero contribution from corners outside the image, times sigmoid(om[144 + 9g + k]).
//
// One 512-thread workgroup per CU, persistent over tiles of 64 consecutive pixels.  LDS: the weight rows stay resident
// ([64][576 + 8 pad] bf16: 1,168-byte rows make every ds_read_b128 fragment read conflict-free), the tile's columns [64][576 + 8] beside
// them, an 8 KB output tile.  Per tile: (1) all 8 waves gather -- 72 (tap, group) items per pixel, 9 per thread, the four corner rows of
// an item are always fetched (clamped addresses, weight 0 outside: branch-free, 12 16-byte loads in flight per thread and batch);
// (2) barrier; waves 0-3 contract (32 pixels x 32 couts each, 36 x v_mfma_f32_32x32x16_bf16), add bias, activate, round to bf16 into the
// output tile; (3) barrier; every thread stores one 16-byte piece (a pixel's 64 channels are one 128-byte line) and starts the next gather.
// HBM-bound by construction: om (864 B / pixel) + x + out; the matrix work of a tile is ~1,200 cycles of one wave per SIMD.
#include "conv_bf16.h"

namespace gpemsr {

struct DcnParams {
  const unsigned short* x; const float* om; const unsigned short* wrows; const float* bias; unsigned short* out;
  long long npix; int h, w, x_ld, om_ld, out_ld, act, ntiles;
  unsigned mg_hw, mg_w;                        // floor((2^32 - 1) / (h w)), floor((2^32 - 1) / w): pixel index -> (image, y, x) by mulhi (xdivmod)
};

constexpr int DCN_ROW = 576 + 8;                 // bf16 elements per LDS row (1,168 B = 292 dwords: conflict-free 16-byte fragment reads)
constexpr int DCN_W_BYTES = 64 * DCN_ROW * 2;    // 74,752
constexpr int DCN_LDS = 2 * DCN_W_BYTES + 64 * 64 * 2 + 256;

__global__ __launch_bounds__(512, 1) void dcn_fused16_kernel(DcnParams P) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  char* const wl = dsm;                           // weight rows [cout][DCN_ROW]
  char* const cl = dsm + DCN_W_BYTES;             // column rows [pixel][DCN_ROW]
  char* const ol = dsm + 2 * DCN_W_BYTES;         // output tile [pixel][64] bf16
  float* const bl = reinterpret_cast<float*>(dsm + 2 * DCN_W_BYTES + 64 * 64 * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int e = tid; e < 64 * 72; e += 512) {      // resident weights: 72 16-byte pieces per cout row
    const int r = e / 72, q = e % 72;
    *reinterpret_cast<uint4*>(wl + r * (DCN_ROW * 2) + q * 16) = *reinterpret_cast<const uint4*>(P.wrows + (size_t)r * 576 + q * 8);
  }
  if (tid < 64) bl[tid] = P.bias ? P.bias[tid] : 0.f;

  const int hw = P.h * P.w;
  const float fh = (float)P.h, fw = (float)P.w;
  const unsigned cl_lds = xlds_addr(cl), wl_lds = xlds_addr(wl);
  const int li = lane & 31, lh = lane >> 5;

  // item i of this thread in a tile: e = tid + 512 i -> pixel e / 72, tap (e % 72) / 8, group e % 8 (lanes of a wave share one or two pixels)
  // The gather is a chain of two dependent round trips (offsets / mask logits, then the four corner rows they point at) and one workgroup
  // per CU has only 8 waves to hide them: the offsets of the NEXT tile are requested while this tile is gathered and multiplied (first
  // version, offsets loaded per batch: 1.13 ms per 128^2 level against 1.51 ms for columns + product; the HBM floor is ~0.3 ms).  Corner rows of
  // two batches in flight at once (double-buffered) need 96 + 24 registers beside the 27 offsets and spill: one batch at a time.
  float omv[9][3];
  auto load_om = [&](int tile) {
    const long long p0 = (long long)tile * 64;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = tid + i * 512, pl = e / 72, r = e % 72, k = r >> 3, g = r & 7;
      long long pix = p0 + pl;
      pix = pix < P.npix ? pix : P.npix - 1;
      const float* o = P.om + pix * P.om_ld + (g * 18 + 2 * k);
      omv[i][0] = o[0]; omv[i][1] = o[1]; omv[i][2] = o[144 - 9 * g - k];      // (144 + 9 g + k) - (18 g + 2 k)
    }
  };
  if ((int)blockIdx.x < P.ntiles) load_om(blockIdx.x);

  for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    const long long p0 = (long long)tile * 64;
    // ---- (1) deformable gather -> column rows in LDS: three batches of three items per thread ----
    uint4 v[1][3][4];
    float wq[1][3][4], mk[1][3];
    auto fetch = [&](const int b) {
      constexpr int sl = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int it = 3 * b + i;
        const int e = tid + it * 512;
        const int pl = e / 72, r = e % 72, k = r >> 3, g = r & 7;
        // (image, y, x) of the pixel by reciprocal multiplication: the compiler's 64-bit division is ~150 vector instructions, and the gather
        // is VALU-bound (72 items per pixel); the host guarantees n h w < 2^31
        int pix = (int)p0 + pl;
        const bool live = (long long)pix < P.npix;
        pix = live ? pix : (int)P.npix - 1;
        int img, rem, yq, xq;
        xdivmod(pix, hw, P.mg_hw, img, rem);
        xdivmod(rem, P.w, P.mg_w, yq, xq);
        const float dy = omv[it][0], dx = omv[it][1], ml = omv[it][2];
        mk[sl][i] = __builtin_amdgcn_rcpf(1.f + __expf(-ml));          // sigmoid: v_exp_f32 + v_rcp_f32 (~1e-7 relative; the result is rounded to bf16)
        const float py = (float)(yq - 1 + k / 3) + dy, px = (float)(xq - 1 + k % 3) + dx;
        const bool inside = live && py > -1.f && py < fh && px > -1.f && px < fw;
        const float fy = floorf(py), fx = floorf(px);
        const float ly = py - fy, lx = px - fx;
        const int y0 = (int)fmaxf(fminf(fy, fh), -2.f), x0 = (int)fmaxf(fminf(fx, fw), -2.f);
        const float wts[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
        const char* xb = reinterpret_cast<const char*>(P.x) + ((size_t)(unsigned)(img * hw) * (unsigned)P.x_ld + (unsigned)(g * 8)) * 2u;
        const int yc0 = min(max(y0, 0), P.h - 1), yc1 = min(max(y0 + 1, 0), P.h - 1), xc0 = min(max(x0, 0), P.w - 1), xc1 = min(max(x0 + 1, 0), P.w - 1);
        const bool oky0 = y0 >= 0 && y0 <= P.h - 1, oky1 = y0 + 1 >= 0 && y0 + 1 <= P.h - 1, okx0 = x0 >= 0 && x0 <= P.w - 1, okx1 = x0 + 1 >= 0 && x0 + 1 <= P.w - 1;
        const unsigned pb = (unsigned)P.x_ld * 2u;                           // (host: h w x_ld 2 < 2^32 per image)
        const unsigned r0 = (unsigned)(yc0 * P.w) * pb, r1 = (unsigned)(yc1 * P.w) * pb, c0 = (unsigned)xc0 * pb, c1 = (unsigned)xc1 * pb;
        v[sl][i][0] = *reinterpret_cast<const uint4*>(xb + (r0 + c0)); v[sl][i][1] = *reinterpret_cast<const uint4*>(xb + (r0 + c1));
        v[sl][i][2] = *reinterpret_cast<const uint4*>(xb + (r1 + c0)); v[sl][i][3] = *reinterpret_cast<const uint4*>(xb + (r1 + c1));
        wq[sl][i][0] = (inside && oky0 && okx0) ? wts[0] : 0.f; wq[sl][i][1] = (inside && oky0 && okx1) ? wts[1] : 0.f;
        wq[sl][i][2] = (inside && oky1 && okx0) ? wts[2] : 0.f; wq[sl][i][3] = (inside && oky1 && okx1) ? wts[3] : 0.f;
      }
    };
    auto combine = [&](const int b) {
      constexpr int sl = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int e = tid + (3 * b + i) * 512;
        const int pl = e / 72, r = e % 72, k = r >> 3, g = r & 7;
        unsigned o4[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          const unsigned u0 = z == 0 ? v[sl][i][0].x : (z == 1 ? v[sl][i][0].y : (z == 2 ? v[sl][i][0].z : v[sl][i][0].w));
          const unsigned u1 = z == 0 ? v[sl][i][1].x : (z == 1 ? v[sl][i][1].y : (z == 2 ? v[sl][i][1].z : v[sl][i][1].w));
          const unsigned u2 = z == 0 ? v[sl][i][2].x : (z == 1 ? v[sl][i][2].y : (z == 2 ? v[sl][i][2].z : v[sl][i][2].w));
          const unsigned u3 = z == 0 ? v[sl][i][3].x : (z == 1 ? v[sl][i][3].y : (z == 2 ? v[sl][i][3].z : v[sl][i][3].w));
          // the same sums in the same order as the stand-alone column kernel: ((w0 a + w1 b) + w2 c) + w3 d, then the mask
          float lo = wq[sl][i][0] * xbf_lo(u0); lo += wq[sl][i][1] * xbf_lo(u1); lo += wq[sl][i][2] * xbf_lo(u2); lo += wq[sl][i][3] * xbf_lo(u3);
          float hi = wq[sl][i][0] * xbf_hi(u0); hi += wq[sl][i][1] * xbf_hi(u1); hi += wq[sl][i][2] * xbf_hi(u2); hi += wq[sl][i][3] * xbf_hi(u3);
          o4[z] = xcvt_pk_bf16(lo * mk[sl][i], hi * mk[sl][i]);
        }
        *reinterpret_cast<uint4*>(cl + pl * (DCN_ROW * 2) + (k * 64 + g * 8) * 2) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      }
    };
    fetch(0);
    combine(0);
    fetch(1);
    combine(1);
    fetch(2);
    // every offset of this tile has been consumed: request the next tile's now (they land under the rest of this tile)
    if (tile + (int)gridDim.x < P.ntiles) load_om(tile + gridDim.x);
    combine(2);
    __syncthreads();
    // ---- (2) contraction: wave (mt, nt) = 32 pixels x 32 couts over K = 576 ----
    if (wave < 4) {
      const int mt = wave & 1, nt = wave >> 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const unsigned wa = wl_lds + (unsigned)((nt * 32 + li) * (DCN_ROW * 2) + lh * 16);
      const unsigned ca = cl_lds + (unsigned)((mt * 32 + li) * (DCN_ROW * 2) + lh * 16);
#pragma unroll 6
      for (int ks = 0; ks < 36; ++ks) {
        const bf16x8 fw_ = xlds_read16(wa + ks * 32), fc = xlds_read16(ca + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw_, fc, acc, 0, 0, 0);      // D^T: rows = couts, columns (lanes) = pixels
      }
      // register r of lane (li, lh): cout nt*32 + (r & 3) + 8 (r >> 2) + 4 lh of pixel mt*32 + li
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = nt * 32 + 8 * q4 + 4 * lh;
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = apply_act(acc[4 * q4 + j] + bl[c0 + j], P.act);
        *reinterpret_cast<uint2*>(ol + (mt * 32 + li) * 128 + c0 * 2) = make_uint2(xcvt_pk_bf16(t[0], t[1]), xcvt_pk_bf16(t[2], t[3]));
      }
    }
    __syncthreads();
    // ---- (3) store: 64 pixels x 128 bytes, one 16-byte piece per thread ----
    {
      const int pl = tid >> 3, q = tid & 7;
      const long long pix = p0 + pl;
      if (pix < P.npix) *reinterpret_cast<uint4*>(P.out + pix * P.out_ld + q * 8) = *reinterpret_cast<const uint4*>(ol + pl * 128 + q * 16);
    }
  }
}

}  // namespace gpemsr

using namespace gpemsr;

extern "C" int gpemsr_dcn_conv_bf16(const void* x, int n, int h, int w, int x_ld, const float* om, int om_ld, const void* weight_rows, const float* bias, int act,
                                    void* out, int out_ld, void* stream) {
  GP_REQUIRE(x && om && weight_rows && out && n > 0 && h > 0 && w > 0, "dcn_conv_bf16: null pointer / empty tensor");
  GP_REQUIRE(x_ld % 8 == 0 && out_ld % 8 == 0 && om_ld >= 216, "dcn_conv_bf16: x / out strides must be multiples of 8 channels, om needs 216 channels");
  GP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)weight_rows & 15) == 0, "dcn_conv_bf16: 16-byte alignment of x / out / weight rows");
  GP_REQUIRE(act == GPEMSR_ACT_NONE || act == GPEMSR_ACT_RELU || act == GPEMSR_ACT_LRELU, "dcn_conv_bf16: activation none / ReLU / LeakyReLU");
  const long long npix = (long long)n * h * w;
  GP_REQUIRE(npix + 64 < (1ll << 31) && (long long)h * w * x_ld * 2 < (1ll << 32) && (long long)n * h * w * x_ld < (1ll << 32), "dcn_conv_bf16: tensor too large for 32-bit pixel offsets");
  static dev_once_t once{0};
  if (dev_once_begin(once)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_fused16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DCN_LDS) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "dcn_conv_bf16: cannot raise the dynamic LDS limit to %d bytes", DCN_LDS);
    dev_once_done(once);
  }
  DcnParams P;
  P.x = reinterpret_cast<const unsigned short*>(x); P.om = om; P.wrows = reinterpret_cast<const unsigned short*>(weight_rows); P.bias = bias;
  P.out = reinterpret_cast<unsigned short*>(out);
  P.npix = npix; P.h = h; P.w = w; P.x_ld = x_ld; P.om_ld = om_ld; P.out_ld = out_ld; P.act = act; P.ntiles = (int)((npix + 63) / 64);
  P.mg_hw = 0xFFFFFFFFu / (unsigned)(h * w); P.mg_w = 0xFFFFFFFFu / (unsigned)w;
  const int cus = device_cus();
  const int grid = P.ntiles < cus ? P.ntiles : cus;
  hipLaunchKernelGGL(dcn_fused16_kernel, dim3(grid), dim3(512), DCN_LDS, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("dcn_conv_bf16");
}
