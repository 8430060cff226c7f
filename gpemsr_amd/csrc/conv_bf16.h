// Shared declarations of the bf16 implicit-GEMM convolution family: launch parameters (filled by plan_x in conv_bf16.hip), tile
// geometry, the bias stager -- so that kernels of the family can live in their own translation units (conv_bf16.hip takes four
// minutes to compile; a new kernel should not pay that per edit).
#pragma once
#include "bf16_common.h"
#include <type_traits>

namespace gpemsr {

constexpr int XA_LOADS = 10;      // 16-B A slots per DMA thread (halo_px * R / DMA threads; 3x3 16x32 tile on 4 loader waves: 10)
constexpr int XB_LOADS = 9;       // 16-B B slots per thread per stage (TPS * R * BN / 256; 7x7 row stage of 64 couts: 7)

enum { XS_PLAIN = 0, XS_PIXSHUF = 1, XS_CONVT = 2, XS_KPACK = 3 };

struct XParams {
  const unsigned short* src[GPEMSR_MAX_SRC];
  long long img_stride[GPEMSR_MAX_SRC];      // elements
  int ld[GPEMSR_MAX_SRC];
  int c[GPEMSR_MAX_SRC];
  int nsrc;
  int n, h, w, oh, ow, OH, OW;
  int cout;                                  // GEMM N (4*Cout for the transposed form)
  int kw, kk, stride, pad;                   // filter width, taps per chunk, stride, padding
  const unsigned short* weight; long long w_img_stride;
  const float* bias; int act;
  const void* residual; int res_ld, res_f32;
  const float* pixmul;
  int store_mode, cq;
  void* out; int out_ld, out_f32;
  float* out32; int out32_ld;
  float* gn_ws; int gn_parts, gn_cpg;        // [n][gn_parts = tiles per image][cout][2]; channels per GroupNorm group
  long long kpack_img_stride;                // XS_KPACK: elements between images of the packed output
  int tiles_x, tiles_y, tiles_n;
  unsigned mg_x, mg_y, mg_n;                 // floor((2^32 - 1) / tiles_*): division by a run-time tile count as mulhi + one correction (xdivmod)
  int halo_h, halo_w, halo_px;
  int tw_lg;
  int na, nb;                                // DMA slots per thread: A image, B stage image
  int a_bytes, b_bytes;                      // LDS bytes of one A image / one B stage image
  int ring;                                  // B (and, GEMM form, A) ring depth
  int n_abuf;                                // A images in LDS: GEMM: ring; conv: 2 (1 when there is a single chunk)
  int spc;                                   // stages per chunk = kk / TPS
  int nblocks;
  int nbias;                                 // true output channels (bias entries)
  int gpt, ns;                               // resident kernel: workgroups per cout slab, spatial tiles
  int dbg;                                   // diagnostic builds (-DGP16_STAMP) only: descriptor.variant (101: no MFMA loop, 102: no epilogue)
  const float* axs; const float* axh;        // AXF kernels: per (image, input channel) scale / shift applied to the source while it is staged
  int ax_relu;                               // ... followed by ReLU
  float* rowmax;                             // XEPI == 1: [n][rows][tiles_n * WN] (value, column) pairs instead of a stored result
};

struct XGeo { int img, oy0, ox0, n0, tile_in_img; };

// bias -> LDS (zero-padded to a multiple of 8 floats); nbias = number of true output channels
__device__ __forceinline__ void x_stage_bias(const XParams& P, float* bias_lds, int nbias, int nthreads) {
  const int npad = (nbias + 7) & ~7;
  for (int i = threadIdx.x; i < npad; i += nthreads) bias_lds[i] = (P.bias && i < nbias) ? P.bias[i] : 0.f;
}


// kernels of the family that live outside conv_bf16.hip
int launch_convt64_resident(const XParams& P, size_t lds, hipStream_t st);                // convt_bf16.hip
int launch_gemm_direct(const XParams& P, bool lean, bool rowmax, size_t lds, hipStream_t st);      // gemm_bf16.hip: 1x1 / Linear / matrix products, activations straight into registers

}  // namespace gpemsr
