/* gpemsr_hip.h -- C ABI of libgpemsr_hip.so, the MI355X (gfx950) kernel library
 * behind the GPEMSR stage-3 super-resolution forward.
 *
 * The reference (jtshou/GPEMSR) has no FFI: every op on its hot path is a
 * PyTorch ATen call (or basicsr/torchvision op) issued from
 *   GPEMSR-CREMI/GPEMSR/model/GPEMSR.py:323-456  (GPEMSR.forward)
 * Each entry point below replaces the ATen/third-party call(s) named in its
 * comment; the Python host (gpemsr_amd/) binds them with ctypes and a reference
 * maintainer would bind them the same way.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer; activations are float32 NHWC
 *    ([n][h][w][c], c fastest) with an explicit per-pixel stride `ld`
 *    (elements) so a tensor can be a channel slice of a wider buffer
 *    (this is how torch.cat along C is made free);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued, nothing
 *    synchronises, nothing allocates (graph-capture safe);
 *  - return value 0 = success; otherwise a negative GPEMSR_E* code and
 *    gpemsr_last_error() describes it.  No exceptions cross the ABI.
 *  - the library owns no memory; the caller owns inputs, outputs, workspaces.
 */
#ifndef GPEMSR_HIP_H
#define GPEMSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPEMSR_ABI_VERSION 1

enum { GPEMSR_OK = 0, GPEMSR_EINVAL = -1, GPEMSR_ELAUNCH = -2, GPEMSR_EUNSUPPORTED = -3 };

/* activation applied to (acc + bias) */
enum {
  GPEMSR_ACT_NONE = 0,
  GPEMSR_ACT_RELU = 1,          /* basicsr ResidualBlockNoBN, VGG, SpyNet, VQGAN blocks */
  GPEMSR_ACT_LRELU = 2,         /* LeakyReLU(0.1): model/GPEMSR.py:96,168,321 */
  GPEMSR_ACT_SIGMOID = 3,
  GPEMSR_ACT_LRELU_SIGMOID = 4  /* sigmoid(lrelu(x)): model/GPEMSR.py:398-399 */
};

int gpemsr_abi_version(void);
const char* gpemsr_last_error(void);
/* name, CU count, total HBM bytes of the current device (diagnostics for bench.py) */
int gpemsr_device_info(char* name, int name_len, int* cu_count, int64_t* hbm_bytes);

/* ---------------------------------------------------------------------------
 * Convolution family (implicit GEMM on v_mfma_f32_32x32x2_f32, LDS-staged halo)
 * replaces: F.conv2d / nn.Conv2d (k in {1,3,7}, stride in {1,2}, pad k/2),
 *           nn.ConvTranspose2d(k3,s2,p1,op1)  (model/GPEMSR.py:252-254, blocks.py:35),
 *           nn.Linear and torch.bmm (as 1x1 convs with per-image weights:
 *           blocks.py:75,80; indexer.py:100), nn.PixelShuffle(2) fused into the
 *           store (model/GPEMSR.py:442-448), torch.cat along C (multi-source input),
 *           residual adds and the mask multiply of the MPF (model/GPEMSR.py:403-411).
 * ------------------------------------------------------------------------- */
#define GPEMSR_MAX_SRC 4

typedef struct {
  const float* ptr;  /* NHWC base of this source */
  int32_t ld;        /* elements between consecutive pixels */
  int32_t c;         /* channels taken from this source */
} gpemsr_src_t;

typedef struct {
  int32_t n, h, w;                 /* input images, height, width */
  int32_t nsrc;
  gpemsr_src_t src[GPEMSR_MAX_SRC];/* virtual concat along C, in order */
  int64_t src_image_stride[GPEMSR_MAX_SRC]; /* elements between images; 0 = shared by all n; <0 = dense (h*w*ld) */
  int32_t cout;
  int32_t ksize;                   /* 1, 3 or 7 */
  int32_t stride;                  /* 1 or 2 (ignored when transposed) */
  int32_t transposed;              /* 1 = ConvTranspose2d(k=3,s=2,p=1,op=1): out is 2h x 2w.
                                      2 = ROW-PAIR form of a 7x7 stride-1 convolution with 16 output channels (SpyNet's 32 -> 16 layers,
                                      basicsr spynet_arch.BasicModule via R:model/GPEMSR.py:67): same result, but the 32-row matrix tile
                                      computes the 16 couts of output rows 2i AND 2i+1 from an 8 x 7-tap window; weight =
                                      [tap = ky'*7+kx, ky' < 8][32][cin_pad], rows 0..15 = W[ky'] (0 for ky' = 7), rows 16..31 = W[ky'-1]
                                      (0 for ky' = 0) (gpemsr_amd/packing.py::pack_rowpair7); cout stays 16, out is h x w.
                                      3 = WINOGRAD F(2x2, 3x3) form of a 3x3 stride-1 convolution (same result to fp32 rounding, 16/36 of
                                      the multiplies; csrc/conv_wino.hip): every source c % 8 == 0, cout % 32 == 0, no pixel_shuffle /
                                      gn_partials / cos_partials; weight = U[cin / 8][16 positions][cout][8] = G g G^T in the order the
                                      kernel stages it (gpemsr_amd/packing.py::pack_winograd).
                                      4 = 1-D WINOGRAD F(2, 7) form of a 7x7 stride-1 convolution along the image rows (SpyNet's 32 -> 64 and
                                      64 -> 32 layers, basicsr spynet_arch.BasicModule via R:model/GPEMSR.py:67,98-100; same result to fp32
                                      rounding, 8/14 of the multiplies; csrc/conv7_wino.hip): ONE source with c % 8 == 0, cout % 32 == 0, no
                                      residual / pixmul / pixel_shuffle / gn_partials / cos_partials; weight = U[cin/8][7 ky][8 nu][2][cout][4]
                                      = G g per filter row (gpemsr_amd/packing.py::pack_winograd7).
                                      5 = WINOGRAD F(4x4, 3x3) form of a 3x3 stride-1 convolution (same result to fp32 rounding -- ~2e-5 of
                                      the result's scale at 512 channels --, 36/144 of the multiplies; csrc/conv_wino4.hip; the VQGAN prior's
                                      128-512-channel convolutions, R:model/blocks.py:5-29, and the 64-channel fusion / reconstruction /
                                      VGG layers): every source c % 8 == 0, cout % 64 == 0 (or any cout with U zero-padded to the next multiple of 64:
                                      plain store / residual only), act NONE / RELU / LRELU; epilogues: plain store
                                      (+ gn_partials), residual (+ pixmul), pixel_shuffle (cout % 256 == 0, alone), cos_partials (cout == 64,
                                      h % 16 == 0, w % 32 == 0, operand map in `residual`); weight = U[cin/8][36 positions][2][cout][4] = G g G^T
                                      (gpemsr_amd/packing.py::pack_winograd4).
                                      6 = 2-D WINOGRAD F(2x2, 7x7) form of a 7x7 stride-1 convolution (the same SpyNet layers as form 4; same
                                      result to fp32 rounding -- ~5e-6 of the result's scale --, 64/196 of the multiplies;
                                      csrc/conv7_wino2d.hip): ONE source with c % 8 == 0, cout % 16 == 0 (32 couts per workgroup, or 16 on
                                      v_mfma_f32_16x16x4_f32 when cout % 32 != 0), act NONE / RELU / LRELU, plain store
                                      with 8-byte aligned rows; weight = U[cin/8][64 positions = 8 xi + nu][2][cout][4] = G g G^T
                                      (gpemsr_amd/packing.py::pack_winograd77) */
  const float* weight;             /* packed [tap][cout][cin_pad], tap = ky*k+kx, cin fastest, cin padded per source to 8
                                      (k>=3) or 32 (k=1).  transposed: [tap = 2*dy+dx][n' = (co/32)*128 + q*32 + co%32][cin_pad],
                                      q = 2*py+px, the phase-stacked 2x2-tap form (gpemsr_amd/packing.py::pack_convT) */
  int64_t weight_image_stride;     /* elements; 0 = one weight set for all images (normal conv) */
  const float* bias;               /* [cout] or NULL */
  int32_t act;                     /* GPEMSR_ACT_* */
  const float* residual;           /* added after act; same geometry as out; NULL = none */
  int32_t res_ld;
  const float* pixmul;             /* [n][oh][ow] multiplier applied last; NULL = none */
  int32_t pixel_shuffle;           /* 1: store as PixelShuffle(2); weight rows pre-permuted so that
                                      cout index = (2*i+j)*(cout/4)+c ; out is [n][2oh][2ow][cout/4] */
  float* out;
  int32_t out_ld;
  float* gn_partials;              /* optional: per (tile, channel) sum / sum of squares of (conv + bias) -- the first pass of the GroupNorm
                                      that follows (R:model/blocks.py:5-6,16-21) -- as [n][parts][cout][2], parts = gpemsr_conv2d_gn_parts(d);
                                      feed to gpemsr_groupnorm_finish + gpemsr_groupnorm_apply.  Needs act NONE, no residual / pixmul /
                                      pixel_shuffle / transposed, cout % 4 == 0.  NULL = none */
  float* cos_partials;             /* optional: do NOT store the result; accumulate, per 4-row strip and 16-pixel patch column, the sums
                                      (b.a, a.a, b.b) of the result b against the tensor a given in `residual` (same geometry) -- the
                                      16x16-patch cosine of two VGG relu1_2 maps (R:model/GPEMSR.py:387-395) without the second map in
                                      memory: [n][h/4][w/16][4] floats, then gpemsr_patch_cosine_finish.  33..64 output channels, h % 16 == 0,
                                      w % 32 == 0.  NULL = none */
  const float* a_scale;            /* optional (transposed = 5 only, one source): the source is read as relu(a_scale[n][c] x + a_shift[n][c]) -- the
                                      GroupNorm + ReLU of the producing layer (R:model/blocks.py:5-29: block.1 / block.2 of a ResidualBlock) folded
                                      into the consuming convolution's input transform; tables from gpemsr_groupnorm_scale_shift; padding stays
                                      zero.  NULL = the source as stored */
  const float* a_shift;            /* with a_scale */
  int32_t a_relu;                  /* with a_scale: must be 1 */
} gpemsr_conv_desc;

int gpemsr_conv2d(const gpemsr_conv_desc* d, void* stream);
/* cos_partials workspace -> the cosine map [n][ph][pw] of R:model/GPEMSR.py:387-395 (ph = h/16, pw = w/16) */
int gpemsr_patch_cosine_finish(const float* ws, int n, int ph, int pw, float* out, void* stream);
/* records per image a launch of `d` writes to d->gn_partials (the tiling is chosen by the library); < 0: error */
int gpemsr_conv2d_gn_parts(const gpemsr_conv_desc* d);
/* introspection (no reference counterpart; bench.py's per-kernel roofline table): the name of the kernel instantiation
 * gpemsr_conv2d would launch for `d`, written to buf[cap] as text.  Nothing is launched.  0 or a negative error code */
int gpemsr_conv2d_kernel_name(const gpemsr_conv_desc* d, char* buf, int cap);

/* The same 3x3 stride-1 convolution on the bf16 matrix pipe.  nsplit = 2: every fp32 operand is split into hi + lo bf16 and the
 * product evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation (relative error ~2^-16 per product: fp32-grade for the 1e-3
 * parity bar, 5.3x less matrix-pipe time); nsplit = 1: plain bf16 operands.  `d` is interpreted as for gpemsr_conv2d except that
 * d->weight is ignored: weight_bf16 = [plane (hi, lo)][cin_total/16][tap][k-half][cout][8] bf16 -- the kernel's staging order
 * (gpemsr_amd/packing.py::pack_conv_split / _stage_order),
 * plane_stride in elements.  Every source needs c % 16 == 0 and 16-byte aligned rows.  Activations stay fp32 in HBM. */
int gpemsr_conv2d_split(const gpemsr_conv_desc* d, const void* weight_bf16, int64_t plane_stride, int nsplit, void* stream);
/* 1x1 / Linear / batched-matmul form of gpemsr_conv2d_split: d->ksize == 1, stride 1, every source c % 32 == 0.  With
 * d->weight_image_stride != 0 (in bf16 elements) image i uses the weights at weight_bf16 + i*stride -- the attention
 * products of model/blocks.py:75,80, whose B operand is an activation.  gpemsr_split_pack_rows turns such fp32 rows
 * [n][rows][k] into that layout ([n][plane (hi, lo)][k/16][k-half][rows][8] bf16; per-image stride 2*rows*k, plane
 * stride rows*k). */
int gpemsr_split_pack_rows(const float* src, int n, int rows, int k, int ld, int64_t img_stride, void* dst_bf16, void* stream);

/* ---------------------------------------------------------------------------
 * bf16 data path (precision = "bf16": BASELINE.json configs[2], "bf16 MFMA").  Activations are bf16 NHWC in HBM
 * ([n][h][w][ld], ld % 8 == 0, channel counts multiples of 16), 1-channel images (LR slices, prior image, masks, flows)
 * and the indexer's logits stay fp32.  Entry points mirror the fp32 ones above and replace the same reference calls.
 * ------------------------------------------------------------------------- */
typedef struct {
  const void* ptr;   /* bf16 NHWC base of this source */
  int32_t ld;        /* elements between consecutive pixels (multiple of 8) */
  int32_t c;         /* channels taken from this source (multiple of 16; every source of a launch a multiple of 32, or 16-granular) */
} gpemsr_src16_t;

typedef struct {
  int32_t n, h, w;
  int32_t nsrc;
  gpemsr_src16_t src[GPEMSR_MAX_SRC];      /* virtual concat along C, in order */
  int64_t src_image_stride[GPEMSR_MAX_SRC]; /* elements between images; 0 = shared by all n; <0 = dense (h*w*ld) */
  int32_t cout;
  int32_t ksize;                   /* 1, 3 or 7 */
  int32_t stride;                  /* 1, or 2 for 3x3 with cout > 32 (ignored when transposed) */
  int32_t transposed;              /* 1 = ConvTranspose2d(k=3,s=2,p=1,op=1) */
  const void* weight;              /* bf16, staged order [cin_total/CK][tap][CK/8][cout][8] (CK = 32, or 16 when a source is an odd
                                      multiple of 16); transposed: tap = 2*dy+dx of the phase-stacked 2x2 form, cout -> 4*cout rows
                                      (gpemsr_amd/packing.py::pack_conv_bf16 / pack_convT_bf16) */
  int64_t weight_image_stride;     /* elements; != 0: image i uses weight + i*stride (1x1 only: attention products) */
  const float* bias;               /* fp32 [cout] or NULL */
  int32_t act;                     /* GPEMSR_ACT_* */
  const void* residual;            /* added after act; bf16 (res_f32 = 0) or fp32 (res_f32 = 1); NULL = none */
  int32_t res_ld, res_f32;
  const float* pixmul;             /* fp32 [n][oh][ow] multiplier applied last; NULL = none */
  int32_t pixel_shuffle;           /* same meaning as in the fp32 descriptor; cout % 32 == 0 */
  int32_t kpack;                   /* 1: store the bf16 result as the B operand of a later product: [n][cout/8][oh*ow][8] */
  void* out; int32_t out_ld;       /* bf16 (out_f32 = 0) or fp32 (out_f32 = 1) */
  int32_t out_f32;
  float* out32; int32_t out32_ld;  /* optional: the un-rounded fp32 result as well (master copy of residual trunks); NULL = none */
  float* gn_partials;              /* optional: per (tile, channel) sum / sum of squares of (conv + bias) -- the first pass of
                                      GroupNorm (model/blocks.py:5-6) -- as [n][parts][cout][2], parts = gpemsr_conv2d_bf16_gn_parts();
                                      feed to gpemsr_groupnorm_finish.  NULL = none */
  int32_t variant;                 /* 0 = default tile choice; other values select alternative tilings (tuning only) */
  int32_t gn_cpg;                  /* with gn_partials: channels per GroupNorm group (0 / 1: unknown).  When it is a multiple of 2 / 4 the
                                    * kernel adds 2 / 4 neighbouring channels BEFORE the cross-lane reduction and leaves zeros in the other
                                    * channel slots of the workspace -- the per-group totals gpemsr_groupnorm_finish forms are unchanged */
  const float* a_scale;            /* optional, with a_shift: fp32 [n][cin].  The SOURCE is read as relu?(a_scale[img][c] * x + a_shift[img][c])
                                    * (fma in fp32, rounded to bf16; zero padding applies to the transformed tensor) -- the apply pass of
                                    * Normalize + ReLU (R:model/blocks.py:5-6,13-20: GroupNorm(32, eps 1e-6, affine) between the two convolutions
                                    * of a ResidualBlock) folded into the convolution that consumes it; tables from gpemsr_groupnorm_scale_shift.
                                    * Only for layers gpemsr_conv2d_bf16_axf_ok() admits (3x3, stride 1, one dense source of 64 or k*32 channels). */
  const float* a_shift;
  int32_t a_relu;                  /* with a_scale: 1 = ReLU after the affine map */
  int32_t weight_forms;            /* bit 0 (transposed, 64 input channels, cout % 64 == 0): the weight buffer carries, behind the staged form's
                                    * 16 * cin * cout elements, the nine non-zero (tap, phase) blocks per 64-cout slab
                                    * [cout/64][cin/32][block 9][4][64 couts][8] (gpemsr_amd/packing.py::pack_convT_bf16): the layer runs
                                    * on the weights-resident transposed kernel.
                                    * bit 1 (transposed, cin % 32 == 0): behind the staged form, the COMPACT form
                                    * [cin/32][tap 4][piece 4][cout/32][rows_t][8], rows_t = 128, 64, 64, 32 = the phases a tap feeds
                                    * (increasing q) x 32 couts: the loader-wave kernel stages 18 KB per chunk instead of 32 KB.
                                    * 0: staged form only */
  float* rowmax;                   /* optional (1x1 form, cout > 64, sources % 64 == 0): do NOT store the result; leave, per GEMM row and
                                    * column part, the maximum of (product + bias) and its column (lowest on ties) as float pairs
                                    * (value, column bits): [n][oh*ow][parts], parts = gpemsr_conv2d_bf16_rowmax_parts(d) -- the arg-max of
                                    * the codebook logits (R:model/indexer.py:100 + R:model/codebook.py:34-43) without the
                                    * [rows][1024] logits in memory; gpemsr_rowmax_finish folds the parts.  NULL = none */
} gpemsr_conv16_desc;

int gpemsr_conv2d_bf16(const gpemsr_conv16_desc* d, void* stream);
/* 1: this launch geometry has a kernel that takes a_scale / a_shift; 0: it has not (apply GroupNorm separately); < 0: error code */
int gpemsr_conv2d_bf16_axf_ok(const gpemsr_conv16_desc* d);
/* scale[n][c] = rstd * gamma, shift[n][c] = beta - mean * rstd * gamma from mean_rstd[n][groups][2] (gpemsr_groupnorm_finish): the
 * per-channel affine form of GroupNorm's apply pass, bit-compatible with gpemsr_groupnorm_apply_bf16 (same fma) */
int gpemsr_groupnorm_scale_shift(const float* mean_rstd, const float* gamma, const float* beta, int n, int c, int groups,
                                 float* scale, float* shift, void* stream);
/* rows of the gn_partials workspace per image for this launch geometry (>= 1), or a negative error code */
int gpemsr_conv2d_bf16_gn_parts(const gpemsr_conv16_desc* d);
/* row-maximum records per GEMM row for d->rowmax (>= 1), or a negative error code; gpemsr_rowmax_finish: ws[rows][parts] (value,
 * column) pairs -> idx[rows] int32, the column of the row maximum (lowest column on ties: torch's argmax on CPU) */
int gpemsr_conv2d_bf16_rowmax_parts(const gpemsr_conv16_desc* d);
int gpemsr_rowmax_finish(const float* ws, int64_t rows, int parts, int32_t* idx, void* stream);
/* the same introspection as gpemsr_conv2d_kernel_name for the bf16 family (tile choice of plan_x as text); nothing is launched */
int gpemsr_conv2d_bf16_kernel_name(const gpemsr_conv16_desc* d, char* buf, int cap);

/* GroupNorm on bf16 tensors (model/blocks.py:5-6,13-28).  The first pass (per-channel sum / sum of squares) comes either from
 * the producing convolution's epilogue (gpemsr_conv16_desc.gn_partials) or from gpemsr_groupnorm_stats_bf16; both fill
 * ws[n][parts][c][2].  gpemsr_groupnorm_finish folds it into mean_rstd[n][groups][2] (fixed order, deterministic);
 * gpemsr_groupnorm_apply_bf16 normalises (+ReLU, + bf16 residual). */
int gpemsr_groupnorm_stats_bf16(const void* x, int n, int hw, int c, int ld, float* ws, int parts, void* stream);
int gpemsr_groupnorm_finish(const float* ws, int n, int hw, int c, int groups, int parts, float eps, float* mean_rstd, void* stream);
int gpemsr_groupnorm_apply_bf16(const void* x, int n, int hw, int c, int ld, int groups, const float* mean_rstd,
                                const float* gamma, const float* beta, int relu, const void* residual, int res_ld,
                                void* out, int out_ld, void* stream);
/* row softmax (blocks.py:77): s[rows][cols] fp32 (s_f32 = 1) or bf16 -> p bf16; cols % 8 == 0, <= 8192 */
int gpemsr_softmax_rows_bf16(const void* s, int s_f32, int64_t rows, int cols, int s_ld, void* p, int p_ld, void* stream);
/* codebook lookup (codebook.py:41): out[r] = bf16(table[idx[r]]), fp32 table */
int gpemsr_gather_rows_bf16(const float* table, int dim, const int32_t* idx, int64_t rows, void* out, int out_ld, void* stream);
/* bf16 rows [n][rows][c] -> B-operand layout [n][c/8][rows][8] of gpemsr_conv2d_bf16's 1x1 form (attention: k, v^T) */
int gpemsr_pack_rows_bf16(const void* src, int n, int rows, int c, int ld, int64_t img_stride, void* dst, void* stream);
/* the same with perm16 != 0: inside every group of 16 rows the rows are stored in the order 0-3, 8-11, 4-7, 12-15 (rows % 16 == 0) */
int gpemsr_pack_rows_bf16_ex(const void* src, int n, int rows, int c, int ld, int64_t img_stride, void* dst, int perm16, void* stream);
/* Single-head attention of NonLocalBlock (R:model/blocks.py:75-80: attn = softmax(q k^T) over keys, A = attn v) with the score matrix
 * kept on chip (flash style): q bf16 [n][tokens][q_ld] (the C^-1/2 of :76 folded into it), kp = k as [n][C/8][tokens][8] (the "kpack"
 * store of gpemsr_conv2d_bf16), vtp = v^T as [n][tokens/8][C][8] with the keys of every 16-group in the perm16 order above (the v^T
 * product over a perm16-packed B operand), bias_v [C] or NULL (added after the product: softmax rows sum to 1) ->
 * out bf16 [n][tokens][out_ld].  channels == 512, tokens % 128 == 0; fp32 accumulation, online softmax in fp32. */
int gpemsr_flash_attention_bf16(const void* q, int q_ld, const void* kp, const void* vtp, const float* bias_v, int n, int tokens, int channels,
                                void* out, int out_ld, void* stream);
/* format changes at the module boundary / between the fp32 and bf16 parts of the path */
int gpemsr_cast_f32_bf16(const float* x, int64_t pixels, int c, int x_ld, void* out, int out_ld, void* stream);
int gpemsr_cast_bf16_f32(const void* x, int64_t pixels, int c, int x_ld, float* out, int out_ld, void* stream);
/* fp32 -> two bf16 tensors, hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to 2^-17): the A operands of the three-product form of the
 * indexer's nn.Linear (indexer.py:100; logits keep fp32 precision on the bf16 matrix pipe).  c, row strides % 4 == 0 */
int gpemsr_split_f32_bf16x2(const float* x, int64_t pixels, int c, int x_ld, void* hi, int hi_ld, void* lo, int lo_ld, void* stream);
/* bf16 counterparts of gpemsr_bilinear / _pool3s2_maxavg / _spynet_prep (16-channel bf16 level input, channels 8..15 zero; flows
 * stay fp32) / _dcn_columns (x and columns bf16, offsets + mask logits fp32) / _patch_cosine / _temporal_gate /
 * _frame_mix_lrelu / _threeda_combine / _copy_channels (c % 8 == 0) */
int gpemsr_bilinear_bf16(const void* x, int n, int h, int w, int c, int ld, int oh, int ow, int align_corners, float mul,
                         void* out, int out_ld, void* stream);
/* nn.MaxPool2d(2, 2) on a bf16 tensor (the loss network's pools, R:model/VGG.py:22,24); c, strides % 8 == 0 */
int gpemsr_maxpool2_bf16(const void* x, int n, int h, int w, int c, int ld, void* out, int out_ld, void* stream);
int gpemsr_pool3s2_maxavg_bf16(const void* x, int n, int h, int w, int c, int ld, void* out, int out_ld, void* stream);
int gpemsr_spynet_prep_bf16(const float* ref, const float* supp, const float* flow_coarse, int n, int h, int w,
                            const float* mean3, const float* std3, float* up_flow, void* inp16, void* stream);
int gpemsr_dcn_columns_bf16(const void* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                            void* col, void* stream);
/* Modulated deformable convolution in ONE kernel (csrc/dcn_bf16.hip): the deformable sampling of gpemsr_dcn_columns_bf16 into LDS and the
 * 64 x (9 x 64) contraction on the matrix cores from there -- the column tensor never exists in HBM.  Replaces basicsr DCNv2Pack.forward ->
 * torchvision.ops.deform_conv2d (R:model/GPEMSR.py:79-94 call sites :115,122,131,138) after its conv_offset convolution, for 64 -> 64
 * channels, 8 deformable groups, 3x3, stride 1, pad 1.  x [n][h][w][x_ld >= 64] bf16; om [n][h][w][om_ld >= 216] fp32 = the raw conv_offset
 * output (144 offsets: group g, tap k -> dy = om[18g+2k], dx = om[18g+2k+1]; 72 mask logits om[144+9g+k], sigmoid applied here);
 * weight_rows [64 couts][9 taps][64 channels] bf16; bias[64] fp32 or NULL; act: none / ReLU / LeakyReLU; out [n][h][w][out_ld >= 64] bf16. */
int gpemsr_dcn_conv_bf16(const void* x, int n, int h, int w, int x_ld, const float* om, int om_ld, const void* weight_rows,
                         const float* bias, int act, void* out, int out_ld, void* stream);
int gpemsr_patch_cosine_bf16(const void* a, const void* b, int n, int h, int w, int c, float* out, void* stream);
int gpemsr_temporal_gate_bf16(const void* aligned, const void* emb, const void* emb_ref, int b, int t, int hw, int c,
                              void* af, void* stream);
int gpemsr_frame_mix_lrelu_bf16(const void* af, int64_t pixels, int t, int c, const float* m, const float* bias, void* out,
                                void* stream);
int gpemsr_threeda_combine_bf16(const void* feat, const void* attn, const void* attn_add, const void* f2, const void* f3,
                                int64_t count, void* out, void* stream);
int gpemsr_copy_channels_bf16(const void* src, int src_ld, void* dst, int dst_ld, int64_t pixels, int c, void* stream);
int gpemsr_copy_channels_f32_bf16(const float* src, int src_ld, void* dst, int dst_ld, int64_t pixels, int c, void* stream);
/* Fused mask front end (model/GPEMSR.py:385-395): VGG19 relu1_2 of the prior image `ref_img` [n][s*h][s*w] and of the bilinearly
 * up-sampled (align_corners=False) LR slice `lr` [n][h][w], both fp32 1-channel images (the reference expands them to 3 identical
 * channels; w1 = conv1_1 weights summed over the input channels, [64][9], tap = 3*ky + kx), and the cosine similarity of their
 * co-located 16x16x64 patches -> out [n][s*h/16][s*w/16].  conv1_1 runs with fp32-accurate inputs (hi + lo bf16 halves), conv1_2
 * with bf16 operands (w2_bf16 = vgg.slice1.2 in the staged order of gpemsr_conv2d_bf16), fp32 accumulation and reduction.
 * Neither feature map touches HBM.  scale == 1: `lr` is already at the HR size (up-sampled once by gpemsr_bilinear: 4 MB per slice) and both
 * images are read the same way -- the form the engine uses; scale > 1 resamples the LR slice on the fly inside the kernel (no up-sampled
 * image in HBM, but 36 loads + ~250 vector operations per halo-pixel group: 4 ms slower per step at batch 16). */
int gpemsr_vgg_mask_bf16(const float* ref_img, const float* lr, int n, int h, int w, int scale, const float* w1, const float* b1,
                         const void* w2_bf16, const float* b2, float* out, void* stream);
/* gpemsr_conv2d_stem1 with bf16 output (cout % 8 == 0); gpemsr_conv2d_direct with fp32 or bf16 input / output (fp32 packed
 * weights as for gpemsr_conv2d_direct; the 64 -> 1 3x3 form takes an fp32 residual) */
int gpemsr_conv2d_stem1_bf16(const float* x, int n, int h, int w, const float* weight, const float* bias, int cout, int act,
                             void* out, int out_ld, void* stream);
int gpemsr_conv2d_direct_bf16(const void* x, int x_f32, int n, int h, int w, int ld, int cin, const float* weight,
                              const float* bias, int cout, int ksize, int stride, int act, const float* residual, int res_ld,
                              void* out, int out_f32, int out_ld, void* stream);
/* One-output-channel convolutions over 64 bf16 channels as tap partial products on the matrix cores (csrc/tap_sum.hip): the input
 * is read from HBM once, fp32 result [n][h][w] with pixel stride out_ld.  wfrag: MFMA A-operand fragments of the taps as bf16
 * hi + lo halves (packing.pack_cout1_taps / pack_upconv_out).
 *   conv_c64_cout1: Conv2d(64 -> 1, 3x3, pad 1) + act + fp32 residual -- conv_last (model/GPEMSR.py:318,455); out_u8 (optional,
 *                   [n][h][w]): additionally the reference's tensor2img of the result (util/util.py:145-163), so the 8-bit image
 *                   leaves the network's last kernel (conv_last + base + uint8 in one pass).
 *   upconv_out_c64: ConvTranspose2d(64 -> 64, k3 s2 p1 op1) then Conv2d(64 -> 1, 3x3, pad 1) with nothing in between, composed
 *                   into one 5x5 stride-2 operator 64 -> 1 (the VQGAN decoder's last up-block + output_layer, model/vqgan.py):
 *                   x [n][h][w] -> out [n][2h][2w]; consts = [9 bias-through-tap sums, b2, Wy0 5x64, Wx0 5x64, Wc 64] fp32. */
int gpemsr_conv_c64_cout1_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                               const float* residual, int res_ld, float* out, int out_ld, uint8_t* out_u8, void* stream);
int gpemsr_upconv_out_c64_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* consts, float* out,
                               int out_ld, void* stream);
/* The same two operators for the exact-fp32 path: fp32 NHWC input (ld % 4 == 0), v_mfma_f32_32x32x2_f32, fp32 weights
 * (wfrag: [32 k-steps][64 lanes] floats, packing.pack_cout1_taps_f32 / pack_upconv_out_f32). */
int gpemsr_conv_c64_cout1_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* bias, int act,
                              const float* residual, int res_ld, float* out, int out_ld, uint8_t* out_u8, void* stream);
int gpemsr_upconv_out_c64_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* consts, float* out,
                              int out_ld, void* stream);
/* Conv2d(16 -> 2, 7x7, pad 3) + fp32 residual [n][h][w][res_ld >= 2] -> fp32 out [n][h][w][out_ld >= 2]: SpyNet's flow-update
 * convolution (basicsr SpyNet BasicModule, last conv) as row sums on the matrix cores + a vertical 7-sum (csrc/tap_sum.hip);
 * wfrag from packing.pack_rowsum7. */
int gpemsr_conv7_c16_cout2_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, const float* residual,
                                int res_ld, float* out, int out_ld, void* stream);
/* Conv2d(32 -> 16, 7x7, pad 3) on bf16 NHWC tensors (basicsr SpyNet BasicModule's fourth convolution, R:model/GPEMSR.py:67,99) on
 * v_mfma_f32_16x16x32_bf16 with the weights resident in LDS (csrc/conv7_bf16.hip): x [n][h][w][ld >= 32], wfrag = [49 taps][4 k-groups]
 * [16 couts][8] bf16 (packing.pack_conv7_c32_cout16), bias fp32 [16] or NULL, act NONE / RELU / LRELU, out bf16 [n][h][w][out_ld >= 16]. */
int gpemsr_conv7_c32_cout16_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                                 void* out, int out_ld, void* stream);
/* Conv2d(8 -> 32, 7x7, pad 3): the first convolution of a SpyNet BasicModule.  x bf16 NHWC with pixel stride ld >= 8 (the first 8
 * channels of every pixel are read: gpemsr_spynet_prep_bf16's 16-channel tensor qualifies), four taps per v_mfma_f32_16x16x32_bf16;
 * wfrag = [13 tap groups][2 cout tiles][64 lanes][8] bf16 (packing.pack_conv7_c8_cout32), out bf16 [n][h][w][out_ld >= 32]. */
int gpemsr_conv7_c8_cout32_bf16(const void* x, int n, int h, int w, int ld, const void* wfrag, const float* bias, int act,
                                void* out, int out_ld, void* stream);
/* the same for fp32 activations (exact-fp32 path): v_mfma_f32_32x32x2_f32, wfrag [7 kx][8 k-steps][64 lanes] floats (packing.pack_rowsum7_f32) */
int gpemsr_conv7_c16_cout2_f32(const float* x, int n, int h, int w, int ld, const float* wfrag, const float* bias, const float* residual,
                               int res_ld, float* out, int out_ld, void* stream);

/* Direct (VALU) convolution for tiny channel counts: cout <= 16, any k<=7, stride 1/2/4.
 * replaces: POD.flowdsconv* (model/GPEMSR.py:70-75,101-106), SpyNet's last 16->2 conv,
 * conv_last / decoder.output_layer / refmaskconv3 (cout = 1).  weight layout as above. */
int gpemsr_conv2d_direct(const float* x, int n, int h, int w, int ld, int cin,
                         const float* weight, const float* bias, int cout, int ksize, int stride,
                         int act, const float* residual, int res_ld, float* out, int out_ld, void* stream);

/* 1-channel -> cout (multiple of 4) 3x3 stride-1 stem convolution; x is a dense 1-channel image [n][h][w]; weight in the
 * packed [9][cout][8] layout.  HBM-write-bound.  replaces: vgg conv1_1 on the expanded 1-channel image (model/GPEMSR.py:386,390;
 * weights summed over the 3 identical input channels), conv_first (:329), refmaskconv1 (:396), indexer input conv. */
int gpemsr_conv2d_stem1(const float* x, int n, int h, int w, const float* weight, const float* bias, int cout,
                        int act, float* out, int out_ld, void* stream);

/* ---------------------------------------------------------------------------
 * GroupNorm(32, eps) statistics + fused apply (+ReLU, +residual)
 * replaces: model/blocks.py:5-6,13-28  (Normalize -> ReLU, and x + block(x))
 * stats: ws must hold n*parts*c*2 floats; mean_rstd gets [n][groups][2].
 * ------------------------------------------------------------------------- */
int gpemsr_groupnorm_stats(const float* x, int n, int hw, int c, int ld, int groups, float eps,
                           float* ws, int parts, float* mean_rstd, void* stream);
int gpemsr_groupnorm_apply(const float* x, int n, int hw, int c, int ld, int groups, const float* mean_rstd,
                           const float* gamma, const float* beta, int relu,
                           const float* residual, int res_ld, float* out, int out_ld, void* stream);

/* row softmax in place: x[rows][cols] (blocks.py:77), row argmax -> int32 (codebook.py:38-40,
 * ties -> lowest index) and row gather out[r] = table[idx[r]] (codebook.py:41). */
int gpemsr_softmax_rows(float* x, int64_t rows, int cols, void* stream);
/* the same on rows that sit ld >= cols floats apart (score matrices padded to the GEMM's 32-column granule) */
int gpemsr_softmax_rows_ld(float* x, int64_t rows, int cols, int ld, void* stream);
int gpemsr_argmax_rows(const float* x, int64_t rows, int cols, int32_t* idx, void* stream);
int gpemsr_gather_rows(const float* table, int dim, const int32_t* idx, int64_t rows, float* out, int out_ld,
                       void* stream);

/* ---------------------------------------------------------------------------
 * Resampling
 * ------------------------------------------------------------------------- */
/* F.interpolate(mode='bilinear'): out = mul * bilinear(x).  align_corners 0/1.
 * replaces model/GPEMSR.py:99,107-110,119,123,128,132,211,215,385,403-411,451-455 */
int gpemsr_bilinear(const float* x, int n, int h, int w, int c, int ld, int oh, int ow, int align_corners,
                    float mul, float* out, int out_ld, void* stream);
/* F.avg_pool2d(2,2) (SpyNet pyramid) */
int gpemsr_avgpool2(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream);
/* MaxPool2d(3,2,1) and AvgPool2d(3,2,1) (count_include_pad) written side by side:
 * out[..., 0:c] = max, out[..., c:2c] = avg  == torch.cat([max, avg], 1)  (model/GPEMSR.py:201-204) */
int gpemsr_pool3s2_maxavg(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream);
/* One SpyNet level input (basicsr SpyNet.process): up = 2*bilinear_x2(flow, align_corners=True)
 * (zeros when flow == NULL), inp = cat[norm(ref), flow_warp(norm(supp), up, border), up] (8 ch).
 * ref/supp are the 1-channel raw pyramids; mean3/std3 are HOST pointers to the three
 * ImageNet mean/std constants that basicsr broadcasts over the (1-channel) input. */
int gpemsr_spynet_prep(const float* ref, const float* supp, const float* flow_coarse, int n, int h, int w,
                       const float* mean3, const float* std3, float* up_flow, float* inp8, int inp_ld, void* stream);
/* Modulated deformable sampling (torchvision deform_conv2d, k3 p1): builds the column tensor
 * col[n][h][w][9*c] (tap-major) from x and the raw conv_offset output `om` ([.., 3*groups*9]:
 * chunk 0,1 -> offsets (basicsr DCNv2Pack cat(o1,o2)), chunk 2 -> mask logits). */
int gpemsr_dcn_columns(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                       float* col, void* stream);

/* ---------------------------------------------------------------------------
 * GPEMSR-specific fused elementwise / reductions
 * ------------------------------------------------------------------------- */
/* model/GPEMSR.py:387-395: cosine similarity of co-located 16x16xC patches. a,b: [n][h][w][c]. */
int gpemsr_patch_cosine(const float* a, const float* b, int n, int h, int w, int c, float* out, void* stream);
/* ThreeDA temporal gating (model/GPEMSR.py:179-187): af[b][h][w][t*c] = aligned * sigmoid(sum_c emb*emb_ref) */
int gpemsr_temporal_gate(const float* aligned, const float* emb, const float* emb_ref, int b, int t, int hw, int c,
                         float* af, void* stream);
/* nn.Conv3d(t,t,1) over the frame axis + LeakyReLU (model/GPEMSR.py:191,193) on af[b][hw][t*c] */
int gpemsr_frame_mix_lrelu(const float* af, int64_t pixels, int t, int c, const float* m, const float* bias,
                           float* out, void* stream);
/* model/GPEMSR.py:219-221: out = feat*sigmoid(attn)*2 + attn_add + f2 + f3 */
int gpemsr_threeda_combine(const float* feat, const float* attn, const float* attn_add, const float* f2,
                           const float* f3, int64_t count, float* out, void* stream);
/* util/util.py:145-163 tensor2img: clamp[0,1] -> *255 -> round-half-even -> uint8 */
int gpemsr_tensor2img_u8(const float* x, int64_t count, uint8_t* out, void* stream);
/* strided channel copy (assembling small concat buffers): dst[p][0:c] = src[p][0:c] */
int gpemsr_copy_channels(const float* src, int src_ld, float* dst, int dst_ld, int64_t pixels, int c, void* stream);

/* image regrouping: dst image j = src image (j / div) * mul + add  (elems_per_image % 4 == 0).
 * replaces the x[:, i] / [:, center].clone() indexing of model/GPEMSR.py:325,427-437,175. */
int gpemsr_copy_images(const float* src, float* dst, int64_t n_dst, int64_t elems_per_image, int div, int mul, int add,
                       void* stream);
/* ---- stage-3 training loss, forward (train_stage3.py:343-359) ------------------------------
 * nn.MaxPool2d(2,2) of torchvision vgg19 (model/VGG.py:17-29: features 4, 9, 18, 27), NHWC, c % 4 == 0 */
int gpemsr_maxpool2(const float* x, int n, int h, int w, int c, int ld, float* out, int out_ld, void* stream);
/* ContextualLoss.forward input normalisation (model/contextual.py:222-224): (x - mean_c) / std_c, 3-channel NHWC;
 * mean3 / std3 are host pointers */
int gpemsr_normalize3(const float* x, int64_t pixels, int ld, const float* mean3, const float* std3, float* out, int out_ld,
                      void* stream);
/* compute_cosine_distance (model/contextual.py:115-127): per-channel mean of y over all pixels of all images
 * (workspace >= ceil(pixels/1024)*c floats) ... */
int gpemsr_cx_channel_mean(const float* y, int64_t pixels, int c, int ld, float* workspace, int64_t workspace_floats,
                           float* mean, void* stream);
/* ... and out = F.normalize(x - mean, p=2, dim=channel) per pixel.  The similarity GEMM S[n,i,j] = <x^_i, y^_j> is
 * gpemsr_conv2d (1x1, per-image weights = y^). */
int gpemsr_cx_center_normalize(const float* x, const float* mean, int64_t pixels, int c, int ld, float* out, int out_ld,
                               void* stream);
/* compute_relative_distance + compute_cx (model/contextual.py:103-112) on rows of S: dist = clamp(1-S,0);
 * cx = exp((1 - dist/(min_j dist + 1e-5))/h) / (sum_j + 1e-5).  cols % 4 == 0, <= 16384. */
int gpemsr_cx_rows(const float* sim, int64_t rows, int cols, float band_width, float* cx, void* stream);
/* contextual_loss (model/contextual.py:44-52): rmax[n,j] = max_i cx[n,i,j]; cw[n,j] = exp((1 - dist[n,i*,j])/h);
 * cx_image[n] = sum_j rmax*cw / sum_j cw; loss = mean_n -log(cx_image + 1e-5).
 * workspace >= 2 * n * ceil(rows/128) * cols floats. */
int gpemsr_cx_reduce(const float* cx, const float* sim, int n, int rows, int cols, float band_width, float* workspace,
                     int64_t workspace_floats, float* rmax, float* cw, float* cx_image, float* loss, void* stream);

/* image gather: dst image j = src image idx[j] (idx: int32 on the device).  Volume mode: the sliding
 * 5-slice windows of output_GPEMSR.py:54-128 pick their frames' cached per-frame features instead of recomputing them. */
int gpemsr_gather_images(const float* src, const int* idx, float* dst, int64_t n_dst, int64_t elems_per_image, void* stream);


/* ===========================================================================================================
 * Stage-3 training step, backward + optimizer (train_stage3.py:343-366: loss_total.backward(); optimizer_G.step()).
 * The reference gets these from torch.autograd / torch.optim; each entry point names the ATen backward it stands for.
 * Gradient outputs marked "+=" ACCUMULATE into caller-zeroed buffers (one contribution per consumer of a tensor).
 * Data gradients of convolutions are convolutions: gpemsr_conv2d with re-packed weights (stride-2 convs <-> the
 * transposed form), see gpemsr_amd/train.py.
 * ========================================================================================================= */
/* Weight gradient (aten::convolution_backward, grad_weight): dw[co][cin_off+ci][ky][kx] += sum_pixels dz * x.
 * x: [n][h][w][x_ld] (channels [0,cin) used), dz: [n][oh][ow][dz_ld] (channels [0,cout)); ksize 1|3, pad k/2,
 * stride 1|2|4; dw is the OIHW tensor [cout][cin_total][k][k].  ConvTranspose2d(k3,s2,p1,op1) weights [Cin][Cout][3][3]:
 * call with x := dOut (2h x 2w, Cout channels), dz := the layer input (Cin channels), stride 2.
 * ws: >= gpemsr_conv2d_wgrad_workspace(...) floats (fewer is legal: slabs get longer). f32 MFMA, deterministic. */
int64_t gpemsr_conv2d_wgrad_workspace(int cin, int cout, int ksize, int n, int oh, int ow);
int gpemsr_conv2d_wgrad(const float* x, int x_ld, int cin, const float* dz, int dz_ld, int cout, int n, int h, int w,
                        int oh, int ow, int ksize, int stride, float* ws, int64_t ws_floats, float* dw, int cin_total,
                        int cin_off, void* stream);
/* dz = dy * act'(y) with act' taken from the saved OUTPUT y (relu/lrelu/sigmoid/lrelu_sigmoid backward); h,w,c are the
 * conv-output geometry; pixel_shuffle = 1: dy/y are [n][2h][2w][c/4] and the nn.PixelShuffle(2) backward is applied. */
int gpemsr_act_bwd(const float* dy, int dy_ld, const float* y, int y_ld, int n, int h, int w, int c, int act,
                   int pixel_shuffle, float* dz, int dz_ld, void* stream);
/* db[c] += sum_pixels dz (grad_bias); ws >= min(pixels,512)*c floats */
int gpemsr_bias_grad(const float* dz, int64_t pixels, int c, int ld, float* ws, int64_t ws_floats, float* db, void* stream);
/* dst += alpha * src on channel slices (gradient of a residual add / fan-out accumulation) */
int gpemsr_axpy(const float* src, int src_ld, float* dst, int dst_ld, int64_t pixels, int c, float alpha, void* stream);
/* out = x * m[pixel] (the MPF mask multiply, model/GPEMSR.py:403-411, un-fused in training) and its backward:
 * dx += dy * m (dx may be NULL); dm[pixel] += sum_c dy * x */
int gpemsr_mul_pix(const float* x, int x_ld, const float* m, int64_t pixels, int c, float* out, int out_ld, void* stream);
int gpemsr_mul_pix_bwd(const float* dy, int dy_ld, const float* x, int x_ld, const float* m, int64_t pixels, int c,
                       float* dx, int dx_ld, float* dm, void* stream);
/* upsample_bilinear2d_backward: dx[n][h][w] += mul * sum of the (oh x ow) output gradients, gather form, same index rule
 * as gpemsr_bilinear */
int gpemsr_bilinear_bwd(const float* dy, int dy_ld, int n, int h, int w, int c, int oh, int ow, int align_corners, float mul,
                        float* dx, int dx_ld, void* stream);
/* torchvision deform_conv2d backward w.r.t. input (dx +=, float atomics; may be NULL), offsets and mask logits
 * (dom += in the conv_offset channel layout); dcol is the gradient of gpemsr_dcn_columns' output */
int gpemsr_dcn_columns_bwd(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                           const float* dcol, float* dx, int dx_ld, float* dom, int dom_ld, void* stream);
/* the same, BIT-STABLE: the scatter into dx accumulates 64-bit fixed-point integers with integer atomics (associative: the sum does not
 * depend on the order), scale 2^(38 - exponent of *dcol_absmax) with dcol_absmax = max |dcol| as a DEVICE scalar (e.g. torch's
 * deterministic amax); dx_fix: zero-initialised int64 workspace [n*h*w*c] (NULL together with dx); a second kernel adds the converted
 * sums into dx (+=, one writer per element).  What loss.backward() through torchvision.ops.deform_conv2d does with float atomics. */
int gpemsr_dcn_columns_bwd_det(const float* x, int n, int h, int w, int c, int ld, const float* om, int om_ld, int groups,
                               const float* dcol, const float* dcol_absmax, long long* dx_fix, float* dx, int dx_ld,
                               float* dom, int dom_ld, void* stream);
/* ThreeDA pieces (model/GPEMSR.py:179-221), all +=, dense c == 64 tensors */
int gpemsr_temporal_gate_bwd(const float* aligned, const float* emb, const float* emb_ref, const float* daf, int b, int t,
                             int hw, int c, float* d_aligned, float* d_emb, float* d_emb_ref, void* stream);
int gpemsr_frame_mix_lrelu_bwd(const float* af, const float* out, const float* dout, int64_t pixels, int t, int c,
                               const float* m, float* d_af, float* dm, float* dbias, float* ws, int64_t ws_floats, void* stream);
int gpemsr_pool3s2_maxavg_bwd(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld, float* dx,
                              int dx_ld, void* stream);
int gpemsr_threeda_combine_bwd(const float* feat, const float* attn, const float* dout, int64_t count, float* dfeat,
                               float* dattn, float* dadd, float* df2, float* df3, void* stream);
/* max_pool2d(2,2) backward (VGG, first maximum of a window wins as in ATen) */
int gpemsr_maxpool2_bwd(const float* x, int n, int h, int w, int c, int ld, const float* dy, int dy_ld, float* dx, int dx_ld,
                        void* stream);
/* backward of gpemsr_gather_images / gpemsr_copy_images: dtarget[i] += sum_{j: idx[j]==i} dsrc[j] (fixed order) */
int gpemsr_scatter_add_images(const float* dsrc, const int* idx, float* dtarget, int n_src, int n_dst, int64_t elems_per_image,
                              void* stream);
/* Codebook.forward (R:model/codebook.py:20-31) past the arg-min: zq = embedding[idx] -> zq [rows][zq_ld]; loss[0] =
 * mean((zq.detach() - z)^2) + beta * mean((zq - z.detach())^2); dz -= ... i.e. dz += grad_scale * 2 (z - zq) / (rows*dim),
 * dembedding[idx] += grad_scale * beta * 2 (zq - z) / (rows*dim) (float atomics; either may be NULL).  The straight-through decoder
 * input z + (zq - z).detach() has the value zq; the caller adds its gradient to dz.  ws >= 1024 floats.  (Stage-1 generator phase.) */
int gpemsr_vq_codebook_loss(const float* z, int z_ld, const float* embedding, const int32_t* idx, int64_t rows, int dim, float beta,
                            float grad_scale, float* dz, int dz_ld, float* dembedding, float* zq, int zq_ld, float* ws,
                            int64_t ws_floats, float* loss, void* stream);
/* torch.nn.L1Loss()(GT, SR) (mean) -> loss[0]; dsr += grad_scale * sign(sr - gt) / count (dsr may be NULL); ws >= 1024 */
int gpemsr_l1_loss(const float* sr, const float* gt, int64_t count, float grad_scale, float* dsr, float* ws, int64_t ws_floats,
                   float* loss, void* stream);
/* contextual_loss backward w.r.t. the similarity matrix (model/contextual.py:36-52 through autograd); scale = dL/d(cx_loss).
 * idx_ws: n*cols int32, coef_ws: 2*n*cols*(1 + ceil(rows/128)) floats; dsim [n][rows][cols] is written. */
int gpemsr_cx_backward(const float* sim, const float* cx, const float* rmax, const float* cw, const float* cx_image, int n,
                       int rows, int cols, float band_width, float scale, int32_t* idx_ws, float* coef_ws, float* dsim,
                       void* stream);
int gpemsr_cx_center_normalize_bwd(const float* x, const float* mean, const float* g, int64_t pixels, int c, int ld, int g_ld,
                                   float* dx, int dx_ld, void* stream);
/* 1-channel image -> the normalised 3-channel VGG input (train_stage3.py:356-358 + model/contextual.py:222-224) and back */
int gpemsr_gray_normalize3(const float* x, int64_t pixels, const float* mean3, const float* std3, float* out, void* stream);
int gpemsr_gray_normalize3_bwd(const float* g, int64_t pixels, const float* std3, float* dx, void* stream);
/* per-image transpose dst[n][cols][rows] = src[n][rows][cols] (y^T for the dS . y^ product) */
int gpemsr_transpose_images(const float* src, float* dst, int n, int rows, int cols, void* stream);
/* torch.optim.Adam.step (train_stage3.py:163,365; no amsgrad), `step` counts from 1 */
int gpemsr_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, void* stream);

/* ---- stage-2 (indexer) training step, train_stage2.py:351-366: what the VQGAN-style blocks add to the stage-3 set ----
 * GroupNorm(groups, eps)(+ReLU) backward (native_group_norm_backward + threshold_backward): x is the layer input, mean_rstd
 * the forward statistics [n][groups][2]; dx += ; dgamma/dbeta += (may both be NULL).  ws >= n*parts*c*2 + n*c*2 + n*groups*2
 * floats with parts = clamp(hw/64, 1, 64). */
int gpemsr_groupnorm_bwd(const float* x, int ld, const float* dy, int dy_ld, int n, int hw, int c, int groups,
                         const float* mean_rstd, const float* gamma, const float* beta, int relu, float* ws, int64_t ws_floats,
                         float* dx, int dx_ld, float* dgamma, float* dbeta, void* stream);
/* F.softmax(dim=-1) backward on rows, in place over dp: ds = p * (dp - sum_j dp*p)   (model/blocks.py:77) */
int gpemsr_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int cols, void* stream);
/* torch.nn.CrossEntropyLoss() (mean): loss[0]; row_loss[rows] scratch; dlogits (may be NULL) = grad_scale * (softmax - onehot) / rows */
int gpemsr_cross_entropy(const float* logits, const int32_t* target, int64_t rows, int cols, float grad_scale, float* row_loss,
                         float* loss, float* dlogits, void* stream);

/* ---------------------------------------------------------------------------
 * Adversarial phase of stage-1 training (R:train_stage1.py:300-345; PatchGAN discriminator R:model/discriminator.py:9-32).  fp32 NHWC.
 * ------------------------------------------------------------------------- */
/* nn.Conv2d(k = 4, stride 1 | 2, padding 0) as a GEMM: col[(n*oh + oy)*ow + ox][kp], k = (ky*4 + kx)*c + ci, zero for 16c <= k < kp;
 * oh = (h - 4)/stride + 1.  The product with W2[cout][k] is gpemsr_conv2d's 1x1 form, the weight gradient gpemsr_conv2d_wgrad's. */
int gpemsr_im2col4(const float* x, int n, int h, int w, int c, int ld, int stride, float* col, int kp, void* stream);
/* its adjoint: dx[n][iy][ix][ci] (+)= sum of the column-gradient entries that read that pixel (gather form, deterministic) */
int gpemsr_col2im4(const float* dcol, int n, int h, int w, int c, int stride, int kp, float* dx, int dx_ld, int accumulate, void* stream);
/* nn.LeakyReLU(slope) and its backward from the activation's OUTPUT: dx (+)= dy * (y > 0 ? 1 : slope) */
int gpemsr_lrelu_slope(const float* x, int64_t count, float slope, float* y, void* stream);
int gpemsr_lrelu_slope_bwd(const float* dy, const float* y, int64_t count, float slope, float* dx, int accumulate, void* stream);
/* out[0] (+)= scale * sum(x) (square = 0) or scale * sum(x^2) (square = 1): torch.mean / the R1 penalty's sum of squares; one workgroup */
int gpemsr_sum_scaled(const float* x, int64_t count, float scale, int square, float* out, int accumulate, void* stream);
/* Second-order terms of nn.InstanceNorm2d (no affine) for the R1 penalty (torch.autograd.grad(..., create_graph=True) through the
 * discriminator, R:train_stage1.py:360-372): x the layer input [n][hw][c], mean_rstd [n][c][2] its forward statistics
 * (gpemsr_groupnorm_stats with groups = c), dy the gradient that entered the layer's backward, g = dL/d(dx) the gradient arriving at that
 * backward's RESULT -> gdy = dL/d(dy), gx (+)= dL/dx.  Formulas in csrc/stage1_adv.hip. */
int gpemsr_instnorm_bwd_bwd(const float* x, const float* dy, const float* g, const float* mean_rstd, int n, int hw, int c, float* gx, float* gdy,
                            int accumulate_gx, void* stream);

/* ---------------------------------------------------------------------------
 * PNG edges of the inference loop on the device.  Byte / integer kernels, csrc/png.hip.
 * ------------------------------------------------------------------------- */
/* R:output_GPEMSR.py:95 `cv2.imwrite(path, output)`: n 8-bit grayscale images (img + i*img_stride, rows row_stride bytes apart) -> n complete
 * PNG files at out + i*out_stride, gpemsr_png_gray8_size(h, w) bytes each: signature, IHDR, ONE IDAT chunk whose zlib stream uses stored
 * deflate blocks (filter type 0 on every scanline), IEND; Adler-32 and both CRC-32s computed on the device.  Any PNG reader decodes the
 * same pixels cv2 would have written.  workspace: gpemsr_png_encode_workspace(n, h, w) bytes, 8-byte aligned. */
int64_t gpemsr_png_gray8_size(int h, int w);
int64_t gpemsr_png_encode_workspace(int n, int h, int w);
int gpemsr_png_encode_gray8(const uint8_t* img, int n, int h, int w, int64_t img_stride, int row_stride, uint8_t* out, int64_t out_stride,
                            void* workspace, int64_t workspace_bytes, void* stream);
/* The same files with a COMPRESSED zlib stream (csrc/png_huff.hip): one dynamic-Huffman deflate block of literals per image, scanline filter
 * None or Sub (whichever has the lower order-0 entropy).  File sizes differ per image: sizes[i] bytes are valid at out + i*out_stride;
 * out_stride >= gpemsr_png_huff_capacity(h, w) (a multiple of 16), out 16-byte aligned, workspace gpemsr_png_huff_workspace(n, h, w) bytes. */
int64_t gpemsr_png_huff_capacity(int h, int w);
int64_t gpemsr_png_huff_workspace(int n, int h, int w);
int gpemsr_png_encode_gray8_huff(const uint8_t* img, int n, int h, int w, int64_t img_stride, int row_stride, uint8_t* out, int64_t out_stride,
                                 int64_t* sizes, void* workspace, int64_t workspace_bytes, void* stream);
/* R:data/util.py:75-88 `cv2.imread(path, IMREAD_UNCHANGED).astype(float32) / 255` for n non-interlaced 8-bit grayscale PNGs of one size:
 * idat = the files' IDAT payloads back to back (image i: bytes offsets[i] .. offsets[i+1], a complete zlib stream; the host reads chunk
 * lengths and IHDR, 50 bytes per file), raw = scratch of n*h*(w+1) bytes, out[n][h][w] = pixel / divisor (255; IEEE division, as numpy's).  One wave per image: lane 0
 * inflates (stored, fixed and dynamic Huffman blocks; tables and the 32 KB window in LDS), all lanes check Adler-32 and undo the scanline filters.  status[i]: 0 ok, 1 bad zlib
 * header, 2 bad block, 3 bad Huffman table, 4 bad symbol / distance, 5 size mismatch, 6 input exhausted, 7 Adler-32 mismatch, 8 bad filter, 9 wider than 16,384 pixels (two scanlines live in LDS). */
int gpemsr_png_decode_gray8(const uint8_t* idat, const int64_t* offsets, int n, int h, int w, uint8_t* raw, float* out, float divisor,
                            int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPEMSR_HIP_H */
