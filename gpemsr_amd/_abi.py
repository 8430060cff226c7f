"""ctypes binding of libgpemsr_hip.so (include/gpemsr_hip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  (Build it with ``python -m gpemsr_amd.build``
or ``__graft_entry__.build()``.)
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.environ.get("GPEMSR_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                               "libgpemsr_hip.so")   # env override: A/B kernel builds
_lib = None

MAX_SRC = 4
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_LRELU_SIGMOID = 0, 1, 2, 3, 4

# every symbol include/gpemsr_hip.h declares (checked by tests/test_host_cpu.py::test_c_abi_library_exports_every_declared_symbol)
SYMBOLS = [
    "gpemsr_abi_version", "gpemsr_last_error", "gpemsr_device_info", "gpemsr_conv2d", "gpemsr_conv2d_split", "gpemsr_split_pack_rows", "gpemsr_conv2d_direct", "gpemsr_conv2d_stem1",
    "gpemsr_groupnorm_stats", "gpemsr_groupnorm_apply", "gpemsr_softmax_rows", "gpemsr_softmax_rows_ld", "gpemsr_argmax_rows",
    "gpemsr_gather_rows", "gpemsr_bilinear", "gpemsr_avgpool2", "gpemsr_pool3s2_maxavg", "gpemsr_spynet_prep",
    "gpemsr_dcn_columns", "gpemsr_patch_cosine", "gpemsr_temporal_gate", "gpemsr_frame_mix_lrelu",
    "gpemsr_png_gray8_size", "gpemsr_png_encode_workspace", "gpemsr_png_encode_gray8", "gpemsr_png_decode_gray8",
    "gpemsr_png_huff_capacity", "gpemsr_png_huff_workspace", "gpemsr_png_encode_gray8_huff",
    "gpemsr_threeda_combine", "gpemsr_tensor2img_u8", "gpemsr_copy_channels", "gpemsr_copy_images",
    "gpemsr_gather_images", "gpemsr_maxpool2", "gpemsr_normalize3", "gpemsr_cx_channel_mean", "gpemsr_cx_center_normalize",
    "gpemsr_cx_rows", "gpemsr_cx_reduce",
    # stage-3 training step: backward + optimizer
    "gpemsr_conv2d_wgrad_workspace", "gpemsr_conv2d_wgrad", "gpemsr_act_bwd", "gpemsr_bias_grad", "gpemsr_axpy", "gpemsr_mul_pix",
    "gpemsr_mul_pix_bwd", "gpemsr_bilinear_bwd", "gpemsr_dcn_columns_bwd", "gpemsr_dcn_columns_bwd_det", "gpemsr_temporal_gate_bwd", "gpemsr_frame_mix_lrelu_bwd",
    "gpemsr_pool3s2_maxavg_bwd", "gpemsr_threeda_combine_bwd", "gpemsr_maxpool2_bwd", "gpemsr_scatter_add_images", "gpemsr_l1_loss",
    "gpemsr_cx_backward", "gpemsr_cx_center_normalize_bwd", "gpemsr_gray_normalize3", "gpemsr_gray_normalize3_bwd",
    "gpemsr_transpose_images", "gpemsr_adam_step",
    # stage-2 (indexer) training step
    "gpemsr_groupnorm_bwd", "gpemsr_softmax_bwd_rows", "gpemsr_cross_entropy",
    # bf16 data path
    "gpemsr_conv2d_bf16", "gpemsr_conv2d_bf16_gn_parts", "gpemsr_conv2d_bf16_axf_ok", "gpemsr_groupnorm_scale_shift", "gpemsr_pack_rows_bf16_ex", "gpemsr_flash_attention_bf16", "gpemsr_maxpool2_bf16", "gpemsr_im2col4", "gpemsr_col2im4", "gpemsr_lrelu_slope", "gpemsr_lrelu_slope_bwd", "gpemsr_sum_scaled", "gpemsr_instnorm_bwd_bwd", "gpemsr_groupnorm_stats_bf16", "gpemsr_groupnorm_finish", "gpemsr_groupnorm_apply_bf16",
    "gpemsr_softmax_rows_bf16", "gpemsr_gather_rows_bf16", "gpemsr_pack_rows_bf16", "gpemsr_cast_f32_bf16", "gpemsr_cast_bf16_f32",
    "gpemsr_bilinear_bf16", "gpemsr_pool3s2_maxavg_bf16", "gpemsr_spynet_prep_bf16", "gpemsr_dcn_columns_bf16", "gpemsr_dcn_conv_bf16", "gpemsr_patch_cosine_bf16",
    "gpemsr_temporal_gate_bf16", "gpemsr_frame_mix_lrelu_bf16", "gpemsr_threeda_combine_bf16", "gpemsr_copy_channels_bf16",
    "gpemsr_copy_channels_f32_bf16", "gpemsr_conv2d_stem1_bf16", "gpemsr_conv2d_direct_bf16", "gpemsr_vgg_mask_bf16",
    "gpemsr_conv_c64_cout1_bf16", "gpemsr_upconv_out_c64_bf16", "gpemsr_conv7_c16_cout2_bf16", "gpemsr_conv_c64_cout1_f32", "gpemsr_upconv_out_c64_f32", "gpemsr_vq_codebook_loss", "gpemsr_conv7_c16_cout2_f32", "gpemsr_split_f32_bf16x2", "gpemsr_conv2d_gn_parts", "gpemsr_patch_cosine_finish",
    "gpemsr_conv2d_kernel_name", "gpemsr_conv2d_bf16_kernel_name", "gpemsr_conv7_c32_cout16_bf16", "gpemsr_conv7_c8_cout32_bf16", "gpemsr_conv2d_bf16_rowmax_parts", "gpemsr_rowmax_finish",
]


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("c", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("nsrc", C.c_int32),
        ("src", Src * MAX_SRC),
        ("src_image_stride", C.c_int64 * MAX_SRC),
        ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("transposed", C.c_int32),
        ("weight", C.c_void_p), ("weight_image_stride", C.c_int64),
        ("bias", C.c_void_p), ("act", C.c_int32),
        ("residual", C.c_void_p), ("res_ld", C.c_int32),
        ("pixmul", C.c_void_p), ("pixel_shuffle", C.c_int32),
        ("out", C.c_void_p), ("out_ld", C.c_int32),
        ("gn_partials", C.c_void_p), ("cos_partials", C.c_void_p),
        ("a_scale", C.c_void_p), ("a_shift", C.c_void_p), ("a_relu", C.c_int32),
    ]


class ConvDesc16(C.Structure):
    """gpemsr_conv16_desc (include/gpemsr_hip.h), field by field."""
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("nsrc", C.c_int32),
        ("src", Src * MAX_SRC),
        ("src_image_stride", C.c_int64 * MAX_SRC),
        ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("transposed", C.c_int32),
        ("weight", C.c_void_p), ("weight_image_stride", C.c_int64),
        ("bias", C.c_void_p), ("act", C.c_int32),
        ("residual", C.c_void_p), ("res_ld", C.c_int32), ("res_f32", C.c_int32),
        ("pixmul", C.c_void_p), ("pixel_shuffle", C.c_int32), ("kpack", C.c_int32),
        ("out", C.c_void_p), ("out_ld", C.c_int32), ("out_f32", C.c_int32),
        ("out32", C.c_void_p), ("out32_ld", C.c_int32),
        ("gn_partials", C.c_void_p), ("variant", C.c_int32), ("gn_cpg", C.c_int32),
        ("a_scale", C.c_void_p), ("a_shift", C.c_void_p), ("a_relu", C.c_int32), ("weight_forms", C.c_int32), ("rowmax", C.c_void_p),
    ]


def lib_path() -> str:
    return _LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"gpemsr_amd: HIP kernel library not found at {_LIB_PATH}. Build it first "
            "(python -m gpemsr_amd.build). There is no CPU fallback for the product path.")
    lib = C.CDLL(_LIB_PATH)
    lib.gpemsr_last_error.restype = C.c_char_p
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise RuntimeError(f"gpemsr_amd: {_LIB_PATH} does not export {s}")
    v = lib.gpemsr_abi_version()
    if v != 1:
        raise RuntimeError(f"gpemsr_amd: ABI version {v} != 1")
    p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.gpemsr_conv2d.argtypes = [C.POINTER(ConvDesc), p]
    lib.gpemsr_conv2d_gn_parts.argtypes = [C.POINTER(ConvDesc)]
    lib.gpemsr_conv2d_kernel_name.argtypes = [C.POINTER(ConvDesc), C.c_char_p, C.c_int]
    lib.gpemsr_patch_cosine_finish.argtypes = [p, i32, i32, i32, p, p]
    lib.gpemsr_conv2d_split.argtypes = [C.POINTER(ConvDesc), p, i64, i32, p]
    lib.gpemsr_conv2d_direct.argtypes = [p, i32, i32, i32, i32, i32, p, p, i32, i32, i32, i32, p, i32, p, i32, p]
    lib.gpemsr_conv2d_stem1.argtypes = [p, i32, i32, i32, p, p, i32, i32, p, i32, p]
    lib.gpemsr_groupnorm_stats.argtypes = [p, i32, i32, i32, i32, i32, f32, p, i32, p, p]
    lib.gpemsr_groupnorm_apply.argtypes = [p, i32, i32, i32, i32, i32, p, p, p, i32, p, i32, p, i32, p]
    lib.gpemsr_softmax_rows.argtypes = [p, i64, i32, p]
    lib.gpemsr_softmax_rows_ld.argtypes = [p, i64, i32, i32, p]
    lib.gpemsr_argmax_rows.argtypes = [p, i64, i32, p, p]
    lib.gpemsr_gather_rows.argtypes = [p, i32, p, i64, p, i32, p]
    lib.gpemsr_bilinear.argtypes = [p, i32, i32, i32, i32, i32, i32, i32, i32, f32, p, i32, p]
    lib.gpemsr_avgpool2.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_pool3s2_maxavg.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_spynet_prep.argtypes = [p, p, p, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), p, p, i32, p]
    lib.gpemsr_dcn_columns.argtypes = [p, i32, i32, i32, i32, i32, p, i32, i32, p, p]
    lib.gpemsr_patch_cosine.argtypes = [p, p, i32, i32, i32, i32, p, p]
    lib.gpemsr_temporal_gate.argtypes = [p, p, p, i32, i32, i32, i32, p, p]
    lib.gpemsr_frame_mix_lrelu.argtypes = [p, i64, i32, i32, p, p, p, p]
    lib.gpemsr_threeda_combine.argtypes = [p, p, p, p, p, i64, p, p]
    lib.gpemsr_tensor2img_u8.argtypes = [p, i64, p, p]
    lib.gpemsr_png_gray8_size.argtypes = [i32, i32]
    lib.gpemsr_png_gray8_size.restype = C.c_int64
    lib.gpemsr_png_encode_workspace.argtypes = [i32, i32, i32]
    lib.gpemsr_png_encode_workspace.restype = C.c_int64
    lib.gpemsr_png_encode_gray8.argtypes = [p, i32, i32, i32, i64, i32, p, i64, p, i64, p]
    lib.gpemsr_png_decode_gray8.argtypes = [p, p, i32, i32, i32, p, p, f32, p, p]
    lib.gpemsr_png_huff_capacity.argtypes = [i32, i32]
    lib.gpemsr_png_huff_capacity.restype = C.c_int64
    lib.gpemsr_png_huff_workspace.argtypes = [i32, i32, i32]
    lib.gpemsr_png_huff_workspace.restype = C.c_int64
    lib.gpemsr_png_encode_gray8_huff.argtypes = [p, i32, i32, i32, i64, i32, p, i64, p, p, i64, p]
    lib.gpemsr_copy_channels.argtypes = [p, i32, p, i32, i64, i32, p]
    lib.gpemsr_copy_images.argtypes = [p, p, i64, i64, i32, i32, i32, p]
    lib.gpemsr_split_pack_rows.argtypes = [p, i32, i32, i32, i32, i64, p, p]
    lib.gpemsr_gather_images.argtypes = [p, p, p, i64, i64, p]
    lib.gpemsr_maxpool2.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p]
    f3 = C.POINTER(C.c_float)
    lib.gpemsr_normalize3.argtypes = [p, i64, i32, f3, f3, p, i32, p]
    lib.gpemsr_cx_channel_mean.argtypes = [p, i64, i32, i32, p, i64, p, p]
    lib.gpemsr_cx_center_normalize.argtypes = [p, p, i64, i32, i32, p, i32, p]
    lib.gpemsr_cx_rows.argtypes = [p, i64, i32, C.c_float, p, p]
    lib.gpemsr_cx_reduce.argtypes = [p, p, i32, i32, i32, C.c_float, p, i64, p, p, p, p, p]
    lib.gpemsr_conv2d_wgrad_workspace.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.gpemsr_conv2d_wgrad_workspace.restype = C.c_int64
    lib.gpemsr_conv2d_wgrad.argtypes = [p, i32, i32, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, i64, p, i32, i32, p]
    lib.gpemsr_act_bwd.argtypes = [p, i32, p, i32, i32, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_bias_grad.argtypes = [p, i64, i32, i32, p, i64, p, p]
    lib.gpemsr_axpy.argtypes = [p, i32, p, i32, i64, i32, f32, p]
    lib.gpemsr_mul_pix.argtypes = [p, i32, p, i64, i32, p, i32, p]
    lib.gpemsr_mul_pix_bwd.argtypes = [p, i32, p, i32, p, i64, i32, p, i32, p, p]
    lib.gpemsr_bilinear_bwd.argtypes = [p, i32, i32, i32, i32, i32, i32, i32, i32, f32, p, i32, p]
    lib.gpemsr_dcn_columns_bwd.argtypes = [p, i32, i32, i32, i32, i32, p, i32, i32, p, p, i32, p, i32, p]
    lib.gpemsr_dcn_columns_bwd_det.argtypes = [p, i32, i32, i32, i32, i32, p, i32, i32, p, p, p, p, i32, p, i32, p]
    lib.gpemsr_temporal_gate_bwd.argtypes = [p, p, p, p, i32, i32, i32, i32, p, p, p, p]
    lib.gpemsr_frame_mix_lrelu_bwd.argtypes = [p, p, p, i64, i32, i32, p, p, p, p, p, i64, p]
    lib.gpemsr_pool3s2_maxavg_bwd.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p, i32, p]
    lib.gpemsr_threeda_combine_bwd.argtypes = [p, p, p, i64, p, p, p, p, p, p]
    lib.gpemsr_maxpool2_bwd.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p, i32, p]
    lib.gpemsr_scatter_add_images.argtypes = [p, p, p, i32, i32, i64, p]
    lib.gpemsr_l1_loss.argtypes = [p, p, i64, f32, p, p, i64, p, p]
    lib.gpemsr_cx_backward.argtypes = [p, p, p, p, p, i32, i32, i32, f32, f32, p, p, p, p]
    lib.gpemsr_cx_center_normalize_bwd.argtypes = [p, p, p, i64, i32, i32, i32, p, i32, p]
    lib.gpemsr_gray_normalize3.argtypes = [p, i64, f3, f3, p, p]
    lib.gpemsr_gray_normalize3_bwd.argtypes = [p, i64, f3, p, p]
    lib.gpemsr_transpose_images.argtypes = [p, p, i32, i32, i32, p]
    lib.gpemsr_adam_step.argtypes = [p, p, p, p, i64, f32, f32, f32, f32, f32, i32, p]
    lib.gpemsr_groupnorm_bwd.argtypes = [p, i32, p, i32, i32, i32, i32, i32, p, p, p, i32, p, i64, p, i32, p, p, p]
    lib.gpemsr_softmax_bwd_rows.argtypes = [p, p, i64, i32, p]
    lib.gpemsr_cross_entropy.argtypes = [p, p, i64, i32, f32, p, p, p, p]
    lib.gpemsr_device_info.argtypes = [C.c_char_p, i32, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    lib.gpemsr_conv2d_bf16.argtypes = [C.POINTER(ConvDesc16), p]
    lib.gpemsr_conv2d_bf16_gn_parts.argtypes = [C.POINTER(ConvDesc16)]
    lib.gpemsr_conv2d_bf16_kernel_name.argtypes = [C.POINTER(ConvDesc16), C.c_char_p, C.c_int]
    lib.gpemsr_conv2d_bf16_rowmax_parts.argtypes = [C.POINTER(ConvDesc16)]
    lib.gpemsr_rowmax_finish.argtypes = [p, C.c_longlong, i32, p, p]
    lib.gpemsr_groupnorm_stats_bf16.argtypes = [p, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_groupnorm_finish.argtypes = [p, i32, i32, i32, i32, i32, f32, p, p]
    lib.gpemsr_groupnorm_scale_shift.argtypes = [p, p, p, i32, i32, i32, p, p, p]
    lib.gpemsr_conv2d_bf16_axf_ok.argtypes = [C.POINTER(ConvDesc16)]
    lib.gpemsr_groupnorm_apply_bf16.argtypes = [p, i32, i32, i32, i32, i32, p, p, p, i32, p, i32, p, i32, p]
    lib.gpemsr_softmax_rows_bf16.argtypes = [p, i32, i64, i32, i32, p, i32, p]
    lib.gpemsr_gather_rows_bf16.argtypes = [p, i32, p, i64, p, i32, p]
    lib.gpemsr_pack_rows_bf16.argtypes = [p, i32, i32, i32, i32, i64, p, p]
    lib.gpemsr_im2col4.argtypes = [p, i32, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_col2im4.argtypes = [p, i32, i32, i32, i32, i32, i32, p, i32, i32, p]
    lib.gpemsr_lrelu_slope.argtypes = [p, i64, f32, p, p]
    lib.gpemsr_lrelu_slope_bwd.argtypes = [p, p, i64, f32, p, i32, p]
    lib.gpemsr_sum_scaled.argtypes = [p, i64, f32, i32, p, i32, p]
    lib.gpemsr_instnorm_bwd_bwd.argtypes = [p, p, p, p, i32, i32, i32, p, p, i32, p]
    lib.gpemsr_maxpool2_bf16.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_pack_rows_bf16_ex.argtypes = [p, i32, i32, i32, i32, i64, p, i32, p]
    lib.gpemsr_flash_attention_bf16.argtypes = [p, i32, p, p, p, i32, i32, i32, p, i32, p]
    lib.gpemsr_cast_f32_bf16.argtypes = [p, i64, i32, i32, p, i32, p]
    lib.gpemsr_cast_bf16_f32.argtypes = [p, i64, i32, i32, p, i32, p]
    lib.gpemsr_split_f32_bf16x2.argtypes = [p, i64, i32, i32, p, i32, p, i32, p]
    lib.gpemsr_bilinear_bf16.argtypes = [p, i32, i32, i32, i32, i32, i32, i32, i32, f32, p, i32, p]
    lib.gpemsr_pool3s2_maxavg_bf16.argtypes = [p, i32, i32, i32, i32, i32, p, i32, p]
    lib.gpemsr_spynet_prep_bf16.argtypes = [p, p, p, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), p, p, p]
    lib.gpemsr_dcn_columns_bf16.argtypes = [p, i32, i32, i32, i32, i32, p, i32, i32, p, p]
    lib.gpemsr_dcn_conv_bf16.argtypes = [p, i32, i32, i32, i32, p, i32, p, p, i32, p, i32, p]
    lib.gpemsr_patch_cosine_bf16.argtypes = [p, p, i32, i32, i32, i32, p, p]
    lib.gpemsr_temporal_gate_bf16.argtypes = [p, p, p, i32, i32, i32, i32, p, p]
    lib.gpemsr_frame_mix_lrelu_bf16.argtypes = [p, i64, i32, i32, p, p, p, p]
    lib.gpemsr_threeda_combine_bf16.argtypes = [p, p, p, p, p, i64, p, p]
    lib.gpemsr_copy_channels_bf16.argtypes = [p, i32, p, i32, i64, i32, p]
    lib.gpemsr_copy_channels_f32_bf16.argtypes = [p, i32, p, i32, i64, i32, p]
    lib.gpemsr_conv2d_stem1_bf16.argtypes = [p, i32, i32, i32, p, p, i32, i32, p, i32, p]
    lib.gpemsr_vgg_mask_bf16.argtypes = [p, p, i32, i32, i32, i32, p, p, p, p, p, p]
    lib.gpemsr_conv2d_direct_bf16.argtypes = [p, i32, i32, i32, i32, i32, i32, p, p, i32, i32, i32, i32, p, i32, p, i32, i32, p]
    lib.gpemsr_conv_c64_cout1_bf16.argtypes = [p, i32, i32, i32, i32, p, p, i32, p, i32, p, i32, p, p]
    lib.gpemsr_upconv_out_c64_bf16.argtypes = [p, i32, i32, i32, i32, p, p, p, i32, p]
    lib.gpemsr_conv7_c16_cout2_bf16.argtypes = [p, i32, i32, i32, i32, p, p, p, i32, p, i32, p]
    lib.gpemsr_conv7_c32_cout16_bf16.argtypes = [p, i32, i32, i32, i32, p, p, i32, p, i32, p]
    lib.gpemsr_conv7_c8_cout32_bf16.argtypes = [p, i32, i32, i32, i32, p, p, i32, p, i32, p]
    lib.gpemsr_conv7_c16_cout2_f32.argtypes = [p, i32, i32, i32, i32, p, p, p, i32, p, i32, p]
    lib.gpemsr_conv_c64_cout1_f32.argtypes = [p, i32, i32, i32, i32, p, p, i32, p, i32, p, i32, p, p]
    lib.gpemsr_upconv_out_c64_f32.argtypes = [p, i32, i32, i32, i32, p, p, p, i32, p]
    lib.gpemsr_vq_codebook_loss.argtypes = [p, i32, p, p, i64, i32, f32, f32, p, i32, p, p, i32, p, i64, p, p]
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gpemsr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"gpemsr_amd: {what} failed (code {rc}): {msg}")
