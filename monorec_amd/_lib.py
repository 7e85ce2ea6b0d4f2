"""ctypes binding of libmonorec_hip.so (include/monorec_hip.h).

There is deliberately no fallback: if the HIP library is missing or does not export the whole
ABI, importing the kernels raises - the product path never silently runs on PyTorch/CPU ops.
"""
import ctypes
import os

MR_MAX_SOURCES = 3
MR_MAX_FRAMES = 8

ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_SIGMOID, ACT_ABS_TANH_AFFINE = range(5)
IN_DIRECT, IN_UPSAMPLE2, IN_MAXPOOL2 = range(3)
TF_NONE, TF_RESNET_NORM = range(2)

_c_float_p = ctypes.POINTER(ctypes.c_float)


class ConvDesc(ctypes.Structure):
    """mirror of `mr_conv_desc` (include/monorec_hip.h) - field order and types must match."""
    _fields_ = [
        ("src", ctypes.c_void_p * MR_MAX_SOURCES),
        ("src_channels", ctypes.c_int32 * MR_MAX_SOURCES),
        ("num_src", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("src_h", ctypes.c_int32), ("src_w", ctypes.c_int32),
        ("in_mode", ctypes.c_int32),
        ("in_transform", ctypes.c_int32),
        ("kh", ctypes.c_int32), ("kw", ctypes.c_int32),
        ("stride_h", ctypes.c_int32), ("stride_w", ctypes.c_int32),
        ("pad_top", ctypes.c_int32), ("pad_left", ctypes.c_int32),
        ("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32),
        ("dst", ctypes.c_void_p),
        ("out_channels", ctypes.c_int32),
        ("dst_total_channels", ctypes.c_int32), ("dst_channel_offset", ctypes.c_int32),
        ("dst_plane_h", ctypes.c_int32), ("dst_plane_w", ctypes.c_int32),
        ("out_step_h", ctypes.c_int32), ("out_step_w", ctypes.c_int32),
        ("out_off_h", ctypes.c_int32), ("out_off_w", ctypes.c_int32),
        ("packed_weights", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("activation", ctypes.c_int32),
        ("act_p0", ctypes.c_float), ("act_p1", ctypes.c_float),
        ("cout_blocks_per_wg", ctypes.c_int32),
        ("pixel_blocks_per_wave", ctypes.c_int32),
        ("chunk_channels", ctypes.c_int32),
        ("split_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p),
        ("num_phases", ctypes.c_int32),
        ("phase_weights", ctypes.c_void_p * 4),
        ("phase_pad_top", ctypes.c_int32 * 4), ("phase_pad_left", ctypes.c_int32 * 4),
        ("phase_out_off_h", ctypes.c_int32 * 4), ("phase_out_off_w", ctypes.c_int32 * 4),
        ("waves_per_wg", ctypes.c_int32),
        ("compute_dtype", ctypes.c_int32),
        ("phase_kh", ctypes.c_int32 * 4), ("phase_kw", ctypes.c_int32 * 4),
        ("k_split_waves", ctypes.c_int32),
    ]


class WinoDesc(ctypes.Structure):
    """mirror of `mr_wino_desc` (include/monorec_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p * MR_MAX_SOURCES), ("src_channels", ctypes.c_int32 * MR_MAX_SOURCES), ("num_src", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
                ("dst", ctypes.c_void_p), ("out_channels", ctypes.c_int32),
                ("packed_weights", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("activation", ctypes.c_int32), ("act_p0", ctypes.c_float), ("cout_blocks_per_wave", ctypes.c_int32),
                ("variant", ctypes.c_int32),
                ("src_row_pitch", ctypes.c_int32), ("src_plane_floats", ctypes.c_int32), ("dst_split_columns", ctypes.c_int32)]


LAYOUT_F32_NCHW, LAYOUT_BF16_B8 = 0, 1


class B8ConvDesc(ctypes.Structure):
    """mirror of `mr_b8_conv_desc` (include/monorec_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p * MR_MAX_SOURCES), ("src_channels", ctypes.c_int32 * MR_MAX_SOURCES), ("src_layout", ctypes.c_int32 * MR_MAX_SOURCES),
                ("num_src", ctypes.c_int32), ("batch", ctypes.c_int32), ("src_h", ctypes.c_int32), ("src_w", ctypes.c_int32),
                ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride_h", ctypes.c_int32), ("stride_w", ctypes.c_int32),
                ("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32),
                ("dst", ctypes.c_void_p), ("dst_layout", ctypes.c_int32), ("out_channels", ctypes.c_int32),
                ("dst_plane_h", ctypes.c_int32), ("dst_plane_w", ctypes.c_int32), ("out_step_h", ctypes.c_int32), ("out_step_w", ctypes.c_int32),
                ("bias", ctypes.c_void_p), ("activation", ctypes.c_int32), ("act_p0", ctypes.c_float),
                ("cout_blocks_per_wg", ctypes.c_int32), ("pixel_blocks_per_wave", ctypes.c_int32), ("waves_per_wg", ctypes.c_int32),
                ("num_phases", ctypes.c_int32),
                ("phase_weights", ctypes.c_void_p * 4), ("phase_kh", ctypes.c_int32 * 4), ("phase_kw", ctypes.c_int32 * 4),
                ("phase_pad_top", ctypes.c_int32 * 4), ("phase_pad_left", ctypes.c_int32 * 4),
                ("phase_out_off_h", ctypes.c_int32 * 4), ("phase_out_off_w", ctypes.c_int32 * 4)]


MR_MAX_COPY_SEGMENTS = 24
MR_ABI_VERSION = 19            # include/monorec_hip.h


class CopySegment(ctypes.Structure):
    """mirror of `mr_copy_segment` (include/monorec_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_int64)]


class LaunchItem(ctypes.Structure):
    """mirror of `mr_launch_item` (include/monorec_hip.h)."""
    _fields_ = [("kind", ctypes.c_int32), ("arg", ctypes.c_int32), ("desc", ctypes.c_void_p)]


LAUNCH_CONV2D, LAUNCH_WINO3X3, LAUNCH_WINO_T, LAUNCH_WINO_1D, LAUNCH_UPCONV, LAUNCH_COOKTOOM_1D, LAUNCH_WINO44, LAUNCH_CONV_B8, LAUNCH_WINO44S, LAUNCH_WINO44W = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9


class HeadDesc(ctypes.Structure):
    """mirror of `mr_head_desc` (include/monorec_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("batch", ctypes.c_int32), ("channels", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32)]


MR_MAX_HEADS = 4

# every symbol include/monorec_hip.h declares: (restype, argtypes)
ABI = {
    "mr_conv_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_conv_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv_packed_weight_floats_bf16": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_conv_packed_weight_floats_bf16x3": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_conv_pack_weights_bf16x3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_void_p]),
    "mr_conv_pack_weights_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv2d_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(ConvDesc)]),
    "mr_conv2d_f32": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "mr_wino_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32]),
    "mr_wino_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv3x3_winograd_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_conv3x3_winograd_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_wino_t_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32]),
    "mr_wino_t_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_void_p]),
    "mr_convt4x4s2_winograd_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_convt4x4s2_winograd_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_cost_volume_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_float, _c_float_p,
                                          ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_cost_volume_mode_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_float, _c_float_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                               ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_cost_volume_relaxed_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_float, _c_float_p, ctypes.c_int32, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_cost_volume_tiled_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_float, _c_float_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_cost_volume_patch_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_float, _c_float_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_maxpool3x3s2_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_void_p]),
    "mr_maxpool2x2_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_void_p]),
    "mr_resnet_normalize_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "mr_exact_const_division": (ctypes.c_int, [ctypes.c_float]),
    "mr_run_launches": (ctypes.c_int, [ctypes.POINTER(LaunchItem), ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "mr_upconv_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mr_upconv2x2_winograd_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_wino1d_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32]),
    "mr_wino1d_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv1d3_winograd_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_conv1d3_winograd_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_int32, ctypes.c_void_p]),
    "mr_cooktoom1d_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_cooktoom1d_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv1d_cooktoom_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_conv1d_cooktoom_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mr_wino_t_packed_weight_floats_tail": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    "mr_wino_t_pack_weights_tail_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_void_p]),
    "mr_wino_packed_weight_floats_tail": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    "mr_wino_pack_weights_tail_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_void_p]),
    "mr_b8_packed_weight_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_b8_pack_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv2d_b8_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(B8ConvDesc)]),
    "mr_conv2d_b8": (ctypes.c_int, [ctypes.POINTER(B8ConvDesc), ctypes.c_void_p]),
    "mr_pool2x2_framemax_b8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_void_p]),
    "mr_max_over_frames_b8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]),
    "mr_f32_nchw_to_b8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]),
    "mr_b8_to_f32_nchw": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]),
    "mr_cost_volume_b8_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _c_float_p,
                                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_cost_volume_b8_lean_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _c_float_p,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "mr_mask_classifier_b8_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_gather_small_f32": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_copy_segments": (ctypes.c_int, [ctypes.POINTER(CopySegment), ctypes.c_int32, ctypes.c_void_p]),
    "mr_max_over_frames_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                              ctypes.c_void_p]),
    "mr_nonzero_mean_over_frames_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                       ctypes.c_void_p]),
    "mr_pool2x2_framemax_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mr_apply_mask_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]),
    "mr_mask_classifier_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "mr_depth_heads_f32": (ctypes.c_int, [ctypes.POINTER(HeadDesc), ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mr_sparse_metric_sums_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_float,
                                                 ctypes.c_void_p, ctypes.c_void_p]),
    "mr_static_mask_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    "mr_pointcloud_append_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                                ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_resample_ksize_bilinear": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mr_resample_coeffs_bilinear": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_void_p, ctypes.c_void_p]),
    "mr_preprocess_image_u8_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                                  ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_lidar_inverse_depth_u16_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_dso_inverse_depth_u16_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_double, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mr_wino44_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    "mr_wino44_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv3x3_winograd44_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_conv3x3_winograd44_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_wino44s_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    "mr_wino44s_pack_weights_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_void_p]),
    "mr_conv3x3_winograd44s_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_conv3x3_winograd44s_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_conv3x3_winograd44w_lds_bytes": (ctypes.c_int64, [ctypes.POINTER(WinoDesc)]),
    "mr_conv3x3_winograd44w_f32": (ctypes.c_int, [ctypes.POINTER(WinoDesc), ctypes.c_void_p]),
    "mr_abi_version": (ctypes.c_int, []),
    "mr_error_string": (ctypes.c_char_p, [ctypes.c_int]),
}

# exported by the diagnostic library only (python -m monorec_amd.build --timeline; include/monorec_hip.h under MR_DIAGNOSTIC_LIBRARY): typed
# when present, never required
DIAGNOSTIC_ABI = {
    "mr_diagnostic_forms": (ctypes.c_int, []),      # bit 0: F(2,7) instantiations present
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmonorec_hip.so")
_lib = None


def load():
    """Load (once) and type the shared library. Raises RuntimeError if it is missing/incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MR_HIP_LIBRARY") or LIB_PATH      # MR_HIP_LIBRARY: e.g. the timeline-instrumented diagnostic build
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the gfx950 kernels first (python -m monorec_amd.build or "
            "__graft_entry__.build()). monorec_amd has no CPU/PyTorch fallback for its hot path.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in ABI.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{path} does not export {name}; rebuild it") from e
        fn.restype = restype
        fn.argtypes = argtypes
    lib.has_diagnostic_forms = True
    for name, (restype, argtypes) in DIAGNOSTIC_ABI.items():
        fn = getattr(lib, name, None)
        if fn is None:
            lib.has_diagnostic_forms = False
            continue
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mr_abi_version() != MR_ABI_VERSION:
        raise RuntimeError("libmonorec_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(code, what=""):
    """Turn a non-zero return code into a RuntimeError with hipGetErrorString / MR_ERR text."""
    if code != 0:
        msg = load().mr_error_string(int(code))
        raise RuntimeError(f"{what}: {msg.decode() if msg else code} (code {code})")
