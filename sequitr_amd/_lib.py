"""ctypes loader for libsequitr_hip.so (the C-ABI of include/sequitr_hip.h).

The product path has NO fallback: if the library is missing or a symbol is
absent this module raises, loudly.  Build with ``make -C sequitr_amd/csrc`` or
``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libsequitr_hip.so")

c_void_p, c_int, c_float, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64


class WgradItem(ctypes.Structure):
    """sq_wgrad_item of include/sequitr_hip.h"""
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("db", c_void_p),
                ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
                ("Cout", ctypes.c_int32), ("K", ctypes.c_int32), ("convT_cout", ctypes.c_int32), ("dw_scale", c_float),
                ("accumulate", ctypes.c_int32), ("mosaic_R", ctypes.c_int32), ("mosaic_Cc", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]

# name -> (restype, argtypes); must list every symbol include/sequitr_hip.h declares
SIGNATURES = {
    "sq_version": (c_int, []),
    "sq_last_error": (ctypes.c_char_p, []),
    "sq_conv2d_nhwc_fwd_f32": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_int, c_void_p]),
    "sq_maxpool2x2_fwd_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "sq_avgpool2x2_fwd_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "sq_convT2x2s2_nhwc_fwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_bridge_fwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_conv1x1_argmax_fwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "sq_argmax_u8": (c_int, [c_void_p] * 2 + [c_int64, c_int, c_void_p]),
    "sq_pixelnorm_fwd_f32": (c_int, [c_void_p] * 2 + [c_int64, c_int, c_float, c_void_p]),
    "sq_upsample_nn2x_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "sq_wsoftmax_ce_partials": (c_int64, [c_int64]),
    "sq_wsoftmax_ce_fwd_bwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float] + [c_void_p] * 4),
    "sq_conv_weight_transform_f32": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "sq_conv2d_nhwc_wgrad_workspace_f32": (c_int64, [c_int] * 6),
    "sq_conv2d_nhwc_wgrad_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_act_bwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_maxpool2x2_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "sq_broadcast2x2_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_float, c_void_p]),
    "sq_sumpool2x2_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_float, c_void_p]),
    "sq_bridge_bwd_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_void_p]),
    "sq_space_to_depth2_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "sq_conv1x1_small_bwd_workspace_f32": (c_int64, [c_int64, c_int, c_int]),
    "sq_conv1x1_small_bwd_f32": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "sq_dropout_fwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_float, ctypes.c_uint32, c_int, c_void_p, c_void_p]),
    "sq_dropout_bwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_float, c_void_p]),
    "sq_adam_step_f32": (c_int, [c_void_p] * 4 + [c_int64] + [c_float] * 4 + [c_int, c_float, c_void_p]),
    "sq_axpy_f32": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "sq_mask_centroids_workspace": (c_int64, [c_int, c_int, c_int]),
    "sq_mask_centroids_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p]),
    "sq_weightmap_workspace": (c_int64, [c_int, c_int, c_int]),
    "sq_edt_sq_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sq_weightmap_edt_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     ctypes.c_double, ctypes.c_double, c_void_p]),
    "sq_weightmap2_workspace": (c_int64, [c_int, c_int, c_int]),
    "sq_wm2_boundary_points_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sq_delaunay2d_batch_i32": (c_int64, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64]),
    "sq_weightmap2_delaunay_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                           c_int, ctypes.c_double, ctypes.c_double, c_void_p]),
    "sq_frame_stats_workspace": (c_int64, [c_int, c_int, c_int]),
    "sq_frame_stats": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sq_frames_to_tiles": (c_int, [c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_stitch_masks_u8": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "sq_dense_workspace_f32": (c_int64, [c_int, c_int, c_int]),
    "sq_dense_fwd_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "sq_convT_conv3x3_fwd_f32": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "sq_volume_centroids_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p]),
    "sq_mosaic_pack_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_mosaic_unpack_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_zero_insert2x_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sq_gather_odd2x_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sq_bn_workspace_f32": (c_int64, [c_int64, c_int]),
    "sq_bn_stats_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_void_p]),
    "sq_bn_stats_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_void_p]),
    "sq_bn_fold_f32": (c_int, [c_void_p] * 4 + [c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "sq_bn_update_moving_f32": (c_int, [c_void_p] * 4 + [c_float, c_int64, c_int, c_void_p]),
    "sq_bn_apply_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "sq_bn_apply_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "sq_bn_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_float] + [c_void_p] * 4
                      + [c_int64, c_int, c_void_p]),
    "sq_bn_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_float] + [c_void_p] * 4
                      + [c_int64, c_int, c_void_p]),
    "sq_adam_step_dev_f32": (c_int, [c_void_p] * 4 + [c_int64] + [c_float] * 4 + [c_void_p, c_float, c_void_p]),
    "sq_adam_advance_dev": (c_int, [c_void_p, c_float, c_float, c_float, c_void_p]),
    "sq_adam_advance_warmup_dev": (c_int, [c_void_p, c_float, c_float, c_float, c_int, c_void_p]),
    "sq_adam_apply_dev_f32": (c_int, [c_void_p] * 4 + [c_int64, c_float, c_float, c_float, c_void_p, c_float, c_void_p]),
    "sq_adam_multi_chunk": (c_int, []),
    "sq_adam_apply_multi_dev_f32": (c_int, [c_void_p, c_int, c_int64, c_float, c_float, c_float, c_void_p, c_float, c_void_p]),
    "sq_pixelnorm_bwd_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_void_p]),
    "sq_pixelnorm_bwd2_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_float, c_void_p]),
    "sq_pixelnorm_bwd_act_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "sq_broadcast2x2_act_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "sq_resize_nearest_f32": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "sq_lerp_f32": (c_int, [c_void_p] * 3 + [c_int64, c_int64, c_float, c_void_p, c_void_p]),
    "sq_scale_f32": (c_int, [c_void_p] * 2 + [c_int64, c_int64, c_float, c_void_p, c_int, c_void_p]),
    "sq_act_fwd_f32": (c_int, [c_void_p] * 2 + [c_int64, c_int, c_void_p]),
    "sq_dot_per_sample_workspace_f32": (c_int64, [c_int]),
    "sq_dot_per_sample_f32": (c_int, [c_void_p] * 4 + [c_int, c_int64, c_void_p]),
    "sq_mbstd_fwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_int64, c_void_p]),
    "sq_mbstd_map_workspace": (c_int64, [c_int]),
    "sq_mbstd_map_fwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_mbstd_map_bwd_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_mbstd_map_bwd2_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_mbstd_map_fwd_bf16": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_mbstd_map_bwd_bf16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_mbstd_map_bwd2_bf16": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int64, c_int, c_void_p]),
    "sq_wgan_losses_fwd_f32": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "sq_wgan_losses_bwd_f32": (c_int, [c_void_p] * 8 + [c_int, c_void_p]),
    "sq_dense_wgrad_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_float, c_int, c_void_p]),
    "sq_wgrad1x1_small_workspace_f32": (c_int64, [c_int64, c_int, c_int]),
    "sq_wgrad1x1_small_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "sq_conv2d_concat_nhwc_fwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "sq_conv3x3_pool_fwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_conv3x3_head_fwd_f32": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "sq_conv3x3_first_block_fwd_f32": (c_int, [c_void_p] * 7 + [c_int] * 3 + [c_void_p]),
    "sq_conv_packed_weights_elems_bf16": (c_int64, [c_int] * 3),
    "sq_conv_pack_weights_bf16": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_float, c_int, c_void_p]),
    "sq_conv2d_nhwc_fwd_bf16": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "sq_conv2d_nhwc_fwd_mixed_f32": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "sq_conv2d_nhwc_wgrad_workspace_mixed_f32": (c_int64, [c_int] * 6),
    "sq_conv2d_nhwc_wgrad_mixed_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_wgrad_scaled_mixed_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_wgrad_scaled_f32": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "sq_conv3x3_first_fwd_bf16": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_wgrad_workspace_bf16": (c_int64, [c_int] * 6),
    "sq_conv2d_nhwc_wgrad_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_convT2x2s2_wgrad_bf16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "sq_cast_f32_to_bf16": (c_int, [c_void_p] * 2 + [c_int64, c_void_p]),
    "sq_cast_bf16_to_f32": (c_int, [c_void_p] * 2 + [c_int64, c_void_p]),
    "sq_head_concat_fwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_head_concat_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_maxpool2x2_fwd_bf16": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "sq_maxpool2x2_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "sq_conv_pack_weights_multi_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sq_conv_pack_weights_multi_scaled_bf16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "sq_conv2d_nhwc_fwd_dropout_bf16": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_float, ctypes.c_uint32, c_void_p, c_void_p]),
    "sq_conv3x3_first_fwd_mask_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_fwd_mask_bf16": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "sq_conv2d_nhwc_dgrad_maskgate_bf16": (c_int, [c_void_p] * 3 + [c_float, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_wgrad_mixed_mosaic_f32": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_mixed_mosaic_f32": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "sq_conv2d_nhwc_dgrad_actgate_mixed_f32": (c_int, [c_void_p] * 3 + [c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_fwd_dropout_pool_bf16": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_float, ctypes.c_uint32, c_void_p, c_void_p]),
    "sq_conv3x3_first_block_dropout_pool_bf16": (c_int, [c_void_p] * 9 + [c_int] * 3 + [c_float, ctypes.c_uint32, c_void_p, c_void_p]),
    "sq_relu_scale_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_float, c_void_p]),
    "sq_bridge_bwd_s2d_bf16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "sq_maxpool2x2_bwd_add_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "sq_maxpool2x2_bwd_add_gate_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_dgrad_gate_bf16": (c_int, [c_void_p] * 3 + [c_float, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_dgrad_junction_bf16": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "sq_conv1x1_head_bwd_gate_bf16": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_float, c_void_p]),
    "sq_conv1x1_head_wce_fwd_bf16": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "sq_conv1x1_head_wce_bwd_bf16": (c_int, [c_void_p] * 10 + [c_int64, c_int, c_int, c_float, c_void_p]),
    "sq_conv1x1_head_wce_bwd_loss_bf16": (c_int, [c_void_p] * 12 + [c_int64, c_int, c_int, c_float, c_void_p]),
    "sq_act_dropout_bwd_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_float, c_int, c_void_p]),
    "sq_conv2d_nhwc_dgrad_relu_bf16": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "sq_pixelnorm_fwd_bf16": (c_int, [c_void_p] * 2 + [c_int64, c_int, c_float, c_void_p]),
    "sq_pixelnorm_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "sq_pixelnorm_bwd2_bf16": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_float, c_void_p]),
    "sq_sumpool2x2_bf16": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_float, c_void_p]),
    "sq_broadcast2x2_bf16": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_float, c_void_p]),
    "sq_broadcast2x2_act_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "sq_act_fwd_bf16": (c_int, [c_void_p] * 2 + [c_int64, c_int, c_void_p]),
    "sq_conv1x1_smallin_fwd_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "sq_conv1x1_smallout_fwd_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "sq_wgrad1x1_small_workspace_bf16": (c_int64, [c_int64, c_int, c_int]),
    "sq_wgrad1x1_small_bf16": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_float, c_void_p]),
    "sq_conv2d_nhwc_fwd_avgpool_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_fwd_pixelnorm_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_dgrad_actgate_bf16": (c_int, [c_void_p] * 3 + [c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "sq_conv2d_nhwc_mosaic_bf16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p, c_int64, c_void_p]),
    "sq_conv2d_nhwc_wgrad_scaled_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_wgrad_mosaic_bf16": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_float, c_void_p]),
    "sq_conv2d_nhwc_wgrad_group_workspace_bf16": (c_int64, [c_void_p, c_int]),
    "sq_conv2d_nhwc_wgrad_group_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "sq_act_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_bridge_fwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "sq_bridge_bwd_bf16": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_void_p]),
    "sq_dropout_fwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_float, ctypes.c_uint32, c_int, c_void_p, c_void_p]),
    "sq_dropout_bwd_bf16": (c_int, [c_void_p] * 3 + [c_int64, c_float, c_void_p]),
    "sq_convT2x2s2_nhwc_fwd_bf16": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "sq_convT2x2s2_bridge_both_fwd_bf16": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "sq_conv1x1_head_fwd_bf16": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "sq_conv1x1_head_bwd_workspace_bf16": (c_int64, [c_int64, c_int, c_int]),
    "sq_conv1x1_head_bwd_bf16": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "sq_conv3x3_first_wgrad_workspace_bf16": (c_int64, [c_int] * 5),
    "sq_conv3x3_first_wgrad_bf16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
}

_lib = None


class SequitrHipError(RuntimeError):
    pass


def load():
    """Load the shared library once and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SequitrHipError(
            "libsequitr_hip.so not found at %s -- the HIP back end is mandatory "
            "(no CPU fallback); build it with `make -C sequitr_amd/csrc`" % LIB_PATH)
    # torch ships its own libamdhip64 (soname libamdhip64.so.7).  It must be in the process
    # BEFORE this library is loaded, so that our DT_NEEDED libamdhip64.so.7 binds to the same
    # HIP runtime torch allocates from; loaded the other way round the process ends up with
    # two runtimes and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SequitrHipError("libsequitr_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        msg = load().sq_last_error()
        raise SequitrHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
