"""ctypes binding of libsgv3d_hip.so (the C ABI declared in include/sgv3d_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails this module
raises.  PyTorch is only used by callers for device memory and streams; every compute call here
takes raw device pointers (``tensor.data_ptr()``) and the current HIP stream handle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsgv3d_hip.so")

c_int, c_void_p, c_size_t, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_longlong


class ConvDesc(ctypes.Structure):
    """Mirror of ``sgv3d_conv_desc`` (include/sgv3d_hip.h)."""
    _fields_ = [(n, c_int) for n in (
        "batch", "in_h", "in_w", "cin", "out_h", "out_w", "cout", "kh", "kw", "stride", "pad", "dil",
        "x_ld", "x_coff", "y_ld", "y_coff", "res_ld", "relu", "mode", "deconv_ks", "k_pad", "cout_pad",
        "tile", "x_nchw", "k_order", "split_k")]


CONV_NORMAL, CONV_DECONV, CONV_NCHW_OUT, CONV_GROUP_PLANES = 0, 1, 2, 3
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x128, TILE_64x64 = 0, 1, 2, 3, 4

# name -> (restype, argtypes); must list every symbol include/sgv3d_hip.h declares
_PROTOS = {
    "sgv3d_last_error": (ctypes.c_char_p, []),
    "sgv3d_abi_version": (c_int, []),
    "sgv3d_voxel_pooling_select_kernel": (c_int, [c_int]),
    "sgv3d_voxel_pooling_kernel_for": (c_int, [c_int] * 6),
    "sgv3d_voxel_pooling_forward": (c_int, [c_int] * 6 + [c_void_p] * 5),
    "sgv3d_voxel_pooling_forward_fresh": (c_int, [c_int] * 6 + [c_void_p] * 5),
    "sgv3d_voxel_pooling_forward_atomic": (c_int, [c_int] * 6 + [c_void_p] * 5),
    "sgv3d_voxel_pooling_cache_clear": (c_int, []),
    "sgv3d_voxel_pooling_cache_stats": (c_int, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "sgv3d_voxel_plan_bytes": (c_size_t, [c_int] * 4),
    "sgv3d_voxel_plan_build": (c_int, [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "sgv3d_voxel_plan_init": (c_int, [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    "sgv3d_voxel_plan_build_cached": (c_int, [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "sgv3d_voxel_plan_stats_offset": (c_size_t, [c_int] * 4),
    "sgv3d_voxel_pooling_workspace_bytes": (c_size_t, [c_int] * 3),
    "sgv3d_voxel_pooling_forward_planned": (c_int, [c_int] * 5 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "sgv3d_voxel_pooling_forward_planned_bf16": (c_int, [c_int] * 5 + [c_void_p] * 3 + [c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_lift_splat_planned_bf16out": (c_int, [c_int] * 6 + [c_void_p] * 3 + [c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_lift_splat_planned": (c_int, [c_int] * 6 + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "sgv3d_voxel_pooling_backward": (c_int, [c_int] * 3 + [c_void_p, c_void_p, c_ll, c_ll, c_ll, c_ll, c_void_p, c_void_p]),
    "sgv3d_calib_prep": (c_int, [c_int] + [c_void_p] * 6),
    "sgv3d_calib_prep_gated": (c_int, [c_int] + [c_void_p] * 7),
    "sgv3d_calib_changed": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_void_p, c_int, c_void_p, c_void_p]),
    "sgv3d_geometry_voxel_index_gated": (c_int, [c_int] * 5 + [c_void_p] * 4 + [ctypes.POINTER(ctypes.c_float)] * 2 + [c_void_p] * 4),
    "sgv3d_dense_gated": (c_int, [c_int] * 3 + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "sgv3d_geometry_voxel_index": (c_int, [c_int] * 5 + [c_void_p] * 4 + [ctypes.POINTER(ctypes.c_float)] * 2 + [c_void_p] * 3),
    "sgv3d_lift": (c_int, [c_int] * 4 + [c_void_p] * 4),
    "sgv3d_lift_bf16": (c_int, [c_int] * 4 + [c_void_p] * 4),
    "sgv3d_conv_pack_geometry": (None, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "sgv3d_conv_pack_weight": (c_int, [c_void_p] + [c_int] * 7 + [c_void_p, c_int, c_int, c_void_p]),
    "sgv3d_conv2d_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "sgv3d_conv2d_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "sgv3d_conv2d_forward_bf16": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "sgv3d_conv2d_forward_bf16io": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_size_t, c_void_p, c_int]),
    "sgv3d_conv_weight_to_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv2d_forward_f32x3": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "sgv3d_centerhead_branches_workspace_bytes": (c_size_t, [c_int] * 4),
    "sgv3d_centerhead_branches_forward": (c_int, [c_int] * 6 + [c_void_p, c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 5
                                          + [c_size_t, c_void_p]),
    "sgv3d_centerhead_bf16_weight_bytes": (c_size_t, [c_int]),
    "sgv3d_centerhead_bf16_pack_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_centerhead_bf16_weight2_bytes": (c_size_t, [c_int]),
    "sgv3d_centerhead_bf16_pack_weight2": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sgv3d_centerhead_f4_weight_floats": (c_size_t, [c_int]),
    "sgv3d_centerhead_f4_pack_weight": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "sgv3d_centerhead_branches_forward_f4": (c_int, [c_int] * 6 + [c_void_p, c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 5
                                             + [c_size_t, c_void_p]),
    "sgv3d_conv3x3_f4res_weight_floats": (c_size_t, [c_int, c_int]),
    "sgv3d_conv3x3_f4res_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv3x3_f4res_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 7),
    "sgv3d_centerhead_branches_forward_bf16": (c_int, [c_int] * 6 + [c_void_p, c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 5),
    "sgv3d_centerhead_branches_forward_bf16x": (c_int, [c_int] * 6 + [c_void_p, c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 5),
    "sgv3d_centerhead_bf16_select_plain": (None, [c_int]),
    "sgv3d_centerhead_bf16_debug_stamps": (None, [c_void_p]),
    "sgv3d_scale_channels_bf16": (c_int, [c_int] * 3 + [c_void_p] * 4),
    "sgv3d_global_avgpool_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "sgv3d_broadcast_channels_bf16": (c_int, [c_int] * 5 + [c_void_p] * 3),
    "sgv3d_upsample_bilinear2x_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_add_mul_sigmoid_bf16": (c_int, [c_ll] + [c_void_p] * 5),
    "sgv3d_deform_im2col3x3_bf16": (c_int, [c_int] * 5 + [c_void_p] * 2 + [c_int, c_void_p, c_void_p]),
    "sgv3d_conv3x3_patch_bf16_weight_bytes": (c_size_t, [c_int, c_int]),
    "sgv3d_conv3x3_patch_bf16_debug_stamps": (None, [c_void_p]),
    "sgv3d_conv3x3_patch_bf16_pack_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv3x3_patch_bf16_forward": (c_int, [c_int] * 11 + [c_void_p] * 6 + [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv_dw_bf16_weight_bytes": (c_size_t, [c_int] * 4),
    "sgv3d_conv_dw_bf16_pack_weight": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p]),
    "sgv3d_conv_dw_bf16_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 7),
    "sgv3d_conv_dw_bf16_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "sgv3d_conv_dw_bf16_forward_splitk": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 7 + [c_size_t, c_void_p]),
    "sgv3d_conv_dw_bf16_pair_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sgv3d_conv_winograd_weight_floats": (c_size_t, [c_int, c_int]),
    "sgv3d_conv_winograd_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv2d_winograd_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "sgv3d_conv_winograd4_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv_winograd4_pack_weight_x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv_pack_weight_x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_conv2d_x3_forward": (c_int, [c_void_p] * 8 + [c_size_t, c_void_p]),
    "sgv3d_conv2d_winograd4_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "sgv3d_conv2d_winograd4_forward": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 7 + [c_size_t, c_void_p]),
    "sgv3d_maxpool3x3s2": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_maxpool3x3s2_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_nchw_to_nhwc": (c_int, [c_int] * 5 + [c_void_p] * 3),
    "sgv3d_nhwc_to_nchw": (c_int, [c_int] * 6 + [c_void_p] * 3),
    "sgv3d_global_avgpool_workspace_bytes": (c_size_t, [c_int] * 2),
    "sgv3d_global_avgpool": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "sgv3d_dense": (c_int, [c_int] * 3 + [c_void_p] * 4 + [c_int, c_void_p, c_void_p]),
    "sgv3d_broadcast_channels": (c_int, [c_int] * 5 + [c_void_p] * 3),
    "sgv3d_scale_channels": (c_int, [c_int] * 3 + [c_void_p] * 4),
    "sgv3d_copy_channels": (c_int, [c_int] * 5 + [c_void_p] * 3),
    "sgv3d_upsample_bilinear2x": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_add_mul_sigmoid": (c_int, [c_ll] + [c_void_p] * 5),
    "sgv3d_bsm_compose": (c_int, [c_int] * 6 + [c_void_p, c_int, ctypes.c_float, c_void_p, c_void_p]),
    "sgv3d_deform_im2col3x3": (c_int, [c_int] * 5 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sgv3d_deform_conv3x3_forward": (c_int, [c_int] * 6 + [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                                          c_void_p]),
    "sgv3d_head_final_conv": (c_int, [c_int] * 6 + [c_void_p] * 6),
    "sgv3d_centerpoint_decode_workspace_bytes": (c_size_t, [c_int] * 3),
    "sgv3d_centerpoint_decode": (c_int, [c_int] * 5 + [c_void_p] * 6 + [c_ll] + [ctypes.c_float] * 6 +
                                 [ctypes.POINTER(ctypes.c_float), c_int, ctypes.c_float, c_int, c_void_p, c_size_t] +
                                 [c_void_p] * 6),
    "sgv3d_centerpoint_decode_tasks_workspace_bytes": (c_size_t, [c_int] * 4),
    "sgv3d_centerpoint_decode_tasks": (c_int, [c_int, c_int, ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_int] +
                                       [ctypes.POINTER(c_void_p)] * 6 + [c_ll] + [ctypes.c_float] * 6 +
                                       [ctypes.POINTER(ctypes.c_float), c_int, ctypes.POINTER(ctypes.c_float), c_int, c_void_p,
                                        c_size_t] + [c_void_p] * 6),
    "sgv3d_centerpoint_merge_tasks": (c_int, [c_int] * 3 + [c_void_p] * 4 + [ctypes.POINTER(ctypes.c_int32)] + [c_void_p] * 5),
    "sgv3d_centerhead_targets": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(ctypes.c_int32)] +
                                 [c_int] * 3 + [ctypes.c_float] * 5 + [ctypes.c_double, c_int, c_int] + [c_void_p] * 5),
    "sgv3d_conv2d_backward_weight_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "sgv3d_conv2d_backward_weight": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 3 + [c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_bf16_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "sgv3d_conv2d_backward_weight_bf16": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 3 + [c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_bf16_batched_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "sgv3d_conv2d_backward_weight_bf16_batched": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "sgv3d_conv2d_backward_weight_bf16_alltaps": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 3 + [c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_bf16_alltaps_batched": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_thin_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "sgv3d_conv2d_backward_weight_batched_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "sgv3d_conv2d_backward_weight_batched": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sgv3d_conv2d_backward_weight_thin": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "sgv3d_conv3x3_thin_forward_batched": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(ctypes.c_int32)] + [ctypes.POINTER(c_void_p)] * 4 + [c_void_p]),
    "sgv3d_conv3x3_thin_backward_batched_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(ctypes.c_int32)]),
    "sgv3d_conv3x3_thin_backward_batched": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(ctypes.c_int32)] + [ctypes.POINTER(c_void_p)] * 6 + [c_void_p, c_size_t, c_void_p]),
    "sgv3d_weight_rot180_transpose": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_void_p]),
    "sgv3d_gather_pack_job_bytes": (c_int, []),
    "sgv3d_gather_pack_elements_per_block": (c_int, []),
    "sgv3d_gather_pack": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sgv3d_zero_insert": (c_int, [c_int] * 7 + [c_void_p] * 3),
    "sgv3d_interleave_phases2": (c_int, [c_int] * 4 + [ctypes.POINTER(c_void_p)] + [ctypes.POINTER(ctypes.c_int32)] * 4 +
                                 [c_void_p, c_void_p]),
    "sgv3d_batchnorm_workspace_bytes": (c_size_t, [c_int]),
    "sgv3d_batchnorm_train_forward": (c_int, [c_ll, c_int] + [c_void_p] * 6 + [ctypes.c_float, ctypes.c_float, c_int] +
                                      [c_void_p] * 4 + [c_size_t, c_void_p]),
    "sgv3d_batchnorm_relu_train_backward_from_x": (c_int, [c_ll, c_int] + [c_void_p] * 9 + [c_void_p, c_size_t, c_void_p]),
    "sgv3d_batchnorm_train_backward": (c_int, [c_ll, c_int] + [c_void_p] * 6 + [c_int] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "sgv3d_adamw_step": (c_int, [c_ll] + [c_void_p] * 4 + [c_int] + [ctypes.c_float] * 6 + [c_void_p] * 2),
    "sgv3d_adamw_step_dev": (c_int, [c_ll] + [c_void_p] * 5 + [ctypes.c_float] * 5 + [c_void_p] * 2),
    "sgv3d_adamw_set_hyper": (c_int, [c_void_p, c_int] + [ctypes.c_float] * 3 + [c_void_p]),
    "sgv3d_grad_sumsq_partials": (c_int, []),
    "sgv3d_grad_sumsq": (c_int, [c_ll, c_void_p, c_void_p, c_void_p]),
    "sgv3d_clip_coef": (c_int, [c_void_p, c_int, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    "sgv3d_centerhead_loss_workspace_bytes": (c_size_t, [c_int]),
    "sgv3d_centerhead_loss_stats": (c_int, [c_int] * 5 + [c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sgv3d_centerhead_loss": (c_int, [c_int] * 5 + [c_void_p] * 6 + [c_ll, c_void_p, c_ll] + [c_void_p] * 4 +
                              [ctypes.POINTER(ctypes.c_float), ctypes.c_float, ctypes.c_float] + [c_void_p] * 6 +
                              [c_ll, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sgv3d_semantic_labels_downsample": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_focal_loss_workspace_bytes": (c_size_t, []),
    "sgv3d_focal_loss_with_logits": (c_int, [c_int] * 3 + [c_void_p, c_ll, c_ll, c_ll, c_void_p, c_int, ctypes.c_float,
                                             ctypes.c_float, c_ll, c_int, c_int, ctypes.c_float, c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    "sgv3d_upsample_bilinear2x_backward": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "sgv3d_add_mul_sigmoid_backward": (c_int, [c_ll] + [c_void_p] * 6),
    "sgv3d_maxpool3x3s2_train_forward": (c_int, [c_int] * 4 + [c_void_p] * 4),
    "sgv3d_maxpool3x3s2_backward": (c_int, [c_int] * 4 + [c_void_p] * 4),
    "sgv3d_dense_backward_weight": (c_int, [c_int] * 3 + [c_void_p] * 4),
    "sgv3d_deform_im2col3x3_backward": (c_int, [c_int] * 5 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sgv3d_rotate_iou_pairs": (c_int, [c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p]),
    "sgv3d_kitti_eval_curves": (c_int, [c_int] + [c_void_p] * 9 + [c_int, ctypes.c_double, c_int, c_ll, c_int] + [c_void_p] * 4),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


class SGV3DError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SGV3DError(
                f"{LIB_PATH} is missing: build it with `make -C sgv3d_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        # torch must be imported first: it bundles its own libamdhip64 (SONAME libamdhip64.so.7) and this
        # library has to bind to that same runtime instance, otherwise the process ends up with two HIP
        # runtimes and pointers / streams from one are unknown to the other ("no ROCm-capable device").
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().sgv3d_last_error()
        raise SGV3DError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def stream_handle(device=None):
    """Current HIP stream of torch as an integer handle (0 = null stream)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """Device pointer of a tensor or None."""
    return None if t is None else t.data_ptr()
