"""ctypes binding of libtaseg_hip.so (the C ABI declared in include/taseg_hip.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is
not on a ROCm device every op raises.  PyTorch is used for device memory and streams
only; all arithmetic of the hot path happens inside the library's HIP kernels.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
from .options import options as _options

LIB_PATH = _options.hip_lib or os.path.join(_HERE, "libtaseg_hip.so")   # override: A/B builds of the kernels

TS_OK = 0
_c = ctypes
_vp, _i64, _i32, _sz = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_size_t

# name -> (restype, argtypes); mirrors include/taseg_hip.h one to one
SIGNATURES = {
    "ts_version": (_c.c_char_p, []),
    "ts_last_error": (_c.c_char_p, []),
    "ts_set_option": (_i32, [_i32, _i64]),
    "ts_get_option": (_i64, [_i32]),
    "ts_hash": (_i32, [_vp, _i64, _vp, _vp]),
    "ts_kernel_hash": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "ts_hash_query_workspace_bytes": (_sz, [_i64]),
    "ts_hash_query": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "ts_count": (_i32, [_vp, _i64, _vp, _i64, _vp]),
    "ts_voxelize_forward": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_voxelize_backward": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devoxelize_forward": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devoxelize_backward": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_convolution_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "ts_convolution_forward": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "ts_convolution_backward": (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32,
                                       _vp, _sz, _vp]),
    "ts_downsample_workspace_bytes": (_sz, [_i64]),
    "ts_downsample": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ts_unique_workspace_bytes": (_sz, [_i64]),
    "ts_unique_i64": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_build_kmap_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "ts_build_kmap": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_build_kmap_sym": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_nbr_from_nbmaps": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "ts_trilinear_workspace_bytes": (_sz, [_i64]),
    "ts_trilinear_map": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ts_conv_nbr": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp]),
    "ts_conv_wgrad": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "ts_conv_wgrad_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "ts_conv_wgrad_det": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _sz, _vp]),
    "ts_conv_wgrad_f16_det": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _sz, _vp]),
    "ts_conv_pair_gemm": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "ts_conv_gather_sum": (_i32, [_vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp]),
    "ts_bn_finalize": (_i32, [_vp, _vp, _c.c_double, _i32, _c.c_float, _c.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ts_bn_act_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "ts_bn_act_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_double, _i64, _i32, _vp, _vp, _vp]),
    "ts_bn_train_workspace_bytes": (_sz, [_i32]),
    "ts_bn_act_train_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _i32, _vp,
                                       _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_bn_act_train_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_bn_sync_stats": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "ts_bn_sync_backward_reduce": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_devoxelize_forward_ld": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp]),
    "ts_devoxelize_backward_runs_ld": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devoxelize_backward_csr_ld": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devox_segments": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ts_devoxelize_backward_cells_ld": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "ts_devoxelize_forward_f16_ld": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp]),
    "ts_devoxelize_backward_csr_f16_ld": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devoxelize_backward_cells_f16_ld": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "ts_devox_csr_workspace_bytes": (_sz, [_i64]),
    "ts_devox_csr": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ts_devoxelize_backward_csr": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_devox_order_workspace_bytes": (_sz, [_i64]),
    "ts_devox_order": (_i32, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ts_devoxelize_backward_runs": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ts_image_plan_workspace_bytes": (_sz, [_i64]),
    "ts_image_plan": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_image_gather_forward": (_i32, [_vp, _i32, _i64, _vp, _vp, _i64, _vp, _vp]),
    "ts_image_gather_backward": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "ts_image_gather_rows_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ts_image_gather_rows_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "ts_avgpool3s2_rows_forward": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ts_avgpool3s2_rows_backward": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ts_leaky_bn_train_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _c.c_float, _i32, _vp, _vp, _vp,
                                         _vp, _vp, _sz, _vp]),
    "ts_leaky_bn_train_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _c.c_float, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_conv3x3c32_packed_bytes": (_sz, []),
    "ts_conv3x3c32_pack": (_i32, [_vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "ts_conv3x3c32_rows": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ts_shuffle_cat_rows_forward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ts_shuffle_cat_rows_backward": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ts_conv3x3_rows_packed_bytes": (_sz, [_i32, _i32]),
    "ts_conv3x3_rows_pack": (_i32, [_vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "ts_conv3x3_rows": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "ts_conv1x1c32_packed_bytes": (_sz, []),
    "ts_conv1x1c32_pack": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "ts_conv1x1c32_rows": (_i32, [_vp, _vp, _vp, _i64, _i32, _c.c_float, _vp, _vp]),
    "ts_conv1x1c32_wgrad": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ts_conv3x3_wgrad_workspace_bytes": (_sz, [_i32, _i32]),
    "ts_conv3x3_wgrad": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ts_conv3x3c32_wgrad_workspace_bytes": (_sz, []),
    "ts_conv3x3c32_wgrad": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ts_cast_weights_f16": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ts_conv_pair_gemm_f16": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "ts_conv_pair_gemm_f16_nat": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "ts_conv_gather_sum_f16": (_i32, [_vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp]),
    "ts_conv_wgrad_f16": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "ts_bn_act_train_forward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _i32,
                                           _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_bn_act_train_backward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_sgd_grad_stats": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "ts_sgd_decide": (_i32, [_vp, _vp, _vp, _c.c_float, _c.c_float, _c.c_float, _i32, _i32, _vp]),
    "ts_sgd_apply": (_i32, [_vp, _vp, _vp, _i64, _vp, _c.c_float, _c.c_float, _c.c_float, _i32, _vp]),
    "ts_bn_sync_stats_f16": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "ts_bn_act_forward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "ts_bn_sync_backward_reduce_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_bn_act_backward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_double, _i64, _i32, _vp, _vp, _vp, _sz,
                                      _vp]),
    "ts_rccl_load": (_i32, [_c.c_char_p]),
    "ts_rccl_unique_id": (_i32, [_vp]),
    "ts_rccl_comm_init": (_i32, [_vp, _i32, _i32, _c.POINTER(_vp)]),
    "ts_rccl_comm_destroy": (_i32, [_vp]),
    "ts_rccl_allreduce_f64": (_i32, [_vp, _vp, _i64, _vp]),
    "ts_bn_sync_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _i32, _i32, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_bn_sync_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _sz, _vp]),
    "ts_conv_block_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "ts_conv_block_wgrad_ws_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "ts_stream_join": (_i32, [_vp, _vp]),
    "ts_conv_block_wgrad_side": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp, _i32, _vp, _sz, _i32, _vp]),
    "ts_set_device": (_i32, [_i32]),
    "ts_scene_counts": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "ts_unvoxelise": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ts_cat_cols": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "ts_copy_cols": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ts_conv_block_forward": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _c.c_float, _c.c_float, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _sz, _vp]),
    "ts_conv_block_eval": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32,
                                  _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_conv_block_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp,
                                      _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                      _vp]),
    "ts_prof_enable": (None, [_i32]),
    "ts_prof_reserve": (_i32, [_i64]),
    "ts_prof_collect": (_i64, [_vp, _i64]),
    "ts_prof_empty_bracket_us": (_i32, [_i32, _vp, _vp]),
    "ts_set_conv_impl": (None, [_i32]),
    "ts_softmax_ce_forward": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ts_ce_lovasz_finish": (_i32, [_vp, _i64, _i32, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp]),
    "ts_ce_lovasz_backward": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _c.c_float, _vp, _vp]),
    "ts_lovasz_errors": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "ts_lovasz_workspace_bytes": (_sz, [_i64, _i32]),
    "ts_lovasz_grad": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "ts_conv_split_planes": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "ts_conv_split_planes_batch": (_i32, [_vp, _i32, _vp]),
    "ts_cast_weights_f16_batch": (_i32, [_vp, _i32, _vp]),
    "ts_conv_class_rows": (_i64, [_i64]),
    "ts_conv_class_rows2": (_i64, [_i64, _i32]),
    "ts_conv_class_plan_workspace_bytes": (_sz, [_i64]),
    "ts_conv_class_plan": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_conv_class_plan_pairs": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_conv_nbr_transposed": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "ts_conv_class_supported": (_i32, [_i32, _i32]),
    "ts_conv_class_gemm": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ts_conv_class_gemm_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ts_conv_class_finish_pays": (_i32, [_i64, _i32]),
    "ts_conv_class_conv": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ts_conv_class_conv_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ts_conv_planes_hint": (None, [_vp, _vp, _i32, _i32, _i32]),
    "ts_debug_phase_stamps": (None, [_vp, _i64]),
    "ts_fuse_scan": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ts_fuse_scans": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "ts_fuse_scans_batch": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "ts_fuse_sweeps": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "ts_project_fov": (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _c.c_float, _vp, _vp, _vp]),
    "ts_project_cam": (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _c.c_float, _vp, _vp, _vp]),
    "ts_segment_min3": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "ts_stage_keep_flags": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ts_stage_layout": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ts_stage_split_voxels": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ts_quantize_workspace_bytes": (_sz, [_i64]),
    "ts_sparse_quantize": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ts_voxel_coords": (_i32, [_vp, _i64, _i32, _c.c_float, _vp, _i32, _vp, _vp, _vp, _vp]),
}



class TsClassPlan(_c.Structure):
    """include/taseg_hip.h: a class-sorted plan of one kernel map, by reference"""
    _fields_ = [("src", _vp), ("tile_info", _vp), ("n_tiles", _vp), ("pos", _vp), ("rows", _vp), ("n", _i64), ("m_pad", _i64),
                ("z_rows", _i64), ("K", _i32), ("groups", _i32), ("mirror", _i32), ("map_id", _vp)]


class TsConvBlockOpts(_c.Structure):
    """include/taseg_hip.h: what a ts_conv_block_* call may use beyond the rulebook"""
    _fields_ = [("fwd_plan", _c.POINTER(TsClassPlan)), ("dgrad_plan", _c.POINTER(TsClassPlan)), ("planes", _vp),
                ("w16_current", _i32), ("addend", _vp), ("wgrad_stream", _vp), ("wgrad_ws", _vp), ("wgrad_ws_bytes", _sz),
                ("wgrad_slot", _i32), ("wgrad_deferred", _i32), ("natural", _i32)]


_lib = None
_lock = threading.Lock()


class BackendError(RuntimeError):
    """A libtaseg_hip entry point returned a TS_ERR_* code."""


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -m taseg_amd.csrc.build` "
                    "(or __graft_entry__.build()).  taseg_amd has no CPU fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so lags behind the header
                fn.restype = res
                fn.argtypes = args
            push_options(lib)
            _lib = lib
    return _lib


# include/taseg_hip.h TS_OPT_*: key -> value taken from the options object
_OPTION_KEYS = {0: lambda o: int(o.gather_positions), 1: lambda o: o.wgrad_wgs, 2: lambda o: int(not o.eval_tail_in_pass2),
                3: lambda o: o.class_finish_rows, 4: lambda o: o.class_finish_rows_half,
                5: lambda o: int(o.debug_bn_ablate or 0), 6: lambda o: int(not o.kmap_sym)}


def push_options(lib=None):
    """hand the library its tuning values (called when it is loaded; call again after changing one of those fields)"""
    lib = lib or load()
    for key, get in _OPTION_KEYS.items():
        if lib.ts_set_option(key, int(get(_options))) != TS_OK:
            raise BackendError(f"ts_set_option({key}): " + lib.ts_last_error().decode("utf-8", "replace"))


def check(rc, what=""):
    if rc != TS_OK:
        msg = load().ts_last_error().decode("utf-8", "replace")
        raise BackendError(f"{what or 'taseg_hip'} failed (code {rc}): {msg}")


def require_device(*tensors):
    """Every tensor handed to the backend must live on a ROCm device."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "taseg_amd: tensor is on '%s'; the HIP backend only accepts ROCm device tensors "
                "(there is no CPU fallback in the product path)" % t.device)


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream (raw getter: the Stream-object path costs ~10 us per call and
    every backend call needs it)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# ---- per-device scratch buffer handed to the library as (ws, ws_bytes) -------------------
_ws = {}


def workspace(nbytes, device):
    """Stream-ordered scratch: one growing buffer per (device, stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        cap = max(int(nbytes * 1.5), 1 << 20)
        buf = torch.empty(cap, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf
