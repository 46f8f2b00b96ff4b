"""ctypes binding of libzeroshape_hip.so (C ABI in include/zeroshape_hip.h).

There is NO fallback: if the library is missing or a symbol is absent this module
raises, and every product entry point that needs the GPU path raises with it.
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZS_LIB_PATH") or os.path.join(_HERE, "libzeroshape_hip.so")  # override: experiments only

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_size_t = ctypes.c_size_t
_c_float = ctypes.c_float

class ConvFuse(ctypes.Structure):
    """include/zeroshape_hip.h: zs_conv_fuse."""
    _fields_ = [("in_mode", ctypes.c_int), ("in_tiles", ctypes.c_int), ("in_groups", ctypes.c_int), ("in_gshift", ctypes.c_int),
                ("in_stats", ctypes.c_void_p), ("in_gamma", ctypes.c_void_p), ("in_beta", ctypes.c_void_p),
                ("in_eps", ctypes.c_float), ("out_mode", ctypes.c_int), ("out_groups", ctypes.c_int),
                ("out_stats", ctypes.c_void_p)]


# name -> (restype, argtypes); mirrors include/zeroshape_hip.h one to one
SIGNATURES = {
    "zs_abi_version": (_c_int, []),
    "zs_last_error": (ctypes.c_char_p, []),
    "zs_chamfer_forward": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                    _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_chamfer_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "zs_chamfer_forward_ws": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t,
                                       _c_void_p]),
    "zs_chamfer_backward": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                     _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                     _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_program_bytes": (_c_size_t, []),
    "zs_sdf_prologue_scratch_bytes": (_c_size_t, []),
    "zs_sdf_workspace_bytes": (_c_size_t, []),
    "zs_sdf_attn_scratch_bytes": (_c_size_t, [_c_int, _c_int]),
    "zs_sdf_prologue": (_c_int, [_c_void_p, _c_size_t, _c_void_p, _c_void_p, _c_int,
                                 _c_void_p, _c_void_p]),
    "zs_sdf_prologue_ex": (_c_int, [_c_void_p, _c_size_t, _c_void_p, _c_void_p, _c_int,
                                    _c_void_p, _c_int, _c_void_p]),
    "zs_sdf_verdict_stats": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_float, _c_float, _c_void_p, _c_void_p,
                                      _c_void_p]),
    "zs_sdf_query_points": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int,
                                     _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_query_grid": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int, _c_int,
                                   _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_split_programs": (_c_int, [_c_void_p, _c_size_t, _c_void_p, _c_size_t, _c_int, _c_void_p]),
    "zs_sdf_query_points_split": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int,
                                           _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_query_grid_split": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int, _c_int,
                                         _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_query_grid_range": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int, ctypes.c_longlong,
                                         ctypes.c_longlong, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_sdf_query_grid_range_split": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int,
                                               ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_void_p,
                                               _c_void_p, _c_void_p, _c_void_p]),
    "zs_erode_mask": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_bf_grid_bytes": (_c_size_t, []),
    "zs_bf_scratch_bytes": (_c_size_t, []),
    "zs_bf_lower_bounds": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                                    _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_pose_max_batch": (_c_int, []),
    "zs_pose_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "zs_pose_best_bytes": (_c_size_t, []),
    "zs_pose_best_init": (_c_int, [_c_void_p, _c_void_p]),
    "zs_pose_search_batch": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_pose_grid_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "zs_pose_gt_grid": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "zs_pose_search_batch_grid": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                           _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_morton_scratch_bytes": (_c_size_t, [_c_int]),
    "zs_morton_sort": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "zs_str_scratch_bytes": (_c_size_t, [_c_int]),
    "zs_str_sort": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_pose_pack_bytes": (_c_size_t, [_c_int]),
    "zs_pose_pack": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "zs_pose_sorted_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "zs_pose_search_batch_sorted": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p,
                                             _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                             _c_int, _c_void_p]),
    "zs_pose_apply": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_standardize_pc": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_icp_scratch_bytes": (_c_size_t, [_c_int]),
    "zs_icp_step": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_normalize_pc": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_fscore": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "zs_mc_scratch_bytes": (_c_size_t, [_c_int]),
    "zs_mesh_sample_scratch_doubles": (_c_size_t, [_c_int]),
    "zs_mc_count": (_c_int, [_c_void_p, _c_int, ctypes.c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_mc_emit": (_c_int, [_c_void_p, _c_int, ctypes.c_float, _c_void_p, _c_int, _c_void_p, _c_void_p,
                            ctypes.c_float, ctypes.c_float, _c_void_p, _c_int, _c_void_p]),
    "zs_mesh_sample": (_c_int, [_c_void_p, _c_int, _c_int, ctypes.c_uint64, _c_void_p, _c_void_p, _c_void_p]),
    "zs_intr_param2mtx": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_unproj_depth": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_valid_norm_fac": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_masked_resample": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                    ctypes.c_float, _c_void_p, _c_void_p, _c_void_p]),
    "zs_seen_surface": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_seen_surface_workspace_bytes": (_c_size_t, [_c_int]),
    "zs_seen_surface_ws": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                    _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "zs_depth_metrics": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, ctypes.c_float,
                                  ctypes.POINTER(ctypes.c_float), _c_int, _c_void_p, _c_void_p, _c_void_p,
                                  _c_void_p]),
    "zs_conv2d_packed_floats": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "zs_conv2d_presplit_weight": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_conv2d_nhwc": (_c_int, [_c_void_p] * 7 + [_c_int] * 13 + [ctypes.c_float, ctypes.c_float, _c_int,
                                                                  _c_void_p]),
    "zs_conv2d_splitk_workspace_bytes": (_c_size_t, []),
    "zs_conv2d_presplit_weight_multi": (_c_int, [_c_void_p] * 4 + [_c_int, ctypes.c_ulonglong, _c_void_p]),
    "zs_conv3x3_tail_nhwc": (_c_int, [_c_void_p] * 5 + [_c_int] * 7 + [_c_void_p, _c_void_p, _c_int, _c_void_p]),
    "zs_conv2d_nhwc_ws": (_c_int, [_c_void_p] * 7 + [_c_int] * 13 + [ctypes.c_float, ctypes.c_float, _c_int,
                                                                     _c_void_p, _c_void_p]),
    "zs_conv2d_nhwc_fused": (_c_int, [_c_void_p] * 7 + [_c_int] * 13 + [ctypes.c_float, ctypes.c_float, _c_int,
                                                                        _c_void_p, _c_void_p, _c_void_p]),
    "zs_conv2d_fused_cols": (_c_int, [_c_int, _c_int]),
    "zs_conv2d_k16_ok": (_c_int, [_c_int] * 7),
    "zs_gn_relu_max_pool_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int] * 10 + [ctypes.c_float, _c_void_p]),
    "zs_group_norm_apply_stats": (_c_int, [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 3 +
                                  [_c_int, _c_int, _c_int, ctypes.c_float, _c_int, _c_void_p]),
    "zs_group_norm_nhwc": (_c_int, [_c_void_p] * 5 + [_c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_int,
                                                      _c_void_p]),
    "zs_group_norm_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "zs_group_norm_nhwc_ws": (_c_int, [_c_void_p] * 5 + [_c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_int,
                                       _c_void_p, _c_void_p]),
    "zs_layer_norm": (_c_int, [_c_void_p] * 4 + [_c_int, _c_int, ctypes.c_float, _c_void_p]),
    "zs_attention": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_attention_split": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_max_pool_nhwc": (_c_int, [_c_void_p, _c_void_p] + [_c_int] * 10 + [_c_void_p]),
    "zs_global_mean_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_upsample2x_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_nchw_to_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_window_tokens": (_c_int, [_c_void_p] * 6 + [_c_int] * 5 + [_c_void_p]),
    "zs_nhwc_to_nchw": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_assemble_tokens": (_c_int, [_c_void_p] * 4 + [_c_int, _c_int, _c_int, _c_void_p]),
    "zs_readout_concat": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    # ---- training ----
    "zs_pack_conv_weight": (_c_int, [_c_void_p, _c_void_p] + [_c_int] * 7 + [_c_void_p]),
    "zs_pack_chunk_elems": (_c_int, []),
    "zs_pack_entry_chunks": (_c_int, [_c_int] * 6),
    "zs_pack_conv_weight_multi": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "zs_pack_entry_inline_split": (_c_int, [_c_int] * 4),
    "zs_pack_conv_weight_multi_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p]),
    "zs_conv2d_wgrad_workspace_bytes": (_c_size_t, [_c_int] * 7),
    "zs_conv2d_wgrad": (_c_int, [_c_void_p] * 5 + [_c_int] * 13 + [_c_float, _c_float] + [_c_int] * 4 + [_c_void_p]),
    "zs_conv2d_dgrad_small_cin": (_c_int, [_c_void_p] * 3 + [_c_int] * 15 + [_c_float, _c_void_p]),
    "zs_standardize_weight": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_void_p]),
    "zs_standardize_weight_multi": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    "zs_standardize_weight_bwd": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_void_p]),
    "zs_act_forward": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_int, _c_float, _c_void_p]),
    "zs_act_backward": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_float, _c_void_p]),
    "zs_posenc3d": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_int, _c_void_p]),
    "zs_add_scaled_rows": (_c_int, [_c_void_p] * 4 + [_c_int, _c_size_t, _c_void_p]),
    "zs_column_sum_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "zs_column_sum": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "zs_layer_norm_bwd_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "zs_layer_norm_bwd": (_c_int, [_c_void_p] * 6 + [_c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "zs_layer_norm_bwd_add": (_c_int, [_c_void_p] * 7 + [_c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "zs_attention_bwd_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "zs_attention_bwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p]),
    "zs_point_attention": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p]),
    "zs_point_attention_probs": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int,
                                          _c_void_p]),
    "zs_point_attention_bwd_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "zs_point_attention_bwd": (_c_int, [_c_void_p] * 5 + [_c_int, _c_void_p] + [_c_int] * 5 + [_c_void_p]),
    "zs_bce_logits_workspace_bytes": (_c_size_t, [_c_size_t]),
    "zs_bce_logits": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_float, _c_float, _c_void_p, _c_void_p,
                               _c_void_p]),
    "zs_bce_logits_bwd": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_float, _c_float, _c_void_p, _c_void_p,
                                   _c_void_p]),
    "zs_midas_loss_workspace_bytes": (_c_size_t, [_c_int]),
    "zs_midas_loss": (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_float, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_midas_loss_bwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_float, _c_int, _c_int, _c_void_p, _c_void_p,
                                                                    _c_void_p, _c_void_p]),
    "zs_intr_loss": (_c_int, [_c_void_p] * 3 + [_c_size_t, _c_void_p, _c_void_p]),
    "zs_intr_loss_bwd": (_c_int, [_c_void_p] * 3 + [_c_size_t] + [_c_void_p] * 4),
    "zs_multi_tensor_chunk_elems": (_c_int, []),
    "zs_adamw_multi": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_float, _c_float, _c_float, _c_int,
                                _c_void_p, _c_void_p, _c_void_p]),
    "zs_copy_multi": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_float, _c_void_p]),
    "zs_sumsq_multi": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_batch_norm_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "zs_batch_norm_train": (_c_int, [_c_void_p] * 9 + [_c_int, _c_int, _c_float, _c_float, _c_int, _c_void_p,
                                                       _c_void_p]),
    "zs_batch_norm_bwd": (_c_int, [_c_void_p] * 10 + [_c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_group_norm_bwd_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "zs_group_norm_bwd": (_c_int, [_c_void_p] * 8 + [_c_int, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "zs_max_pool_bwd_nhwc": (_c_int, [_c_void_p] * 3 + [_c_int] * 10 + [_c_void_p]),
    "zs_global_mean_bwd_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_upsample2x_bwd_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_nhwc_to_nchw_masked": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "zs_seen_surface_bwd": (_c_int, [_c_void_p] * 7 + [_c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "zs_intr_param2mtx_bwd": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "zs_window_tokens_bwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_void_p]),
    "zs_coord_dsp2_bwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p]),
    "zs_transform_points": (_c_int, [_c_void_p] * 5 + [_c_int, _c_int, _c_void_p]),
    "zs_resize_bilinear_nhwc": (_c_int, [_c_void_p, _c_void_p] + [_c_int] * 6 + [_c_void_p]),
    "zs_readout_concat_bwd": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
}

ABI_VERSION = 38
_lib = None


class ZeroShapeHipError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises ZeroShapeHipError when it is absent:
    build it with ``python -m zeroshape_amd.build`` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZeroShapeHipError(
            "HIP library not built: %s is missing (run `python -m zeroshape_amd.build`). "
            "zeroshape_amd has no CPU/PyTorch fallback for this path." % LIB_PATH)
    # torch FIRST: it ships its own libamdhip64 in torch/lib.  Loaded before torch, this library binds the system's copy and the
    # process ends up with two HIP runtimes - every launch on a torch stream then fails with "no ROCm-capable device is
    # detected" (`python __graft_entry__.py smoke`: build() loaded the library before smoke() imported torch)
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise ZeroShapeHipError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ZeroShapeHipError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    v = lib.zs_abi_version()
    if v != ABI_VERSION:
        raise ZeroShapeHipError("ABI mismatch: library %d, binding %d" % (v, ABI_VERSION))
    _lib = lib
    return lib


CALLS = [0]         # C-ABI calls checked so far (launch counting: bench legs report calls per forward / per step)


def check(rc, what):
    CALLS[0] += 1
    if rc != 1:
        msg = load().zs_last_error()
        raise ZeroShapeHipError("%s failed: %s" % (what, (msg or b"").decode(errors="replace")))


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream_ptr(device=None):
    """The HIP stream torch would launch on for `device` right now, as a ctypes pointer (the raw-stream query: a third
    of the cost of torch.cuda.current_stream(...).cuda_stream - it is paid once per launch, ~2,000 times per training step)."""
    import torch
    idx = getattr(device, "index", device)
    if idx is None:
        idx = torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


_NULL_GUARD = contextlib.nullcontext()


def on(device):
    """`with _lib.on(t.device):` = `with torch.cuda.device(t.device):` without the context switch when that device is
    already current (one process per GPU: always) - 0.4 instead of 1.4 us per launch on the host."""
    import torch
    idx = getattr(device, "index", device)
    if idx is None or idx == torch.cuda.current_device():
        return _NULL_GUARD
    return torch.cuda.device(device)
