"""ctypes loader of libmvp_hip.so -- the C-ABI drop-in boundary (include/mvp_hip.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on the GPU
the call raises.  PyTorch is only used for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MVP_LIBRARY') or os.path.join(_HERE, 'libmvp_hip.so')  # MVP_LIBRARY: an experiment build of the same ABI (tools/exp)
_lib = None
MLP_PRECISIONS = {'fp32': 0, 'bf16': 1, 'bf16x3': 3, 'bf16x6': 6}

_i64 = ctypes.c_int64
_ptr = ctypes.c_void_p
_f32 = ctypes.c_float

# name -> argtypes (restype is always int); mirrors include/mvp_hip.h one to one
_SIGNATURES = {
    'mvp_fps_f32': [_ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_fps_f64': [_ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_fps_shape_f32': [_ptr, _i64, _i64, _i64, _i64, _ptr, ctypes.c_int, _ptr],
    'mvp_fps_shape_f64': [_ptr, _i64, _i64, _i64, _i64, _ptr, ctypes.c_int, _ptr],
    'mvp_fps_checked_f32': [_ptr, _i64, _i64, _i64, _i64, _ptr, ctypes.c_int, _ptr, _ptr],
    'mvp_fps_centroid_levels_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_ball_query_f32': [_ptr, _ptr, _i64, _i64, _i64, _f32, _i64, _ptr, _ptr],
    'mvp_ball_query_f64': [_ptr, _ptr, _i64, _i64, _i64, _f32, _i64, _ptr, _ptr],
    'mvp_ball_query_distance_f32': [_ptr, _ptr, _i64, _i64, _i64, _f32, _i64, _ptr, _ptr, _ptr],
    'mvp_ball_query_distance_f64': [_ptr, _ptr, _i64, _i64, _i64, _f32, _i64, _ptr, _ptr, _ptr],
    'mvp_ball_query_grid_f32': [_ptr, _ptr, _i64, _i64, _i64, _f32, _i64, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_knn3_grid_f32': [_ptr, _ptr, _i64, _i64, _i64, _f32, _ptr, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_group_points_forward_strided_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_strided_f32': [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_forward_strided_f32': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_strided_f32': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_forward_strided_f64': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_forward_strided_bf16': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_strided_f64': [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_strided_bf16': [_ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_forward_strided_f64': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_forward_strided_bf16': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_strided_f64': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_strided_bf16': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_forward_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_forward_f64': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_forward_bf16': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_f64': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_points_backward_bf16': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_knn_distance_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_knn_distance_f64': [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_knn3_weights_f32': [_ptr, _ptr, _i64, _i64, _i64, _f32, _ptr, _ptr, _ptr, _ptr],
    'mvp_interpolate_forward_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_forward_f64': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_forward_bf16': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_f64': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interpolate_backward_bf16': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_unproject_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_unproject_u16': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_pixel_knn_bruteforce_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_pixel_knn_projective_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_lift_gather_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_lift_gather_backward_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_lift_f32': [_ptr, ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr,
                     _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_lift_aug_f32': [_ptr, ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr,
                         _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_rotate_rows_f32': [_ptr, _ptr, _i64, _i64, _ptr, _ptr],
    'mvp_copy_slices_f32': [_ptr, _i64, _ptr],
    'mvp_group_rows_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_rows_backward_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_group_lin_rows_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_group_lin_rows_bn_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_colstats_f32': [_ptr, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_relation_rows_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interp_rows_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_mlp_forward_bf16': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, ctypes.c_int, _ptr, _i64, _ptr],
    'mvp_adam_step_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64] + [ctypes.c_double] * 6 + [_ptr],
    'mvp_csr_build_i64': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr],
    'mvp_csr_build_sorted_i64': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr],
    'mvp_gather_rows_backward_csr_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_gather_rows_backward_csr_finish_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_interp_add_rows_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr],
    'mvp_interp_add_rows_bn_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_interp_rows_backward_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    'mvp_bn_rows_forward_f32': [_ptr, _ptr, _ptr, _i64, _i64, _i64, ctypes.c_int, _f32, _f32, ctypes.c_int, _ptr, _ptr, _ptr, _ptr,
                                _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_bn_rows_backward_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int, _ptr,
                                 _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_bn_rows_forward_dropout_f32': [_ptr, _ptr, _ptr, _i64, _i64, ctypes.c_int, _f32, _f32, ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                        _ptr, _f32, ctypes.c_uint64, _ptr],
    'mvp_bn_rows_backward_dropout_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, ctypes.c_int, ctypes.c_int, _ptr, _ptr, _ptr, _ptr,
                                         _ptr, _f32, ctypes.c_uint64, _ptr],
    'mvp_bn_rows_backward_finish_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_bn_finalize_f32': [_ptr, _i64, _i64, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_mlp_weight_grad_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_mlp_weight_grad_ws_f32': [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr],
    'mvp_mlp_input_grad_f32': [_ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_mlp_input_grad_dropout_f32': [_ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, ctypes.c_uint64, _ptr, _ptr, _ptr, _ptr],
    'mvp_mlp_layer_backward_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64,
                                   _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_mlp_layer_backward_ws_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64,
                                   _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_mlp_forward_pool_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32,
                                 _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_pool_finalize_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, ctypes.c_int, _ptr, _ptr, _ptr, _ptr],
    'mvp_pool_backward_stats_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, ctypes.c_int, _ptr, _ptr, _ptr],
    'mvp_sa_fused_forward_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr,
                                 _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_sa_train_forward_f32': [ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr,
                                 _ptr, _ptr, _ptr, _i64, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_sa_train_backward_f32': [ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr,
                                  _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr,
                                  _ptr, _ptr],
    'mvp_sa_train_backward1_f32': [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr,
                                   ctypes.c_int, _ptr, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_pn2_plan_f32': [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, ctypes.c_int, ctypes.c_int, _f32, _ptr, _i64, _ptr, _ptr, _ptr],
    'mvp_sa_geom_sums_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_sa_train_stats1_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_sa_train_stats1_ws_f32': [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr],
    'mvp_mlp_forward_bn_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr,
                               _ptr, _ptr, _ptr],
    'mvp_mlp_forward_rel_bn_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_relation4_rows_f32': [_ptr, _ptr, _i64, _i64, _ptr, _ptr],
    'mvp_mlp_forward_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    'mvp_vote_accumulate_f32': [_ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_vote_gather_f32': [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_vote_finish_f32': [_ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr],
    'mvp_seg_loss_f32': [_ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _ptr],
    'mvp_seg_loss_backward_f32': [_ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    'mvp_seg_confusion_f32': [_ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr],
}
# the shared-MLP entry points with the precision as arguments (csrc/mlp_prec.hip): base parameters + (precision, precision_backward)
for _n in ['mvp_mlp_forward_f32', 'mvp_mlp_forward_bn_f32', 'mvp_mlp_forward_rel_bn_f32', 'mvp_mlp_forward_pool_f32', 'mvp_mlp_input_grad_f32', 'mvp_mlp_input_grad_dropout_f32',
           'mvp_mlp_weight_grad_f32', 'mvp_mlp_weight_grad_ws_f32', 'mvp_mlp_layer_backward_f32', 'mvp_mlp_layer_backward_ws_f32',
           'mvp_sa_fused_forward_f32', 'mvp_sa_train_forward_f32', 'mvp_sa_train_backward_f32']:
    _SIGNATURES[_n[:-4] + '_p_f32'] = _SIGNATURES[_n][:-1] + [ctypes.c_int, ctypes.c_int, _ptr]
# (precision as arguments from the start: there is no process-default twin of this one)
_SIGNATURES['mvp_mlp_layer_backward_wide_p_f32'] = [_ptr] * 9 + [ctypes.c_int, ctypes.c_int, _f32, ctypes.c_uint64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64,
                                                    _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, ctypes.c_int, ctypes.c_int, _ptr]
_SIGNATURES['mvp_mlp_layer_backward_wide_pooled_p_f32'] = [_ptr] * 9 + [ctypes.c_int, ctypes.c_int, _i64, _f32, ctypes.c_uint64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64,
                                                           _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, ctypes.c_int, ctypes.c_int, _ptr]
_SIGNATURES['mvp_mlp_weight_grad_finish_p_f32'] = [_ptr] * 6 + [ctypes.c_int, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, ctypes.c_int, ctypes.c_int, _ptr]
_SIGNATURES['mvp_mlp_weight_grad_finish_rel_p_f32'] = [_ptr] * 6 + [ctypes.c_int, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64, ctypes.c_int, ctypes.c_int, _ptr]
_SIGNATURES['mvp_mlp_weight_grad_finish_act_p_f32'] = [_ptr] * 6 + [ctypes.c_int, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64, ctypes.c_int,
                                                        ctypes.c_int, _ptr]
_SIGNATURES['mvp_mlp_input_grad_wide_p_f32'] = [_ptr] * 8 + [ctypes.c_int, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64,
                                                 ctypes.c_int, ctypes.c_int, _ptr]
EXPORTS = ['mvp_version', 'mvp_strerror', 'mvp_lift_workspace_bytes', 'mvp_ball_query_grid_workspace', 'mvp_knn3_grid_workspace', 'mvp_mlp_weight_grad_workspace_floats', 'mvp_mlp_input_grad_wide_workspace_bytes', 'mvp_group_lin_partial_count', 'mvp_colstats_partial_count',
           'mvp_set_mlp_precision', 'mvp_get_mlp_precision', 'mvp_mlp_layer_backward_partial_count', 'mvp_set_mlp_stream', 'mvp_set_mlp_precision_backward', 'mvp_get_mlp_precision_backward', 'mvp_mlp_precision_scope', 'mvp_set_fps_mode', 'mvp_fps_debug_spin_limit', 'mvp_fps_last_kernel'] + sorted(_SIGNATURES)


def lib():
    """Load libmvp_hip.so once.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libmvp_hip.so is not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                               'or `make -C mvpnet_amd/csrc` ({})'.format(LIB_PATH))
        handle = ctypes.CDLL(LIB_PATH)
        handle.mvp_version.restype = ctypes.c_char_p
        handle.mvp_strerror.restype = ctypes.c_char_p
        handle.mvp_strerror.argtypes = [ctypes.c_int]
        handle.mvp_lift_workspace_bytes.restype = ctypes.c_int64
        handle.mvp_lift_workspace_bytes.argtypes = [_i64, _i64, _i64, _i64, _i64]
        handle.mvp_ball_query_grid_workspace.restype = ctypes.c_int64
        handle.mvp_ball_query_grid_workspace.argtypes = [_i64, _i64, _i64]
        handle.mvp_knn3_grid_workspace.restype = ctypes.c_int64
        handle.mvp_knn3_grid_workspace.argtypes = [_i64, _i64, _i64]
        handle.mvp_group_lin_partial_count.restype = ctypes.c_int64
        handle.mvp_group_lin_partial_count.argtypes = [_i64, _i64, _i64, _i64]
        handle.mvp_mlp_weight_grad_workspace_floats.restype = ctypes.c_int64
        handle.mvp_mlp_weight_grad_workspace_floats.argtypes = []
        handle.mvp_mlp_input_grad_wide_workspace_bytes.restype = ctypes.c_int64
        handle.mvp_mlp_input_grad_wide_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
        handle.mvp_colstats_partial_count.restype = ctypes.c_int64
        handle.mvp_colstats_partial_count.argtypes = [_i64, _i64]
        handle.mvp_set_mlp_precision.restype = ctypes.c_int
        handle.mvp_set_mlp_precision.argtypes = [ctypes.c_int, ctypes.c_int]
        handle.mvp_get_mlp_precision.restype = ctypes.c_int
        handle.mvp_get_mlp_precision.argtypes = []
        handle.mvp_mlp_layer_backward_partial_count.restype = ctypes.c_int64
        handle.mvp_mlp_layer_backward_partial_count.argtypes = [_i64, _i64]
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = handle
        handle.mvp_set_mlp_stream.restype = ctypes.c_int
        handle.mvp_set_mlp_stream.argtypes = [ctypes.c_int]
        handle.mvp_set_fps_mode.restype = ctypes.c_int
        handle.mvp_set_fps_mode.argtypes = [ctypes.c_int]
        handle.mvp_fps_debug_spin_limit.restype = ctypes.c_int
        handle.mvp_fps_debug_spin_limit.argtypes = [ctypes.c_int]
        handle.mvp_fps_last_kernel.restype = ctypes.c_int
        handle.mvp_fps_last_kernel.argtypes = []
        if os.environ.get('MVP_MLP_STREAM') is not None:
            handle.mvp_set_mlp_stream(int(os.environ['MVP_MLP_STREAM']))
        handle.mvp_set_mlp_precision_backward.restype = ctypes.c_int
        handle.mvp_set_mlp_precision_backward.argtypes = [ctypes.c_int]
        handle.mvp_get_mlp_precision_backward.restype = ctypes.c_int
        handle.mvp_get_mlp_precision_backward.argtypes = []
        handle.mvp_mlp_precision_scope.restype = ctypes.c_int
        handle.mvp_mlp_precision_scope.argtypes = [ctypes.c_int, ctypes.c_int]
        if os.environ.get('MVP_MLP_PRECISION_BWD'):
            check(handle.mvp_set_mlp_precision_backward(MLP_PRECISIONS[os.environ['MVP_MLP_PRECISION_BWD']]), 'mvp_set_mlp_precision_backward')
        env = os.environ.get('MVP_MLP_PRECISION')
        if env:
            set_mlp_precision(env, int(os.environ.get('MVP_MLP_MIN_WIDTH', '0')))
    return _lib




def set_mlp_precision(name, min_width=0):
    """Contraction precision of the shared-MLP kernels (mvp_set_mlp_precision): 'fp32' (fp32 MFMA), 'bf16x6' (split-bf16,
    3 pieces / 6 products, fp32-level accuracy), 'bf16x3' (2 pieces / 3 products, ~2^-17 per product), 'bf16' (operands rounded to
    bf16 once, 1 product, ~2^-9 per product: the accuracy of a bf16 autocast with fp32 accumulation and storage -- opt-in, outside the
    fp32 parity bar).  Layers narrower than `min_width` channels stay on the fp32 MFMA.  Also settable through MVP_MLP_PRECISION / MVP_MLP_MIN_WIDTH."""
    if name not in MLP_PRECISIONS:
        raise ValueError('mlp precision must be one of {}'.format(sorted(MLP_PRECISIONS)))
    check(lib().mvp_set_mlp_precision(MLP_PRECISIONS[name], int(min_width)), 'mvp_set_mlp_precision')
    global _mlp_min_width
    _mlp_min_width = int(min_width)


_mlp_min_width = 0


def mlp_min_width():
    """The `min_width` of the last set_mlp_precision (layers narrower than it stay on the fp32 MFMA): the hosts of the split-bf16-only entry
    points (rows.wide_backward_ok, the finish-on-load weight gradient) leave such layers to the per-layer kernels."""
    return _mlp_min_width


def set_mlp_precision_backward(name):
    """Split of the gradient contractions (weight / input gradient, one-kernel layer backward) while the forward runs a split
    precision: 'bf16x3' (default: 2^-17 per product, invisible next to the ~1 % fp32 noise of these gradients), 'bf16x6' or 'bf16'."""
    if name not in ('bf16', 'bf16x3', 'bf16x6'):
        raise ValueError("backward mlp precision must be 'bf16', 'bf16x3' or 'bf16x6'")
    check(lib().mvp_set_mlp_precision_backward(MLP_PRECISIONS[name]), 'mvp_set_mlp_precision_backward')


class mlp_precision:
    """`with mlp_precision('fp32'): ...` / `with mlp_precision('bf16x6', backward='bf16x6'): ...` -- the contraction precision of the
    shared-MLP launches made by THIS thread inside the block (mvp_mlp_precision_scope: a thread-local override, the process-wide defaults
    of set_mlp_precision[_backward] are not touched).  An autograd node built inside the block records the pair and hands it to its
    backward launches as arguments (the `_p_f32` entry points), so `loss.backward()` may run outside the block and on autograd's own
    thread: forward and backward of a node always agree."""

    def __init__(self, forward=None, backward=None):
        self.terms = -1 if forward is None else MLP_PRECISIONS[forward]
        self.bwd = -1 if backward is None else MLP_PRECISIONS[backward]
        if self.bwd == 0:
            raise ValueError("backward mlp precision must be 'bf16', 'bf16x3' or 'bf16x6'")

    def __enter__(self):
        old = lib().mvp_mlp_precision_scope(self.terms, self.bwd)
        if old < 0:
            check(old, 'mvp_mlp_precision_scope')
        self.old = (old // 16 - 1, old % 16 - 1)
        return self

    def __exit__(self, *exc):
        lib().mvp_mlp_precision_scope(*self.old)


def current_precision():
    """(forward terms, backward terms) the CALLING thread's shared-MLP launches use right now (thread-local scope, else the process
    defaults): what an autograd node records in forward and hands to its backward calls (`prec=` of call / call_on), which autograd
    issues from its own thread."""
    h = lib()
    return (h.mvp_get_mlp_precision(), h.mvp_get_mlp_precision_backward())


def get_mlp_precision():
    terms = lib().mvp_get_mlp_precision()
    return {v: k for k, v in MLP_PRECISIONS.items()}[terms]


_FPS_STATUS = {}


def fps_status(device):
    """The per-device status word the sampling calls of this process hand to mvp_fps_checked_f32 (int32, sticky: 1 once a
    multi-workgroup launch timed out and was repaired by the one-workgroup kernel)."""
    t = _FPS_STATUS.get(device)
    if t is None:
        t = _FPS_STATUS[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def fps_timed_out(device=None, reset=False):
    """True when a multi-workgroup sampling launch on `device` gave up waiting for its partner workgroups since the last reset (its
    result was re-computed by the one-workgroup kernel: the indices were right, the launch took ~5x longer).  Synchronises."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    t = _FPS_STATUS.get(device)
    if t is None:
        return False
    hit = bool(int(t.item()))
    if reset:
        t.zero_()
    return hit


def check(code, what):
    if code != 0:
        raise RuntimeError('{} failed: {} (code {})'.format(what, lib().mvp_strerror(code).decode(), code))


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    """Device address of tensor t as a plain int (ctypes converts it for a c_void_p parameter; None = NULL): no ctypes object per
    argument -- a training step passes ~600 of them."""
    return None if t is None else t.data_ptr()


def ptr_at(t, element_offset):
    """Device pointer `element_offset` elements into contiguous tensor t."""
    return t.data_ptr() + int(element_offset) * t.element_size()


def require_gpu(*tensors):
    """CHECK_INPUT of the reference (fps_kernel.cu:12-14): CUDA/HIP tensor + contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('mvpnet_amd ops run on the GPU only (got a {} tensor); there is no CPU fallback'.format(t.device))
        if not t.is_contiguous():
            raise RuntimeError('tensor must be contiguous')


def suffix(t):
    if t.dtype == torch.float32:
        return 'f32'
    if t.dtype == torch.float64:
        return 'f64'
    raise RuntimeError('expected a float32 or float64 tensor, got {}'.format(t.dtype))


def value_suffix(t):
    """Entry-point suffix of the ops that also exist for bfloat16 VALUES (group_points, interpolate)."""
    return 'bf16' if t.dtype == torch.bfloat16 else suffix(t)


_FN = {}  # entry point name -> bound ctypes function (one dict lookup per launch instead of a getattr on the CDLL)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)  # current stream handle as an int, no torch.cuda.Stream object per launch
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _fn(name):
    f = _FN.get(name)
    if f is None:
        f = _FN[name] = getattr(_lib if _lib is not None else lib(), name)
    return f


# REPRODUCIBLE MODE (MVP_DETERMINISTIC=1 / set_deterministic(True)): the weight-gradient kernels flush their workgroups' partial tiles
# through a workspace and an ordered reduction (mvp_mlp_weight_grad_ws_f32, mvp_mlp_layer_backward_ws_f32) instead of fp32 atomics -- the
# last float32 atomics of a training step; with it every parameter after a step is the same bit for bit in every run
# (tests/test_determinism_gpu.py).  Costs ~2 % of the step (the reductions are extra launches, five of them on the critical stream):
# 8.20 -> 8.39 ms, so it is off by default; the atomics' run-to-run noise is ~1e-7 relative per gradient element.  One workspace per
# (device, stream) -- a kernel and its reduction run in order on that stream --, handed out here to every caller of the two entry points.
DW_WORKSPACE = os.environ.get('MVP_DETERMINISTIC', '0') == '1'


def set_deterministic(flag):
    """-> the previous setting."""
    global DW_WORKSPACE
    old, DW_WORKSPACE = DW_WORKSPACE, bool(flag)
    return old


_DW_WS = {}
_DW_WS_NAMES = {'mvp_mlp_weight_grad_f32': 'mvp_mlp_weight_grad_ws_f32', 'mvp_mlp_layer_backward_f32': 'mvp_mlp_layer_backward_ws_f32'}


def dw_workspace(index, stream_handle):
    """The weight-gradient workspace of (device, stream) in the reproducible mode (a kernel and its reduction run in order on that stream)."""
    ws = _DW_WS.get((index, stream_handle))
    if ws is None:
        ws = _DW_WS[(index, stream_handle)] = torch.empty(lib().mvp_mlp_weight_grad_workspace_floats(), dtype=torch.float32,
                                                          device=torch.device('cuda', index))
    return ws


def current_dw_workspace(device):
    """(pointer, floats) of the calling stream's weight-gradient workspace in the reproducible mode, (None, 0) otherwise: for entry points
    that take the workspace as plain arguments (mvp_mlp_layer_backward_wide_p_f32)."""
    if not DW_WORKSPACE:
        return None, 0
    index = device.index
    handle = _raw_stream(index) if (_raw_stream is not None and index == _raw_device()) else torch.cuda.current_stream(device).cuda_stream
    ws = dw_workspace(index, handle)
    return ws.data_ptr(), ws.numel()


def _with_dw_workspace(index, stream_handle, args):
    ws = dw_workspace(index, stream_handle)
    return args + (ws.data_ptr(), ws.numel())


def call_on(stream, name, *args, prec=None):
    """Invoke `name(*args, stream)` on the given torch.cuda.Stream of the CURRENT device (no stream-context switch on the host).
    prec = (precision, precision_backward): the `_p_f32` variant of a shared-MLP entry point, precision as arguments."""
    if DW_WORKSPACE and name in _DW_WS_NAMES:
        name, args = _DW_WS_NAMES[name], _with_dw_workspace(stream.device.index, stream.cuda_stream, args)
    if prec is not None:
        name, args = name[:-4] + '_p_f32', args + prec
    code = _fn(name)(*args, stream.cuda_stream)
    if code != 0:
        check(code, name)


def call(name, tensor_for_device, *args, prec=None):
    """Invoke `name(*args, stream)` on the current stream of tensor_for_device's device.
    prec = (precision, precision_backward): the `_p_f32` variant of a shared-MLP entry point, precision as arguments."""
    index = tensor_for_device.device.index
    if DW_WORKSPACE and name in _DW_WS_NAMES:
        handle = _raw_stream(index) if _raw_stream is not None else torch.cuda.current_stream(tensor_for_device.device).cuda_stream
        name, args = _DW_WS_NAMES[name], _with_dw_workspace(index, handle, args)
    if prec is not None:
        name, args = name[:-4] + '_p_f32', args + prec
    if _raw_stream is not None and index == _raw_device():  # the usual case (one process per GPU): no device guard, no Stream object
        code = _fn(name)(*args, _raw_stream(index))
    elif index == torch.cuda.current_device():
        code = _fn(name)(*args, torch.cuda.current_stream(tensor_for_device.device).cuda_stream)
    else:
        with torch.cuda.device(tensor_for_device.device):
            code = _fn(name)(*args, stream_of(tensor_for_device))
    if code != 0:
        check(code, name)
