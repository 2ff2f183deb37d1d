"""``pointnet2_ops._ext`` — MI355X drop-in for the reference's pybind11 module.

The reference builds ``pointnet2_ops._ext`` from CUDA sources
(EXT/src/bindings.cpp:6-19; EXT = scene_graph_prediction/pointnet2_dir/
pointnet2_ops_lib/pointnet2_ops/_ext-src).  This module exports the same nine
callables with the same argument order, dtypes, allocation behaviour and error
behaviour, but forwards to ``libpn2_hip.so`` (hand-written gfx950 kernels behind
the C ABI of ``include/pn2_hip.h``) through ctypes:

* outputs are allocated here on the input's device (the reference allocates
  them with ``torch::zeros`` inside the C++ wrapper, e.g. EXT/src/ball_query.cpp:19-21);
* kernels are enqueued on torch's current HIP stream, no host sync
  (reference: ``at::cuda::getCurrentCUDAStream()``);
* contiguity / dtype violations and CPU tensors raise ``RuntimeError`` (the
  reference's AT_ASSERTs, EXT/include/utils.h:5-25 and "CPU not supported" in
  every op, e.g. EXT/src/ball_query.cpp:28).

There is NO fallback: if the shared library is missing or a kernel launch
fails this module raises.  torch is used for device memory and streams only.
"""
import collections
import contextlib
import ctypes
import os
import threading

import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("PN2_HIP_LIB") or os.path.join(_PKG_DIR, "libpn2_hip.so")   # override: kernel experiments only

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"pointnet2_ops._ext: {LIB_PATH} not found. Build it with "
        f"`make -C {os.path.join(_PKG_DIR, 'csrc')}` (or __graft_entry__.build()); "
        "there is no CPU/PyTorch fallback."
    )

_lib = ctypes.CDLL(LIB_PATH)
_lib.pn2_abi_version.restype = ctypes.c_int
_ABI = 11
if int(_lib.pn2_abi_version()) != _ABI:
    raise ImportError(f"pointnet2_ops._ext: {LIB_PATH} has ABI version {int(_lib.pn2_abi_version())}, this binding needs {_ABI}: "
                      f"rebuild it (`make -C {os.path.join(_PKG_DIR, 'csrc')}`)")

_c_int, _c_i64, _c_f32, _c_vp, _c_sz = (ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                        ctypes.c_void_p, ctypes.c_size_t)

# symbol -> argtypes; every entry point returns int unless noted
_SIGNATURES = {
    "pn2_furthest_point_sampling": [_c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp, _c_vp],
    "pn2_furthest_point_sampling_ex": [_c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp, _c_int, _c_vp],
    "pn2_furthest_point_sampling_ordered": [_c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp, _c_int, _c_vp],
    "pn2_gather_points": [_c_int] * 4 + [_c_vp] * 4,
    "pn2_gather_points_grad": [_c_int] * 4 + [_c_vp] * 4,
    "pn2_ball_query": [_c_int, _c_int, _c_int, _c_f32, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_ball_query_ws": [_c_int, _c_int, _c_int, _c_f32, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp],
    "pn2_ball_query_algo": [_c_int, _c_int, _c_int, _c_int, _c_f32, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp],
    "pn2_ball_query_unique_resample": [ctypes.c_longlong, _c_int, ctypes.c_uint, _c_vp, _c_vp, _c_vp],
    "pn2_ball_query_group": [_c_int, _c_int, _c_int, _c_f32, _c_int, _c_int, _c_int, _c_int] + [_c_vp] * 6 + [_c_sz, _c_int, _c_vp],
    "pn2_group_points": [_c_int] * 5 + [_c_vp] * 4,
    "pn2_group_points_grad": [_c_int] * 5 + [_c_vp] * 4,
    "pn2_three_nn": [_c_int] * 3 + [_c_vp] * 5,
    "pn2_three_interpolate": [_c_int] * 4 + [_c_vp] * 5,
    "pn2_three_interpolate_grad": [_c_int] * 4 + [_c_vp] * 5,
    "pn2_group_concat_rows": [_c_int] * 7 + [_c_f32] + [_c_vp] * 6,
    "pn2_group_rows_grad": [_c_int] * 7 + [_c_vp] * 4,
    "pn2_bn_running_update": [_c_int, _c_int, _c_vp, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_group_inverse_index": [_c_int] * 4 + [_c_vp] * 4 + [_c_sz, _c_vp],
    "pn2_x3_pack_weight": [_c_int] * 4 + [_c_vp] * 3,
    "pn2_x3_pack_first": [_c_int] * 2 + [_c_vp] * 5,
    "pn2_x3_bwd_fold_first": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 + [_c_int] + [_c_vp] * 4,
    "pn2_x3_gemm_first": [ctypes.c_longlong] + [_c_int] * 3 + [_c_vp] * 7,
    "pn2_group_points_grad_csr": [_c_int] * 5 + [_c_vp] * 5,
    "pn2_three_interpolate_grad_csr": [_c_int] * 4 + [_c_vp] * 6,
    "pn2_x3_gemm": [ctypes.c_longlong] + [_c_int] * 4 + [_c_vp] * 13 + [_c_int, _c_vp, _c_vp],
    "pn2_sa_eval_x3": [_c_int] * 6 + [_c_vp] * 5 + [_c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_vp],
    "pn2_group_rows_grad_csr": [_c_int] * 5 + [_c_i64] + [_c_vp] * 5,
    "pn2_group_lift_rows": [_c_int] * 6 + [_c_f32] + [_c_vp] * 8,
    "pn2_group_lift_rows_grad": [_c_int] * 6 + [_c_f32] + [_c_vp] * 11 + [_c_sz, _c_vp],
    "pn2_group_lift_rows_bf16": [_c_int] * 6 + [_c_f32] + [_c_vp] * 8,
    "pn2_group_lift_rows_grad_bf16": [_c_int] * 6 + [_c_f32] + [_c_vp] * 11 + [_c_sz, _c_vp],
    "pn2_group_lift_rows_seg": [_c_int] * 6 + [_c_f32] + [_c_vp] * 6 + [_c_int, _c_vp, _c_vp, _c_int, _c_int, _c_vp],
    "pn2_group_lift_rows_grad_seg": [_c_int] * 6 + [_c_f32] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 8 + [_c_int, _c_int, _c_vp, _c_sz, _c_vp],
    "pn2_group_rows_grad_csr_bf16": [_c_int] * 5 + [_c_i64] + [_c_vp] * 5,
    "pn2_group_rows_grad_bf16": [_c_int] * 7 + [_c_vp] * 4,
    "pn2_rows_max": [_c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_rows_max_grad": [_c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_three_interpolate_rows": [_c_int] * 6 + [_c_vp] * 5,
    "pn2_three_interpolate_rows_grad": [_c_int] * 6 + [_c_vp] * 5,
    "pn2_three_interpolate_rows_grad_csr": [_c_int] * 6 + [_c_vp] * 6,
    "pn2_lift_split_weight": [_c_int] * 2 + [_c_vp] * 5,
    "pn2_lift_dw_assemble": [_c_int] * 2 + [_c_vp] * 6,
    "pn2_lift_points": [_c_int] * 5 + [_c_f32] + [_c_vp] * 7,
    "pn2_group_lift_stats": [_c_int] * 5 + [_c_vp] * 6,
    "pn2_mlp_gemm_lift": [ctypes.c_longlong, _c_int, _c_int, ctypes.c_longlong] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 5,
    "pn2_mlp_wgrad_lift": [ctypes.c_longlong, _c_int, _c_int, ctypes.c_longlong] + [_c_vp] * 6 + [_c_int] + [_c_vp] * 3,
    "pn2_mlp_dgrad_lift": [ctypes.c_longlong, _c_int, _c_int, ctypes.c_longlong] + [_c_vp] * 9 + [_c_int] + [_c_vp] * 2,
    "pn2_gather_rows": [_c_i64, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_scatter_add_rows": [_c_i64, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_segment_sum_rows": [_c_i64, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_gather2_add_rows": [_c_i64, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_segment_sum2_rows": [_c_i64, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_segment_bn_rows": [_c_i64, _c_int, _c_int, _c_int, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_f32, _c_int, _c_vp, _c_vp,
                            _c_vp, _c_vp],
    "pn2_segment_bn_rows_grad": [_c_i64, _c_int, _c_int, _c_int, _c_i64] + [_c_vp] * 7 + [_c_int] + [_c_vp] * 4,
    "pn2_gcn_linear": [_c_i64, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                       _c_vp, _c_vp, _c_f32, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_gcn_linear_grad_w": [_c_i64, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 5 + [_c_int, _c_vp, _c_vp,
                              _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 5 + [_c_vp],
    "pn2_gcn_linear_grad_x": [_c_i64, _c_int, _c_int, _c_int] + [_c_vp] * 8 + [_c_int, _c_int, _c_vp],
    "pn2_gcn_edge_slice": [_c_i64, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp],
    "pn2_gcn_layer_forward": [_c_vp, _c_vp],
    "pn2_gcn_layer_backward": [_c_vp, _c_vp],
    "pn2_segment_bn_running_update": [_c_i64, _c_int, _c_vp, _c_vp, _c_vp, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_prep_object_boxes": [_c_int, _c_int, _c_int, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_prep_chunk_counts": [_c_int] * 4 + [_c_vp] * 6,
    "pn2_prep_select": [_c_int] * 6 + [ctypes.c_uint] + [_c_vp] * 7,
    "pn2_prep_gather_normalise": [_c_int] * 5 + [_c_vp] * 7,
    "pn2_prep_voxel_keys": [_c_int, _c_int, _c_vp, _c_vp, ctypes.c_double, _c_vp, _c_vp],
    "pn2_floyd_warshall": [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_gen_edge_input": [_c_int] * 4 + [_c_vp] * 4,
    "pn2_mlp_gemm": [ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int] + [_c_vp] * 7 + [_c_int] + [_c_vp] * 6,
    "pn2_mlp_wgrad": [ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4,
    "pn2_mlp_bwd_fused": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 7,
    "pn2_mlp_bwd_fused_fold": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 +
                              [_c_int] + [_c_vp] * 4,
    "pn2_rows_gram": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp],
    "pn2_first_layer_stats": [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_mlp_gemm_first": [ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int] + [_c_vp] * 8,
    "pn2_mlp_bwd_fused_fold_first": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 +
                                    [_c_int] + [_c_vp] * 4,
    "pn2_mlp_gemm_bf16": [ctypes.c_longlong] + [_c_int] * 8 + [_c_vp] * 7 + [_c_int] + [_c_vp] * 6,
    "pn2_mlp_bwd_bf16_fold": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 + [_c_int] +
                             [_c_vp] * 4,
    "pn2_rows_gram_bf16": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp],
    "pn2_mlp_wgrad_bf16": [ctypes.c_longlong] + [_c_int] * 6 + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4,
    "pn2_mlp_bwd_bf16": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 7,
    "pn2_mlp_gemm_first_bf16": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 7,
    "pn2_mlp_bwd_bf16_fold_first": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 + [_c_int] +
                                   [_c_vp] * 4,
    "pn2_mlp_gemm_pool_bf16": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4,
    "pn2_mlp_bwd_bf16_pool": [ctypes.c_longlong, _c_int, _c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 7,
    "pn2_bn_relu_apply_bf16": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_relu_bwd_prep_bf16": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_relu_rows_max_bf16": [ctypes.c_longlong, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_group_concat_rows_bf16": [_c_int] * 7 + [_c_f32, _c_int] + [_c_vp] * 6,
    "pn2_first_layer_dw": [_c_int, _c_int] + [_c_vp] * 6,
    "pn2_bn_finalize": [_c_int, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_bwd_consts": [_c_int, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int,
                          _c_vp, _c_vp],
    "pn2_bn_relu_apply": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_relu_bwd_prep": [ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_relu_rows_max": [ctypes.c_longlong, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_pool_bwd_prep": [ctypes.c_longlong, _c_int] + [_c_vp] * 7,
    "pn2_pool_flip_rows": [_c_int, _c_int] + [_c_vp] * 5,
    "pn2_mlp_gemm_pool": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4,
    "pn2_pool_finalize": [ctypes.c_longlong, _c_int, _c_int] + [_c_vp] * 8,
    "pn2_pool_bwd": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 10 + [_c_sz, _c_vp],
    "pn2_x3_pool_bwd": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 10 + [_c_sz, _c_vp],
    # batched scans with per-scan statistics (segment table)
    "pn2_mlp_gemm_bf16_seg": [ctypes.c_longlong] + [_c_int] * 8 + [_c_vp] * 5 + [_c_int] + [_c_vp] * 2 + [_c_int] +
                             [_c_vp] * 6 + [_c_int, ctypes.c_longlong, _c_vp],
    "pn2_mlp_wgrad_bf16_seg": [ctypes.c_longlong] + [_c_int] * 6 + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 +
                              [_c_int, ctypes.c_longlong, _c_vp],
    "pn2_mlp_bwd_bf16_seg": [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 7 +
                            [_c_int, ctypes.c_longlong, _c_vp],
    "pn2_bn_relu_rows_max_bf16_seg": [ctypes.c_longlong, _c_int, _c_int] + [_c_vp] * 6 + [_c_int, ctypes.c_longlong, _c_vp],
    "pn2_pool_bwd_prep_seg": [ctypes.c_longlong, _c_int] + [_c_vp] * 7 + [_c_int, ctypes.c_longlong, _c_int, _c_vp],
    "pn2_bn_finalize_seg": [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp],
    "pn2_bn_bwd_consts_seg": [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int,
                              _c_vp, _c_vp],
}
for _name, _args in _SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here == ABI mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = _c_int
_lib.pn2_fps_coop_status.argtypes = [_c_int, _c_vp, _c_vp]
_lib.pn2_fps_coop_status.restype = _c_int
_lib.pn2_fps_workspace_bytes.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_fps_workspace_bytes.restype = _c_sz
_lib.pn2_ball_query_workspace_bytes.argtypes = [_c_int, _c_int, _c_int, _c_f32, _c_int]
_lib.pn2_ball_query_workspace_bytes.restype = _c_sz
_lib.pn2_group_inverse_index_workspace_bytes.argtypes = [_c_int] * 4
_lib.pn2_group_inverse_index_workspace_bytes.restype = _c_sz
_lib.pn2_prep_num_chunks.argtypes = [_c_int]
_lib.pn2_prep_num_chunks.restype = _c_int
_lib.pn2_ball_query_grid_bytes.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_ball_query_grid_bytes.restype = _c_sz
_lib.pn2_ball_query_algo_bytes.argtypes = [_c_int, _c_int, _c_int, _c_int, _c_f32, _c_int]
_lib.pn2_ball_query_algo_bytes.restype = _c_sz
_lib.pn2_ball_query_auto.argtypes = [_c_int, _c_int, _c_int, _c_f32, _c_int]
_lib.pn2_ball_query_auto.restype = _c_int
_lib.pn2_group_lift_supported.argtypes = [_c_int]
_lib.pn2_group_lift_supported.restype = _c_int
_lib.pn2_mlp_lift_supported.argtypes = [_c_int] * 3
_lib.pn2_mlp_lift_supported.restype = _c_int
_lib.pn2_mlp_gemm_first_bf16_supported.argtypes = [_c_int] * 3
_lib.pn2_mlp_gemm_first_bf16_supported.restype = _c_int
_lib.pn2_mlp_gemm_pool_bf16_supported.argtypes = [_c_int] * 3
_lib.pn2_mlp_gemm_pool_bf16_supported.restype = _c_int
_lib.pn2_mlp_bwd_bf16_pool_supported.argtypes = [_c_int] * 2
_lib.pn2_mlp_bwd_bf16_pool_supported.restype = _c_int
_lib.pn2_group_lift_rows_grad_workspace_bytes.argtypes = [_c_int] * 5
_lib.pn2_group_lift_rows_grad_workspace_bytes.restype = _c_sz
_lib.pn2_ball_query_group_supported.argtypes = [_c_int, _c_int, _c_int, _c_f32, _c_int, _c_int, _c_int]
_lib.pn2_ball_query_group_supported.restype = _c_int
_lib.pn2_ball_query_group_workspace_bytes.argtypes = [_c_int, _c_int]
_lib.pn2_ball_query_group_workspace_bytes.restype = _c_sz
_lib.pn2_fps_status_offset.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_fps_status_offset.restype = ctypes.c_longlong
_lib.pn2_fps_status_offset_ex.argtypes = [_c_int, _c_int, _c_int, _c_int]
_lib.pn2_fps_status_offset_ex.restype = ctypes.c_longlong
_lib.pn2_fps_set_plan_override.argtypes = [_c_int] * 5
_lib.pn2_fps_set_plan_override.restype = _c_int
_lib.pn2_event_create.argtypes = []
_lib.pn2_event_create.restype = _c_vp
_lib.pn2_event_record.argtypes = [_c_vp, _c_vp]
_lib.pn2_event_record.restype = _c_int
_lib.pn2_event_elapsed_ms.argtypes = [_c_vp, _c_vp, ctypes.POINTER(ctypes.c_float)]
_lib.pn2_event_elapsed_ms.restype = _c_int
_lib.pn2_event_destroy.argtypes = [_c_vp]
_lib.pn2_event_destroy.restype = _c_int
_lib.pn2_fps_set_bucketing.argtypes = [_c_int]
_lib.pn2_fps_set_bucketing.restype = _c_int
_lib.pn2_fps_get_bucketing.argtypes = []
_lib.pn2_fps_get_bucketing.restype = _c_int
if os.environ.get("PN2_FPS_BUCKETING") == "0":       # measurement switch (tools, A/B runs of bench.py)
    _lib.pn2_fps_set_bucketing(0)
_lib.pn2_fps_set_multi.argtypes = [_c_int]
_lib.pn2_fps_set_multi.restype = _c_int
_lib.pn2_fps_get_multi.argtypes = []
_lib.pn2_fps_get_multi.restype = _c_int
if os.environ.get("PN2_FPS_MULTI") == "0":           # measurement switch: one sample per cluster hand-off (round 3)
    _lib.pn2_fps_set_multi(0)
_lib.pn2_fps_ordered_workspace_bytes.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_fps_ordered_workspace_bytes.restype = _c_sz
_lib.pn2_gcn_fused_supported.argtypes = [_c_int, _c_int, _c_int, _c_int]
_lib.pn2_gcn_fused_supported.restype = _c_int
_lib.pn2_gcn_layer_backward_workspace_bytes.argtypes = [ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_int, _c_int]
_lib.pn2_gcn_layer_backward_workspace_bytes.restype = _c_sz
_lib.pn2_group_lift_rows_grad_seg_workspace_bytes.argtypes = [_c_int] * 6
_lib.pn2_group_lift_rows_grad_seg_workspace_bytes.restype = _c_sz
_lib.pn2_mlp_bwd_fused_supported.argtypes = [_c_int, _c_int]
_lib.pn2_mlp_bwd_fused_supported.restype = _c_int
_lib.pn2_mlp_bwd_bf16_supported.argtypes = [_c_int, _c_int]
_lib.pn2_mlp_bwd_bf16_supported.restype = _c_int
_lib.pn2_mlp_bwd_fused_fold_supported.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_mlp_bwd_fused_fold_supported.restype = _c_int
_lib.pn2_mlp_bwd_bf16_fold_supported.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_mlp_bwd_bf16_fold_supported.restype = _c_int
_lib.pn2_mlp_gemm_first_supported.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_mlp_gemm_first_supported.restype = _c_int
_lib.pn2_pool_bwd_supported.argtypes = [_c_int, _c_int, _c_int]
_lib.pn2_pool_bwd_supported.restype = _c_int
_lib.pn2_pool_bwd_workspace_bytes.argtypes = [ctypes.c_longlong, _c_int, _c_int]
_lib.pn2_pool_bwd_workspace_bytes.restype = _c_sz
_lib.pn2_x3_weight_bytes.argtypes = [_c_int, _c_int]
_lib.pn2_x3_weight_bytes.restype = _c_sz
_lib.pn2_sa_eval_x3_supported.argtypes = [_c_int] * 6
_lib.pn2_sa_eval_x3_supported.restype = _c_int
_lib.pn2_x3_gemm_supported.argtypes = [_c_int] * 5
_lib.pn2_x3_gemm_supported.restype = _c_int
_lib.pn2_x3_gemm_first_supported.argtypes = [_c_int] * 3
_lib.pn2_x3_gemm_first_supported.restype = _c_int
_lib.pn2_abi_version.restype = _c_int
_lib.pn2_last_hip_error.restype = _c_int
_lib.pn2_strerror.argtypes = [_c_int]
_lib.pn2_strerror.restype = ctypes.c_char_p

ABI_VERSION = int(_lib.pn2_abi_version())
#: the header revision this binding was written against: a stale prebuilt libpn2_hip.so fails here with a version
#: error instead of an AttributeError on the first missing symbol
EXPECTED_ABI_VERSION = _ABI
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["pn2_fps_workspace_bytes", "pn2_abi_version", "pn2_fps_coop_status",
                                               "pn2_fps_status_offset", "pn2_fps_status_offset_ex", "pn2_fps_set_plan_override", "pn2_fps_set_bucketing", "pn2_fps_get_bucketing",
                                               "pn2_fps_set_multi", "pn2_fps_get_multi", "pn2_fps_ordered_workspace_bytes", "pn2_gcn_fused_supported", "pn2_gcn_layer_backward_workspace_bytes", "pn2_group_lift_rows_grad_seg_workspace_bytes",
                                               "pn2_event_create", "pn2_event_record", "pn2_event_elapsed_ms", "pn2_event_destroy",
                                               "pn2_ball_query_workspace_bytes", "pn2_ball_query_grid_bytes",
                                               "pn2_ball_query_algo_bytes", "pn2_ball_query_auto",
                                               "pn2_ball_query_group_supported", "pn2_ball_query_group_workspace_bytes",
                                               "pn2_group_lift_supported", "pn2_mlp_lift_supported", "pn2_mlp_gemm_pool_bf16_supported", "pn2_mlp_gemm_first_bf16_supported", "pn2_mlp_bwd_bf16_pool_supported", "pn2_group_lift_rows_grad_workspace_bytes",
                                               "pn2_prep_num_chunks", "pn2_group_inverse_index_workspace_bytes",
                                               "pn2_mlp_bwd_fused_supported", "pn2_mlp_bwd_fused_fold_supported",
                                               "pn2_mlp_gemm_first_supported", "pn2_mlp_bwd_bf16_fold_supported",
                                               "pn2_mlp_bwd_bf16_supported", "pn2_pool_bwd_supported",
                                               "pn2_pool_bwd_workspace_bytes", "pn2_x3_weight_bytes", "pn2_sa_eval_x3_supported", "pn2_x3_gemm_supported", "pn2_x3_gemm_first_supported",
                                               "pn2_last_hip_error", "pn2_strerror"])
#: the python layer may use the point-major fused entry points of this backend
HAS_ROWS = True
#: ... and the fused MFMA shared-MLP kernels (pn2_mlp_* / pn2_bn_*)
HAS_FUSED_MLP = True


# --------------------------------------------------------------------------- checks
def _fail(msg):
    raise RuntimeError(msg)


def _check_cuda(t, name):
    if not t.is_cuda:
        _fail("CPU not supported" if name is None else f"{name} must be a CUDA tensor")


def _check(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        _fail(f"{name} must be a torch.Tensor")
    if not t.is_contiguous():
        _fail(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = {torch.float32: "a float", torch.int32: "an int", torch.int64: "a long"}[dtype]
        _fail(f"{name} must be {kind} tensor")


def _f32(t, name):
    _check(t, name, torch.float32)


def _i32(t, name):
    _check(t, name, torch.int32)


def _i64(t, name):
    _check(t, name, torch.int64)


def _same_device(lead, *others):
    # reference: `if (a.is_cuda()) CHECK_CUDA(b)` then "CPU not supported"
    if not lead[0].is_cuda:
        _fail("CPU not supported")
    for t, name in others:
        if t is None:
            continue
        _check_cuda(t, name)
        if t.device != lead[0].device:
            _fail(f"{name} must be on {lead[0].device}")


def _ptr(t):
    return None if t is None else t.data_ptr()


class KernelTimer:
    """Optional per-entry-point HIP-event timing (bench.py): events are recorded on the
    stream the kernels are enqueued on, resolved lazily by `summary()` after a sync.
    `enabled` lets the caller sample a subset of steps (two event records per launch cost
    ~1 ms per 200-launch step); launches on a stream other than `main_stream` are reported
    under `name@side` (work prefetched off the critical path)."""

    def __init__(self, main_stream=None, pool=1536):
        self.records = []          # (name, start_event, end_event, algorithmic_bytes, algorithmic_flops)
        self.enabled = True
        self.main_stream = main_stream
        # fence-free timing events from the library (pn2_event_*), created up front: a torch.cuda.Event record flushes the
        # L2 (~20 us of queue time each, ~11 ms per sampled step of 250 launches)
        self._free = [_lib.pn2_event_create() for _ in range(pool)]
        self._all = list(self._free)

    def event(self):
        if not self._free:
            ev = _lib.pn2_event_create()
            self._all.append(ev)
            return ev
        return self._free.pop()

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        ms = ctypes.c_float()
        for name, s, e, nbytes, nflops in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "alg_bytes": 0, "alg_flops": 0})
            d["calls"] += 1
            if _lib.pn2_event_elapsed_ms(s, e, ctypes.byref(ms)) != 0:
                _fail("pn2_event_elapsed_ms failed")
            d["ms"] += float(ms.value)
            d["alg_bytes"] += nbytes
            d["alg_flops"] += nflops
        return out

    def __del__(self):
        try:
            for ev in self._all:
                _lib.pn2_event_destroy(ev)
        except Exception:
            pass


TIMER = None  # set to a KernelTimer() to profile
DETAIL_TAGS = os.environ.get("PN2_TIMER_DETAIL") == "1"   # per-shape rows for the MLP kernels


_FN = {}
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _call(name, ref, *args, alg_bytes=0, alg_flops=0, tag=None, label=None):
    """Enqueue `name` on the current stream of `ref`'s device (`label`: row name in the kernel timer, default `name`)."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib, name)
    dev = ref.device.index
    # fast path (a step is several hundred of these calls and the host thread is what a scan-sized step waits for): no
    # device context switch when the tensor's device is the current one, raw stream handle without a Stream object
    if _raw_stream is not None and (TIMER is None or not TIMER.enabled) and dev == torch.cuda.current_device():
        rc = fn(*args, _raw_stream(dev))
        if rc != 0:
            detail = _lib.pn2_strerror(rc).decode()
            _fail(f"{name} failed: {detail} (rc={rc}, hipError={_lib.pn2_last_hip_error()})")
        return
    with torch.cuda.device(ref.device):
        stream = torch.cuda.current_stream(ref.device).cuda_stream
        if TIMER is not None and TIMER.enabled:
            ev0, ev1 = TIMER.event(), TIMER.event()
            _lib.pn2_event_record(ev0, stream)
            rc = getattr(_lib, name)(*args, stream)
            _lib.pn2_event_record(ev1, stream)
            label = (label or name) if tag is None else f"{label or name}[{tag}]"
            if TIMER.main_stream is not None and stream != TIMER.main_stream:
                label += "@side"
            TIMER.records.append((label, ev0, ev1, int(alg_bytes), int(alg_flops)))
        else:
            rc = getattr(_lib, name)(*args, stream)
    if rc != 0:
        detail = _lib.pn2_strerror(rc).decode()
        _fail(f"{name} failed: {detail} (rc={rc}, hipError={_lib.pn2_last_hip_error()})")


#: let the library route sparse-ball queries (clouds >= 2048 points, estimated N r^3 <= 4 nsample) through the cell-list
#: kernels (identical results; tests flip it to compare both implementations, or force the cell list with "force")
BALL_QUERY_GRID = True
BQ_SCAN, BQ_CELLS, BQ_SLABS = 0, 1, 2                  # include/pn2_hip.h: PN2_BQ_*
PN2_FPS_FEW_CUS = 1
PN2_FPS_FEWEST_CUS = 2
_sched = threading.local()


@contextlib.contextmanager
def background_geometry(fewest=False):
    """FPS calls made inside run with PN2_FPS_FEW_CUS (include/pn2_hip.h): for geometry that is prefetched on a
    side stream while a training step runs on the main one.  `fewest`: PN2_FPS_FEWEST_CUS — trade sampling latency for
    CUs (the step next to it is longer than the sampling chain).  Results are identical."""
    prev = getattr(_sched, "few_cus", False)
    _sched.few_cus = PN2_FPS_FEW_CUS | (PN2_FPS_FEWEST_CUS if fewest else 0)
    try:
        yield
    finally:
        _sched.few_cus = prev


# ------------------------------------------------------------- the nine reference ops
#: pn2_furthest_point_sampling_ordered for clouds tagged as a sampling order (measurement switch; results never depend on it)
FPS_ORDERED = os.environ.get("PN2_FPS_ORDERED") != "0"
#: ... from this many samples on (below, the sampling rounds cost less than the verification launches: csrc/fps.hip)
FPS_ORDERED_MIN_SAMPLES = 256


def furthest_point_sampling(points, nsamples, ordered=False):
    """(B,N,3) f32 -> (B,nsamples) i32.  EXT/src/sampling.cpp:66-87.  `ordered`: the caller believes the clouds to be in
    farthest-point order already (the centres of the SA level above) — verified on the device, identical results either way
    (include/pn2_hip.h: pn2_furthest_point_sampling_ordered)."""
    _f32(points, "points")
    _same_device((points, "points"))
    B, N = points.size(0), points.size(1)
    nsamples = int(nsamples)
    out = torch.zeros(B, nsamples, dtype=torch.int32, device=points.device)
    flags = int(getattr(_sched, "few_cus", 0) or 0)
    if ordered and FPS_ORDERED and FPS_ORDERED_MIN_SAMPLES <= nsamples <= N and B > 0:
        ws_bytes = int(_lib.pn2_fps_ordered_workspace_bytes(B, N, nsamples))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=points.device)
        _call("pn2_furthest_point_sampling_ordered", points, B, N, nsamples, _ptr(points), _ptr(ws), ws_bytes, _ptr(out),
              flags, alg_bytes=B * (12 * N + 4 * nsamples), label="pn2_furthest_point_sampling")
        # (only shapes whose plan — WITH these flags — is a cluster mode write a status word; the verified-order shortcut
        # itself is a one-workgroup-per-cloud plan and has none)
        off = int(_lib.pn2_fps_status_offset_ex(B, N, nsamples, flags))
        if off >= 0:
            torch._assert_async(ws[off // 4:off // 4 + 1].view(torch.int32) == 0)
        return out
    ws_bytes = int(_lib.pn2_fps_workspace_bytes(B, N, nsamples))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=points.device) if ws_bytes else None
    if getattr(_sched, "few_cus", False):
        _call("pn2_furthest_point_sampling_ex", points, B, N, nsamples, _ptr(points), _ptr(ws), ws_bytes, _ptr(out),
              int(_sched.few_cus), alg_bytes=B * (12 * N + 4 * nsamples), label="pn2_furthest_point_sampling")
    else:
        _call("pn2_furthest_point_sampling", points, B, N, nsamples, _ptr(points), _ptr(ws), ws_bytes, _ptr(out),
              alg_bytes=B * (12 * N + 4 * nsamples))
    off = int(_lib.pn2_fps_status_offset_ex(B, N, nsamples, flags)) if ws is not None else -1
    if off >= 0:
        # a bounded inter-workgroup wait of the cluster kernels expired (CU-masked stream, another process on the GPU):
        # the remaining indices are zeros.  Checked on the device, asynchronously, in stream order before any consumer.
        torch._assert_async(ws[off // 4:off // 4 + 1].view(torch.int32) == 0)
    return out


_FPS_MODES = {None: -1, "resident": 0, "coop": 1, "stream": 2, "hybrid": 3, "bucketed": 4, "multi": 5}


@contextlib.contextmanager
def fps_bucketing(on):
    """Test / measurement hook (pn2_fps_set_bucketing): spatial bucketing of the cluster FPS kernels on / off."""
    prev = int(_lib.pn2_fps_get_bucketing())
    _lib.pn2_fps_set_bucketing(1 if on else 0)
    try:
        yield
    finally:
        _lib.pn2_fps_set_bucketing(prev)        # what it was (PN2_FPS_BUCKETING=0 processes stay off)


@contextlib.contextmanager
def fps_multi(on):
    """Measurement hook (pn2_fps_set_multi): several samples per cluster hand-off on / off.  Results never depend on it."""
    prev = int(_lib.pn2_fps_get_multi())
    _lib.pn2_fps_set_multi(1 if on else 0)
    try:
        yield
    finally:
        _lib.pn2_fps_set_multi(prev)


@contextlib.contextmanager
def fps_plan_override(mode=None, g=0, nc=0, coop_bs=0, bs=0):
    """Test hook (pn2_fps_set_plan_override): force an FPS kernel variant / cluster shape.  Results never depend on it."""
    rc = _lib.pn2_fps_set_plan_override(_FPS_MODES[mode], int(g), int(nc), int(coop_bs), int(bs))
    if rc != 0:
        _fail(f"pn2_fps_set_plan_override failed (rc={rc})")
    try:
        yield
    finally:
        _lib.pn2_fps_set_plan_override(-1, 0, 0, 0, 0)


def gather_points(points, idx):
    """(B,C,N), (B,m) i32 -> (B,C,m).  EXT/src/sampling.cpp:15-39."""
    _f32(points, "points"); _i32(idx, "idx")
    _same_device((points, "points"), (idx, "idx"))
    B, C, N = points.shape
    m = idx.size(1)
    out = torch.empty(B, C, m, dtype=torch.float32, device=points.device)
    _call("pn2_gather_points", points, B, C, N, m, _ptr(points), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * m + 8 * C * m))
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,m), (B,m) -> (B,C,n).  EXT/src/sampling.cpp:40-65."""
    _f32(grad_out, "grad_out"); _i32(idx, "idx")
    _same_device((grad_out, "grad_out"), (idx, "idx"))
    B, C, m = grad_out.shape
    # (stays a scatter: one reference per centre — 2048 x C atomics per cloud, 17 us at the micro shape against 250 for an index +
    # gather; the sampled indices of a cloud are distinct, so no two atomics meet and the result is reproducible anyway)
    out = torch.zeros(B, C, int(n), dtype=torch.float32, device=grad_out.device)
    _call("pn2_gather_points_grad", grad_out, B, C, int(n), m, _ptr(grad_out), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * m + 4 * C * m + 4 * C * int(n)))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,m,3), (B,N,3) -> (B,m,nsample) i32.  EXT/src/ball_query.cpp:8-32."""
    _f32(new_xyz, "new_xyz"); _f32(xyz, "xyz")
    _same_device((new_xyz, "new_xyz"), (xyz, "xyz"))
    B, m = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    nsample = int(nsample)
    idx = torch.empty(B, m, nsample, dtype=torch.int32, device=new_xyz.device)  # kernel writes every slot
    # BALL_QUERY_GRID: True = the library's choice per shape, False = scan, "force" / "cells" / "slabs" = that algorithm
    # (tests and measurements; a shape the algorithm does not cover runs the scan)
    if BALL_QUERY_GRID is True:
        algo = int(_lib.pn2_ball_query_auto(B, N, m, float(radius), nsample))
    else:
        algo = {False: BQ_SCAN, "force": BQ_CELLS, "cells": BQ_CELLS, "slabs": BQ_SLABS}[BALL_QUERY_GRID]
    ws_bytes = int(_lib.pn2_ball_query_algo_bytes(algo, B, N, m, float(radius), nsample))
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=new_xyz.device)      # scratch, no initialisation needed
        _call("pn2_ball_query_algo", new_xyz, algo, B, N, m, float(radius), nsample, _ptr(new_xyz), _ptr(xyz), _ptr(idx),
              _ptr(ws), ws_bytes, alg_bytes=B * (12 * N + 12 * m + 4 * m * nsample), label="pn2_ball_query")
    else:
        _call("pn2_ball_query", new_xyz, B, N, m, float(radius), nsample, _ptr(new_xyz), _ptr(xyz), _ptr(idx),
              alg_bytes=B * (12 * N + 12 * m + 4 * m * nsample))
    return idx


def ball_query_group_supported(B, N, m, radius, nsample, C, use_xyz=True) -> bool:
    """Shapes pn2_ball_query_group covers (nsample <= 256, 3 + C <= 16 floats per grouped row)."""
    return bool(_lib.pn2_ball_query_group_supported(int(B), int(N), int(m), float(radius), int(nsample), int(C),
                                                    int(bool(use_xyz))))


def ball_query_group_pays(B, N, m, radius, nsample) -> bool:
    """The fused query + grouping runs on the slab cell lists: taken where the library would pick them for the query alone
    (crowded balls in large clouds, sparse balls in clouds of >= 2048 points); elsewhere the scan + the grouping kernel
    are faster (tools/bq_bench.py)."""
    return int(_lib.pn2_ball_query_auto(int(B), int(N), int(m), float(radius), int(nsample))) == BQ_SLABS


def ball_query_group(new_xyz, xyz, feats_rows, radius, nsample, use_xyz=True, normalize=False, slab_w=0):
    """Ball query AND QueryAndGroup's gather / centre subtraction / concat in one pass (pn2_ball_query_group):
    new_xyz (B,m,3), xyz (B,N,3), feats_rows (B,N,C)|None -> (idx (B,m,nsample) i32, rows (B,m,nsample,Cx+C) f32);
    bit-identical to ball_query + group_concat_rows.  EXT/src/ball_query_gpu.cu:9-44, group_points_gpu.cu:8-28,
    OPS/pointnet2_utils.py:317-328."""
    _f32(new_xyz, "new_xyz"); _f32(xyz, "xyz")
    others = [(xyz, "xyz")]
    C = 0
    if feats_rows is not None:
        _f32(feats_rows, "features")
        C = feats_rows.size(2)
        others.append((feats_rows, "features"))
    _same_device((new_xyz, "new_xyz"), *others)
    B, m = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    nsample = int(nsample)
    Cx = 3 if use_xyz else 0
    if not ball_query_group_supported(B, N, m, radius, nsample, C, use_xyz):
        _fail(f"pn2_ball_query_group does not cover nsample {nsample}, row width {Cx + C} (radius {radius})")
    idx = torch.empty(B, m, nsample, dtype=torch.int32, device=new_xyz.device)
    rows = torch.empty(B, m, nsample, Cx + C, dtype=torch.float32, device=new_xyz.device)
    ws_bytes = int(_lib.pn2_ball_query_group_workspace_bytes(B, N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=new_xyz.device)          # scratch, no initialisation needed
    _call("pn2_ball_query_group", new_xyz, B, N, m, float(radius), nsample, C, int(bool(use_xyz)), int(bool(normalize)),
          _ptr(new_xyz), _ptr(xyz), _ptr(feats_rows), _ptr(idx), _ptr(rows), _ptr(ws), ws_bytes, int(slab_w),
          alg_bytes=B * (12 * N + 12 * m + 4 * C * N + 4 * (Cx + C) * m * nsample + 4 * m * nsample))
    return idx, rows


def ball_query_unique_resample(idx, seed, want_cnt=True):
    """idx (B,m,nsample) i32 ball-query rows, IN PLACE: padded tails refilled with uniform draws from each row's unique
    hits (GF3D sample_uniformly); -> unique_cnt (B,m) f32 or None."""
    _i32(idx, "idx")
    _same_device((idx, "idx"))
    B, m, ns = idx.shape
    cnt = torch.empty(B, m, dtype=torch.float32, device=idx.device) if want_cnt else None
    _call("pn2_ball_query_unique_resample", idx, B * m, ns, int(seed) & 0xFFFFFFFF, _ptr(idx), _ptr(cnt),
          alg_bytes=8 * B * m * ns + 4 * B * m)
    return cnt


def group_points(points, idx):
    """(B,C,N), (B,npoints,nsample) i32 -> (B,C,npoints,nsample).  EXT/src/group_points.cpp:12-37."""
    _f32(points, "points"); _i32(idx, "idx")
    _same_device((points, "points"), (idx, "idx"))
    B, C, N = points.shape
    npoints, nsample = idx.size(1), idx.size(2)
    out = torch.empty(B, C, npoints, nsample, dtype=torch.float32, device=points.device)
    _call("pn2_group_points", points, B, C, N, npoints, nsample, _ptr(points), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * npoints * nsample + 4 * C * N + 4 * C * npoints * nsample))
    return out


#: group_points_grad / group_rows_grad without a prefetched inverse index build one and gather (bit-reproducible, no atomics);
#: PN2_GROUP_GRAD_CSR=0 restores the reference's atomic scatter
GROUP_GRAD_CSR = os.environ.get("PN2_GROUP_GRAD_CSR", "1") != "0"


def group_points_grad(grad_out, idx, n):
    """(B,C,npoints,nsample) -> (B,C,n).  EXT/src/group_points.cpp:39-62."""
    _f32(grad_out, "grad_out"); _i32(idx, "idx")
    _same_device((grad_out, "grad_out"), (idx, "idx"))
    B, C, npoints, nsample = grad_out.shape
    if GROUP_GRAD_CSR and B * C * npoints * nsample > 0 and int(n) > 0:
        # deterministic form: every point sums its rows in row order through the inverse index (no atomics, no zero fill)
        ptr, refs = group_inverse_index(idx, int(n))
        out = torch.empty(B, C, int(n), dtype=torch.float32, device=grad_out.device)
        _call("pn2_group_points_grad_csr", grad_out, B, C, int(n), npoints, nsample, _ptr(grad_out), _ptr(ptr), _ptr(refs), _ptr(out),
              alg_bytes=B * (4 * npoints * nsample + 4 * C * npoints * nsample + 4 * C * int(n)), label="pn2_group_points_grad")
        return out
    out = torch.zeros(B, C, int(n), dtype=torch.float32, device=grad_out.device)
    _call("pn2_group_points_grad", grad_out, B, C, int(n), npoints, nsample,
          _ptr(grad_out), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * npoints * nsample + 4 * C * npoints * nsample + 4 * C * int(n)))
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32].  EXT/src/interpolate.cpp:14-40."""
    _f32(unknowns, "unknowns"); _f32(knows, "knows")
    _same_device((unknowns, "unknowns"), (knows, "knows"))
    B, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknowns.device)
    _call("pn2_three_nn", unknowns, B, n, m, _ptr(unknowns), _ptr(knows), _ptr(dist2), _ptr(idx),
          alg_bytes=B * (12 * n + 12 * m + 24 * n))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m), (B,n,3) i32, (B,n,3) -> (B,C,n).  EXT/src/interpolate.cpp:42-71."""
    _f32(points, "points"); _i32(idx, "idx"); _f32(weight, "weight")
    _same_device((points, "points"), (idx, "idx"), (weight, "weight"))
    B, C, m = points.shape
    n = idx.size(1)
    out = torch.empty(B, C, n, dtype=torch.float32, device=points.device)
    _call("pn2_three_interpolate", points, B, C, m, n, _ptr(points), _ptr(idx), _ptr(weight), _ptr(out),
          alg_bytes=B * (4 * C * m + 24 * n + 4 * C * n))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) -> (B,C,m).  EXT/src/interpolate.cpp:72-99."""
    _f32(grad_out, "grad_out"); _i32(idx, "idx"); _f32(weight, "weight")
    _same_device((grad_out, "grad_out"), (idx, "idx"), (weight, "weight"))
    B, C, n = grad_out.shape
    if GROUP_GRAD_CSR and B * C * n > 0 and int(m) > 0:
        ptr, refs = group_inverse_index(idx, int(m))               # idx (B, n, 3): slots (j, t) sorted by known point
        out = torch.empty(B, C, int(m), dtype=torch.float32, device=grad_out.device)
        _call("pn2_three_interpolate_grad_csr", grad_out, B, C, n, int(m), _ptr(grad_out), _ptr(weight), _ptr(ptr), _ptr(refs),
              _ptr(out), alg_bytes=B * (4 * C * n + 24 * n + 4 * C * int(m)), label="pn2_three_interpolate_grad")
        return out
    out = torch.zeros(B, C, int(m), dtype=torch.float32, device=grad_out.device)
    _call("pn2_three_interpolate_grad", grad_out, B, C, n, int(m),
          _ptr(grad_out), _ptr(idx), _ptr(weight), _ptr(out),
          alg_bytes=B * (4 * C * n + 24 * n + 4 * C * int(m)))
    return out


# -------------------------------------------------- point-major extras (fast path)
def group_concat_rows(xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius):
    """xyz (B,N,3), new_xyz (B,m,3), feats_rows (B,N,C)|None, idx (B,m,ns) ->
    (B,m,ns,Cx+C): relative (optionally radius-normalised) xyz ++ gathered features."""
    _i32(idx, "idx")
    B, m, ns = idx.shape
    N = xyz.size(1)
    C = 0
    others = [(idx, "idx")]
    if use_xyz:
        _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
        others += [(new_xyz, "new_xyz")]
    if feats_rows is not None:
        _f32(feats_rows, "features")
        C = feats_rows.size(2)
        others.append((feats_rows, "features"))
    _same_device((xyz, "xyz"), *others)
    W = (3 if use_xyz else 0) + C
    out = torch.empty(B, m, ns, W, dtype=torch.float32, device=xyz.device)
    _call("pn2_group_concat_rows", xyz, B, N, m, ns, C, int(bool(use_xyz)), int(bool(normalize)),
          float(radius if radius is not None else 1.0), _ptr(xyz), _ptr(new_xyz), _ptr(feats_rows),
          _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * m * ns + (12 * N + 12 * m if use_xyz else 0) + 4 * C * N + 4 * W * m * ns))
    return out


def _grad_rows(grad_out):
    """fp32 or bf16 gradient rows -> (entry point suffix, element bytes)."""
    if grad_out.dtype == torch.bfloat16:
        if not grad_out.is_contiguous():
            raise RuntimeError("grad_out must be contiguous")
        return "_bf16", 2
    _f32(grad_out, "grad_out")
    return "", 4


def group_rows_grad(grad_out, idx, n, c, col0, out=None):
    """grad_out (B,m,ns,W) fp32 or bf16 -> (B,n,c) fp32 gradient of the gathered feature columns.
    `out`: a ZERO-FILLED contiguous (B,n,c) fp32 tensor to accumulate into (e.g. a slice of a larger batch)."""
    sfx, eb = _grad_rows(grad_out)
    _i32(idx, "idx")
    _same_device((grad_out, "grad_out"), (idx, "idx"))
    B, m, ns, W = grad_out.shape
    if out is None:
        if (GROUP_GRAD_CSR and B * m * ns > 0 and int(n) > 0 and int(c) > 0
                and int(_lib.pn2_group_inverse_index_workspace_bytes(B, int(n), m, ns)) == 256):
            # no prefetched inverse index: build it here (ONE launch — the sort route of very large batches costs more than
            # the atomics it would save: 0.44 against 0.36 ms at 32 x 50000 points, C = 3) and gather — no atomics, no zero
            # fill, fixed summation order (0.60 -> 0.19 ms at the SA2 shape, C = 128)
            return group_rows_grad_csr(grad_out, group_inverse_index(idx, int(n)), n, c, col0)
        out = torch.zeros(B, int(n), int(c), dtype=torch.float32, device=grad_out.device)
    else:
        _f32(out, "out")
        if tuple(out.shape) != (B, int(n), int(c)) or out.device != grad_out.device:
            raise RuntimeError("group_rows_grad: out must be a contiguous float (B, n, c) tensor on the gradient's device")
    _call("pn2_group_rows_grad" + sfx, grad_out, B, int(n), m, ns, int(c), W, int(col0),
          _ptr(grad_out), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * m * ns + eb * int(c) * m * ns + 4 * int(c) * int(n)), label="pn2_group_rows_grad")
    return out


def group_inverse_index(idx, n):
    """idx (B,m,ns) i32 neighbourhood indices into clouds of n points -> (ptr (B*n+1) i32, refs (B*m*ns) i32): the rows
    (b*m + j)*ns + s that gathered each point, sorted by (point, row).  Data only (no features): build it next to the ball
    query, e.g. on the geometry prefetch stream."""
    _i32(idx, "idx")
    _same_device((idx, "idx"))
    B, m, ns = idx.shape
    n = int(n)
    ptr = torch.empty(B * n + 1, dtype=torch.int32, device=idx.device)
    refs = torch.empty(B * m * ns, dtype=torch.int32, device=idx.device)
    ws_bytes = int(_lib.pn2_group_inverse_index_workspace_bytes(B, n, m, ns))
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=idx.device)     # torch allocations are 512-byte aligned
    _call("pn2_group_inverse_index", idx, B, n, m, ns, _ptr(idx), _ptr(ptr), _ptr(refs), _ptr(ws), ws_bytes,
          alg_bytes=B * m * ns * 8 + B * n * 4)
    return ptr, refs


def attach_inverse_index(idx, n):
    """Deprecated spelling of group_inverse_index (round 2 hung the result on the idx tensor as a Python attribute, which
    neither record_stream nor a graph capture's clones can see): returns (ptr, refs) — carry it next to `idx`, e.g. as
    geometry["inv"]."""
    return group_inverse_index(idx, n)


def group_rows_grad_csr(grad_out, inv, n, c, col0, out=None):
    """grad_out (B,m,ns,W) + inv = (ptr, refs) of its index -> (B,n,c); every output row written, fixed summation order.
    `out`: contiguous (B,n,c) fp32 destination (overwritten)."""
    sfx, eb = _grad_rows(grad_out)
    ptr, refs = inv
    _i32(ptr, "ptr"); _i32(refs, "refs")
    _same_device((grad_out, "grad_out"), (ptr, "ptr"), (refs, "refs"))
    B, m, ns, W = grad_out.shape
    n, c = int(n), int(c)
    if ptr.numel() != B * n + 1 or refs.numel() != B * m * ns:
        raise RuntimeError("group_rows_grad_csr: inverse index does not belong to this gradient's neighbourhoods")
    if out is None:
        out = torch.empty(B, n, c, dtype=torch.float32, device=grad_out.device)
    else:
        _f32(out, "out")
        if tuple(out.shape) != (B, n, c) or out.device != grad_out.device:
            raise RuntimeError("group_rows_grad_csr: out must be a contiguous float (B, n, c) tensor on the gradient's device")
    _call("pn2_group_rows_grad_csr" + sfx, grad_out, B, n, c, W, int(col0), B * m * ns, _ptr(grad_out), _ptr(ptr), _ptr(refs),
          _ptr(out), alg_bytes=B * (8 * m * ns + eb * c * m * ns + 4 * c * n + 4 * n), label="pn2_group_rows_grad")
    return out


def group_lift_supported(n0) -> bool:
    """First-layer widths pn2_group_lift_rows covers (multiples of 4 from 16 to 256)."""
    return bool(_lib.pn2_group_lift_supported(int(n0)))


def group_lift_rows(P, xyz, new_xyz, idx, Wx, normalize, radius, stats=None, out_bf16=False, out=None):
    """The first conv of an SA stack applied BEFORE the grouping: P (B,N,N0) = per-point products f Wf^T, Wx (N0,3) the
    coordinate columns -> Y0 (B*m*ns, N0) = P[idx] + Wx rel, rel = (xyz[idx] - new_xyz) (/ radius); `stats` (2,N0) f64 +=
    column sums of y0 and y0^2.  Replaces group_points x 2 + cat + Conv2d 1x1 of the first layer
    (EXT/src/group_points_gpu.cu:8-28, OPS/pointnet2_utils.py:317-328, OPS/pointnet2_modules.py:9-19).
    `out_bf16`: Y0 in bf16 (mixed-precision stacks; the statistics are those of the rounded values)."""
    _f32(P, "P"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _i32(idx, "idx"); _f32(Wx, "Wx")
    _same_device((P, "P"), (xyz, "xyz"), (new_xyz, "new_xyz"), (idx, "idx"), (Wx, "Wx"), (stats, "stats"))
    B, m, ns = idx.shape
    N, N0 = xyz.size(1), P.size(-1)
    if P.numel() != B * N * N0 or tuple(Wx.shape) != (N0, 3) or tuple(new_xyz.shape) != (B, m, 3):
        _fail("group_lift_rows: P must be (B, N, N0), Wx (N0, 3), new_xyz (B, m, 3)")
    if out is not None:          # (a scan's rows of a batch's tensor: the per-scan calls of a segment-table stack)
        if out.numel() != B * m * ns * N0 or out.dtype != (torch.bfloat16 if out_bf16 else torch.float32) or not out.is_contiguous():
            _fail("group_lift_rows: out must be a contiguous (B m ns, N0) tensor of the output type")
        Y = out
    else:
        Y = torch.empty(B * m * ns, N0, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=P.device)
    _call("pn2_group_lift_rows_bf16" if out_bf16 else "pn2_group_lift_rows", P, B, N, m, ns, N0, int(bool(normalize)), float(radius if radius is not None else 1.0),
          _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(P), _ptr(Wx), _ptr(Y), _ptr(stats),
          alg_bytes=B * (4 * m * ns + 12 * N + 12 * m + 4 * N0 * N + (2 if out_bf16 else 4) * N0 * m * ns))
    return Y


def lift_split_weight(W):
    """W (N0, 3 + C) -> (Wx (N0,3), Wf (N0,C), WfT (C,N0)), one launch (pn2_lift_split_weight)."""
    _f32(W, "W")
    N0, K0 = W.shape
    C = K0 - 3
    buf = torch.empty(N0 * 3 + 2 * N0 * C, dtype=torch.float32, device=W.device)
    Wx, Wf, WfT = buf[:N0 * 3].view(N0, 3), buf[N0 * 3:N0 * 3 + N0 * C].view(N0, C), buf[N0 * 3 + N0 * C:].view(C, N0)
    _call("pn2_lift_split_weight", W, N0, C, _ptr(W), _ptr(Wx), _ptr(Wf), _ptr(WfT))
    return Wx, Wf, WfT


def mlp_lift_supported(K, N, ns) -> bool:
    """Shapes the layer above a lifted first layer can take with that layer's output re-formed on the fly
    (pn2_mlp_gemm_lift / pn2_mlp_wgrad_lift / pn2_mlp_dgrad_lift): K = lifted width, N = width above, ns rows per centre."""
    return N <= 128 and bool(_lib.pn2_mlp_lift_supported(int(K), int(N), int(ns)))


def lift_points(P, xyz, new_xyz, Wx, normalize, radius):
    """(Pq (B N, N0), Q (B m, N0)): the lifted first layer's per-point products with the coordinate term of the point folded in,
    and the per-centre term — y0[b, j, s] = Pq[b, idx[b, j, s]] - Q[b, j] (include/pn2_hip.h)."""
    _f32(P, "P"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(Wx, "Wx")
    _same_device((P, "P"), (xyz, "xyz"), (new_xyz, "new_xyz"), (Wx, "Wx"))
    B, N, _ = xyz.shape
    m, N0 = new_xyz.size(1), P.size(-1)
    if P.numel() != B * N * N0 or tuple(Wx.shape) != (N0, 3) or tuple(new_xyz.shape) != (B, m, 3):
        _fail("lift_points: P must be (B, N, N0), Wx (N0, 3), new_xyz (B, m, 3)")
    buf = torch.empty(B * (N + m), N0, dtype=torch.float32, device=P.device)
    Pq, Q = buf[:B * N], buf[B * N:]
    _call("pn2_lift_points", P, B, N, m, N0, int(bool(normalize)), float(radius if radius is not None else 1.0), _ptr(xyz),
          _ptr(new_xyz), _ptr(P), _ptr(Wx), _ptr(Pq), _ptr(Q), alg_bytes=4 * B * (2 * N * N0 + m * N0 + 3 * N + 3 * m))
    return Pq, Q


# ------------------------------------------------------------------ round 6: eval-mode SA level in one kernel (f32x3)
_X3_WS = {}


def _x3_workspace(device):
    """256 bytes per (device, stream): the pass counter of the persistent x3 kernels (launches of one stream run in order; two
    streams must not share it)."""
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _X3_WS.get(key)
    if ws is None:
        ws = _X3_WS[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return ws


def x3_weight_bytes(N, K):
    return int(_lib.pn2_x3_weight_bytes(int(N), int(K)))


def x3_pack_weight(W, perm, out=None):
    """W (N, K) fp32 (N % 32 == 0, K % 16 == 0) -> uint8 fragments of the split-bf16 product (include/pn2_hip.h).
    `out`: a uint8 view of x3_weight_bytes(N, K) bytes to write into (a slice of a level's weight stream)."""
    _f32(W, "W")
    _check_cuda(W, "W")
    N, K = W.shape
    nbytes = x3_weight_bytes(N, K)
    if nbytes == 0:
        _fail("x3_pack_weight: N must be a multiple of 32 and K a multiple of 16")
    if out is None:
        # (every unit is written by the kernel; the padding up to a whole ring slot is read by the weight stream's DMA but
        # never used: no fill launch per call — the training GEMMs pack their weight on every call)
        out = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
    elif out.numel() < nbytes or out.dtype != torch.uint8 or not out.is_contiguous():
        _fail("x3_pack_weight: `out` must be a contiguous uint8 tensor of x3_weight_bytes(N, K) bytes")
    _call("pn2_x3_pack_weight", W, N, K, K, int(bool(perm)), _ptr(W), _ptr(out))
    return out


def sa_eval_x3_supported(mode, ns, C, c1, c_mid, c_out):
    return bool(_lib.pn2_sa_eval_x3_supported(int(mode), int(ns), int(C), int(c1), int(c_mid), int(c_out)))


def sa_eval_x3(mode, xyz, new_xyz, idx, feats, Q, c1, w0_frags, c_mid, wstream, bias_mid, bias_fin, out, col0=0):
    """One scale of an eval-mode SA level in one kernel (pn2_sa_eval_x3): writes columns [col0, col0 + c_out) of the rows
    tensor `out` (B, m, width).  mode 0: feats (B, N, C) rows or None; mode 1: feats = Pq (B N, c1), Q (B m, c1)."""
    _f32(new_xyz, "new_xyz"); _i32(idx, "idx"); _f32(out, "out"); _f32(bias_fin, "bias_fin")
    _same_device((new_xyz, "new_xyz"), (idx, "idx"), (xyz, "xyz"), (feats, "feats"), (Q, "Q"), (out, "out"),
                 (wstream, "wstream"), (w0_frags, "w0_frags"), (bias_mid, "bias_mid"), (bias_fin, "bias_fin"))
    B, m, ns = idx.shape
    N = xyz.size(1)
    c_out = bias_fin.numel()
    ldo = out.size(-1)
    if out.numel() != B * m * ldo or col0 < 0 or col0 + c_out > ldo:
        _fail("sa_eval_x3: out must be (B, m, width) with col0 + c_out <= width")
    if mode == 0:
        C = 0 if feats is None else feats.size(2)
        if feats is not None:
            _f32(feats, "feats")
            if tuple(feats.shape[:2]) != (B, N):
                _fail("sa_eval_x3: feats must be (B, N, C) point-major rows")
        fptr = _ptr(feats) if feats is not None else _ptr(xyz)
    else:
        C = 0
        _f32(feats, "Pq"); _f32(Q, "Q")
        if feats.numel() != B * N * c1 or Q.numel() != B * m * c1:
            _fail("sa_eval_x3: Pq must be (B N, c1) and Q (B m, c1)")
        fptr = _ptr(feats)
    if not sa_eval_x3_supported(mode, ns, C, c1, c_mid, c_out):
        _fail(f"sa_eval_x3: unsupported shape (mode {mode}, ns {ns}, C {C}, widths {c1}/{c_mid}/{c_out})")
    rows = B * m * ns
    flops = 2 * rows * ((16 * c1 if mode == 0 else 0) + c1 * c_mid + (c_mid or c1) * c_out)
    nbytes = B * (4 * m * ns + 12 * m + 4 * c_out * m) + (B * (12 * N + 4 * C * N) if mode == 0 else 4 * c1 * B * (N + m))
    _call("pn2_sa_eval_x3", idx, int(mode), B, N, m, ns, C, _ptr(xyz), _ptr(new_xyz), _ptr(idx), fptr, _ptr(Q), int(c1),
          _ptr(w0_frags), int(c_mid), _ptr(wstream), _ptr(bias_mid), int(c_out), _ptr(bias_fin),
          out.data_ptr() + 4 * int(col0), int(ldo), _ptr(_x3_workspace(idx.device)), alg_bytes=nbytes, alg_flops=flops,
          tag=(f"ns{ns},{c1}/{c_mid}/{c_out}" if DETAIL_TAGS else None))
    return out


#: route the shared-MLP GEMMs of the fp32 node through the split-bf16 product where it covers the shape (csrc/x3_chain.hip):
#: opt-in arithmetic "f32x3" (fused_mlp.set_x3 / bench.py --dtype f32x3 / PN2_X3=1); the exact fp32 MFMA kernels are the default
X3_GEMM = os.environ.get("PN2_X3") == "1"
#: ... from this many rows on (below, the exact kernels' launch is as fast and saves the weight packing)
X3_MIN_ROWS = int(os.environ.get("PN2_X3_MIN_ROWS", "16384"))
#: ... the input-gradient form (pro 2, epi 2) too.  Off by default: at the shapes it covers (K = N = 128) that GEMM moves four
#: (M, 128) tensors and is HBM-bound in both arithmetics — measured 0.442 ms exact vs 0.480 ms f32x3 at M = 1M
#: (profiles/r06_x3_gemm.jsonl); PN2_X3_DGRAD=1 routes it as well (the whole-suite run with the route forced does).
X3_DGRAD = os.environ.get("PN2_X3_DGRAD") == "1"


#: the backward of the layer above a re-formed first layer (pn2_x3_bwd_fold_first) follows X3_GEMM; PN2_X3_BWD_FIRST=0: exact
X3_BWD_FIRST = os.environ.get("PN2_X3_BWD_FIRST", "1") != "0"
#: ... and so does the pooled last layer's backward at K = 64 (pn2_x3_pool_bwd); PN2_X3_POOL_BWD=0: exact
X3_POOL_BWD = os.environ.get("PN2_X3_POOL_BWD", "1") != "0"


def x3_gemm_supported(K, N, pro, epi, ns=0):
    return bool(_lib.pn2_x3_gemm_supported(int(K), int(N), int(pro), int(epi), int(ns)))


def x3_gemm(X, W, pro, epi, X2=None, p=None, stats=None, Yprev=None, e_fin=None, sgn=None, ns=0, M=None):
    """pn2_x3_gemm: Y (M, N) [epi 1, 2] or (pmax, parg) [epi 3] of pro(X) @ W^T on the split-bf16 product; arguments as
    mlp_gemm / mlp_gemm_pool."""
    _f32(W, "W"); _f32(X, "X")
    N, K = W.shape
    M = int(M if M is not None else X.size(0))
    frags = x3_pack_weight(W.contiguous(), perm=False)
    p0 = p1 = p2 = None
    if p is not None:
        p0, p1 = p[0], p[1]
        p2 = p[2] if len(p) > 2 else None
    Y = pmax = parg = None
    if epi == 3:
        psz = min(int(ns), 32)
        pmax = torch.empty(M // psz, N, dtype=torch.float32, device=X.device)
        parg = torch.empty(M // psz, N, dtype=torch.int32, device=X.device)
    else:
        Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
    nbytes = 4 * (M * K * (2 if pro == PRO_GY else 1) + N * K) + (8 * (M // min(int(ns) or 32, 32)) * N if epi == 3 else
                                                                   4 * M * N * (2 if epi == 2 else 1))
    _call("pn2_x3_gemm", X, M, K, N, int(pro), int(epi), _ptr(X), _ptr(X2), _ptr(p0), _ptr(p1), _ptr(p2), _ptr(frags), _ptr(Y),
          _ptr(stats), _ptr(Yprev), _ptr(e_fin), _ptr(pmax), _ptr(parg), _ptr(sgn), int(ns), _ptr(_x3_workspace(X.device)),
          alg_bytes=nbytes,
          alg_flops=2 * M * N * K, tag=(f"M{M},K{K},N{N},pro{int(pro)},epi{int(epi)}" if DETAIL_TAGS else None))
    return (pmax, parg) if epi == 3 else Y


def group_lift_stats(Pq, Q, idx, N, stats):
    """stats (2, N0) f64 += column sums of y0 = Pq[idx] - Q and y0^2; returns gidx (B m ns) int32 = b N + idx, the row of Pq
    every grouped row reads.  Nothing of y0 is stored."""
    _f32(Pq, "Pq"); _f32(Q, "Q"); _i32(idx, "idx")
    _same_device((Pq, "Pq"), (Q, "Q"), (idx, "idx"), (stats, "stats"))
    B, m, ns = idx.shape
    N0 = Pq.size(-1)
    if Pq.numel() != B * int(N) * N0 or Q.numel() != B * m * N0 or stats is None:
        _fail("group_lift_stats: Pq must be (B N, N0), Q (B m, N0), stats (2, N0) float64")
    gidx = torch.empty(B * m * ns, dtype=torch.int32, device=Pq.device)
    _call("pn2_group_lift_stats", Pq, B, int(N), m, ns, N0, _ptr(idx), _ptr(Pq), _ptr(Q), _ptr(gidx), _ptr(stats),
          alg_bytes=B * (8 * m * ns + 4 * N0 * (int(N) + m)), label="pn2_group_lift_rows")
    return gidx


def mlp_gemm_lift(Pq, gidx, Q, ns, fin0, W, stats):
    """Y (M, N) = relu(bn_0(Pq[gidx] - Q[row // ns])) W^T with the column sums of Y, Y^2 in `stats` (csrc/mlp_gemm.hip PRO_LIFT)."""
    _f32(W, "W"); _f32(fin0, "fin0")
    N, K = W.shape
    M, lrows = gidx.numel(), Pq.size(0)
    if Pq.size(1) != K or tuple(fin0.shape) != (4, K) or Q.numel() != (M // ns) * K:
        _fail("mlp_gemm_lift: Pq (rows, K), Q (M / ns, K), fin0 (4, K), W (N, K) expected")
    Y = torch.empty(M, N, dtype=torch.float32, device=W.device)
    _call("pn2_mlp_gemm_lift", W, M, K, N, lrows, _ptr(Pq), _ptr(gidx), _ptr(Q), int(ns), _ptr(fin0), _ptr(W), _ptr(Y),
          _ptr(stats), alg_bytes=4 * (lrows * K + M + (M // ns) * K + M * N + N * K), alg_flops=2 * M * N * K,
          tag=(f"M{M},K{K},N{N},lift" if DETAIL_TAGS else None), label="pn2_mlp_gemm")
    return Y


def mlp_wgrad_lift(Yl, consts, G, Pq, gidx, Q, ns, a_fin, dW=None):
    """dW (N, K) += (c1 G + c2 Yl + c3)^T relu(bn_0(Pq[gidx] - Q[row // ns])) (pn2_mlp_wgrad with the activation re-formed)."""
    M, N = Yl.shape
    K, lrows = Pq.size(1), Pq.size(0)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_wgrad_lift", Yl, M, N, K, lrows, _ptr(G), _ptr(Yl), _ptr(consts), _ptr(Pq), _ptr(gidx), _ptr(Q), int(ns),
          _ptr(a_fin), _ptr(dW), alg_bytes=4 * (2 * M * N + lrows * K + M + (M // ns) * K + N * K), alg_flops=2 * M * N * K,
          tag=(f"M{M},N{N},K{K},lift" if DETAIL_TAGS else None), label="pn2_mlp_wgrad")
    return dW


def mlp_dgrad_lift(G, Yl, consts, Wt, sums, Pq, gidx, Q, ns, e_fin):
    """Gout (M, K) = [(c1 G + c2 Yl + c3) Wt^T] masked by bn_0(y0) > 0, `sums` (2, K) += column sums of Gout, Gout yhat_0;
    y0 = Pq[gidx] - Q[row // ns] re-formed in the epilogue (csrc/mlp_gemm.hip EPI_MASKL).  Wt (K, N)."""
    M, N = Yl.shape
    K, lrows = Pq.size(1), Pq.size(0)
    if tuple(Wt.shape) != (K, N) or tuple(e_fin.shape) != (4, K):
        _fail("mlp_dgrad_lift: Wt (K, N), e_fin (4, K) expected")
    Gout = torch.empty(M, K, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_dgrad_lift", Yl, M, K, N, lrows, _ptr(G), _ptr(Yl), _ptr(consts), _ptr(Wt), _ptr(Gout), _ptr(sums), _ptr(Pq),
          _ptr(gidx), _ptr(Q), int(ns), _ptr(e_fin), alg_bytes=4 * (2 * M * N + M * K + lrows * K + M + (M // ns) * K + N * K),
          alg_flops=2 * M * N * K, tag=(f"M{M},K{N},N{K},pro2,maskl" if DETAIL_TAGS else None), label="pn2_mlp_gemm")
    return Gout


def lift_dw_assemble(acc, Wx, c2, dWf):
    """dW (N0, 3 + C) = [acc[:3 N0] + diag(c2) Wx RR | dWf], one launch (pn2_lift_dw_assemble)."""
    N0, C = dWf.shape
    dW = torch.empty(N0, C + 3, dtype=torch.float32, device=dWf.device)
    _call("pn2_lift_dw_assemble", dWf, N0, C, _ptr(acc), _ptr(Wx), _ptr(c2), _ptr(dWf), _ptr(dW))
    return dW


def group_lift_rows_grad(G, P, Wx, consts, xyz, new_xyz, inv, ns, normalize, radius, acc):
    """Backward of group_lift_rows behind a BatchNorm (include/pn2_hip.h): G (M,N0) masked gradient of the layer above, P
    (B,N,N0) / Wx (N0,3) of the forward, consts (3,N0), inv = (ptr, refs) -> S (B,N,N0) = sum over the rows that gathered each
    point of dL/dy0 = c1 g + c2 y0 + c3;  `acc` (3*N0 + 9) f32 <- [dWx (N0,3) without its c2 Wx RR term | RR (3,3)]."""
    bf = G.dtype == torch.bfloat16           # (mixed-precision stacks: the layer above leaves its masked gradient in bf16)
    if bf:
        if not G.is_contiguous() or not G.is_cuda:
            _fail("group_lift_rows_grad: G must be a contiguous device tensor")
    else:
        _f32(G, "G")
    _f32(P, "P"); _f32(Wx, "Wx"); _f32(consts, "consts"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(acc, "acc")
    ptr, refs = inv
    _i32(ptr, "ptr"); _i32(refs, "refs")
    _same_device((G, "G"), (P, "P"), (Wx, "Wx"), (consts, "consts"), (xyz, "xyz"), (new_xyz, "new_xyz"), (ptr, "ptr"),
                 (refs, "refs"), (acc, "acc"))
    B, N = xyz.size(0), xyz.size(1)
    m = new_xyz.size(1)
    M, N0 = G.shape
    ns = int(ns)
    if (P.numel() != B * N * N0 or M != B * m * ns or ptr.numel() != B * N + 1 or refs.numel() != M
            or tuple(consts.shape) != (3, N0) or tuple(Wx.shape) != (N0, 3) or acc.numel() != 3 * N0 + 9):
        _fail("group_lift_rows_grad: shapes do not belong to one SA level")
    S = torch.empty(B, N, N0, dtype=torch.float32, device=G.device)
    ws_bytes = int(_lib.pn2_group_lift_rows_grad_workspace_bytes(B, N, m, ns, N0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
    _call("pn2_group_lift_rows_grad_bf16" if bf else "pn2_group_lift_rows_grad", G, B, N, m, ns, N0, int(bool(normalize)), float(radius if radius is not None else 1.0),
          _ptr(xyz), _ptr(new_xyz), _ptr(G), _ptr(P), _ptr(Wx), _ptr(consts), _ptr(ptr), _ptr(refs), _ptr(S),
          _ptr(acc), _ptr(ws), ws_bytes,
          alg_bytes=8 * M + (2 if bf else 4) * M * N0 + 4 * B * N * (2 * N0 + 4) + 12 * B * m)
    return S


def group_lift_rows_seg(P, xyz, new_xyz, idx, Wx, normalize, radius, stats, seg, out_bf16=True):
    """group_lift_rows for the S scans of a segment table in ONE launch (grid.y = scan): scan s = the clouds of rows
    [seg.ptr[s], seg.ptr[s+1]), its column sums in stats[s] (S, 2, N0) — bit for bit the sums of S single-scan launches
    (include/pn2_hip.h)."""
    B, m, ns = idx.shape
    N, N0 = xyz.size(1), P.size(-1)
    Y = torch.empty(B * m * ns, N0, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=P.device)
    _call("pn2_group_lift_rows_seg", P, B, N, m, ns, N0, int(bool(normalize)), float(radius if radius is not None else 1.0),
          _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(P), _ptr(Wx), _ptr(Y), int(bool(out_bf16)), _ptr(stats), _ptr(seg.ptr),
          seg.nseg, seg.max_rows // (m * ns),
          alg_bytes=B * (4 * m * ns + 12 * N + 12 * m + 4 * N0 * N + (2 if out_bf16 else 4) * N0 * m * ns),
          label="pn2_group_lift_rows_bf16" if out_bf16 else "pn2_group_lift_rows")
    return Y


def group_lift_rows_grad_seg(G, P, Wx, consts, xyz, new_xyz, inv, ns, normalize, radius, acc, seg):
    """group_lift_rows_grad for the S scans of a segment table in ONE launch sequence: consts (S, 3, N0), acc (S, 3 N0 + 9)
    -> S (B, N, N0).  Every scan's sums are those of its own single-scan call."""
    bf = G.dtype == torch.bfloat16
    ptr, refs = inv
    B, N = xyz.size(0), xyz.size(1)
    m = new_xyz.size(1)
    N0 = G.size(1)
    S_out = torch.empty(B, N, N0, dtype=torch.float32, device=G.device)
    max_clouds = seg.max_rows // (m * int(ns))
    ws_bytes = int(_lib.pn2_group_lift_rows_grad_seg_workspace_bytes(seg.nseg, max_clouds, N, m, int(ns), N0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
    _call("pn2_group_lift_rows_grad_seg", G, B, N, m, int(ns), N0, int(bool(normalize)),
          float(radius if radius is not None else 1.0), _ptr(xyz), _ptr(new_xyz), _ptr(G), int(bf), _ptr(P), _ptr(Wx), _ptr(consts),
          _ptr(ptr), _ptr(refs), _ptr(S_out), _ptr(acc), _ptr(seg.ptr), seg.nseg, max_clouds, _ptr(ws), ws_bytes,
          alg_bytes=(2 if bf else 4) * G.size(0) * N0 + 4 * B * N * (2 * N0 + 4),
          label="pn2_group_lift_rows_grad_bf16" if bf else "pn2_group_lift_rows_grad")
    return S_out


def rows_max(x):
    """x (R,ns,C) -> (max (R,C), argmax (R,C) i32) over the ns axis."""
    _f32(x, "x")
    _same_device((x, "x"))
    R, ns, C = x.shape
    out = torch.empty(R, C, dtype=torch.float32, device=x.device)
    arg = torch.empty(R, C, dtype=torch.int32, device=x.device)
    _call("pn2_rows_max", x, R, ns, C, _ptr(x), _ptr(out), _ptr(arg), alg_bytes=4 * R * ns * C + 8 * R * C)
    return out, arg


def rows_max_grad(grad_out, arg, ns):
    _f32(grad_out, "grad_out"); _i32(arg, "arg")
    _same_device((grad_out, "grad_out"), (arg, "arg"))
    R, C = grad_out.shape
    gx = torch.empty(R, int(ns), C, dtype=torch.float32, device=grad_out.device)
    _call("pn2_rows_max_grad", grad_out, R, int(ns), C, _ptr(grad_out), _ptr(arg), _ptr(gx),
          alg_bytes=4 * R * int(ns) * C + 8 * R * C)
    return gx


def three_interpolate_rows(feats_rows, idx, weight, out=None, col0=0):
    """feats_rows (B,m,C), idx/weight (B,n,3) -> out (B,n,ldo)[..., col0:col0+C]."""
    _f32(feats_rows, "features"); _i32(idx, "idx"); _f32(weight, "weight")
    _same_device((feats_rows, "features"), (idx, "idx"), (weight, "weight"))
    B, m, C = feats_rows.shape
    n = idx.size(1)
    if out is None:
        out = torch.empty(B, n, C, dtype=torch.float32, device=feats_rows.device)
        col0 = 0
    else:
        _f32(out, "out")
    _call("pn2_three_interpolate_rows", feats_rows, B, C, m, n, out.size(2), int(col0),
          _ptr(feats_rows), _ptr(idx), _ptr(weight), _ptr(out), alg_bytes=B * (4 * C * m + 24 * n + 4 * C * n))
    return out


def three_interpolate_rows_grad(grad_out, idx, weight, m, c, col0=0, inv=None):
    """grad_out (B,n,ldg)[..., col0:col0+c] -> (B,m,c).  `inv` = group_inverse_index(idx, m): the atomic-free gather form
    (every output row written, fixed summation order); without it three atomicAdds per element like the reference."""
    _f32(grad_out, "grad_out"); _i32(idx, "idx"); _f32(weight, "weight")
    _same_device((grad_out, "grad_out"), (idx, "idx"), (weight, "weight"))
    B, n, ldg = grad_out.shape
    if inv is not None and int(c) % 4 == 0:
        ptr, refs = inv
        if ptr.numel() != B * int(m) + 1 or refs.numel() != idx.numel():
            _fail("three_interpolate_rows_grad: inverse index does not belong to this idx")
        out = torch.empty(B, int(m), int(c), dtype=torch.float32, device=grad_out.device)
        _call("pn2_three_interpolate_rows_grad_csr", grad_out, B, int(c), int(m), n, ldg, int(col0), _ptr(grad_out), _ptr(weight),
              _ptr(ptr), _ptr(refs), _ptr(out), alg_bytes=B * (4 * int(c) * n + 24 * n + 4 * int(c) * int(m)),
              label="pn2_three_interpolate_rows_grad")
        return out
    out = torch.zeros(B, int(m), int(c), dtype=torch.float32, device=grad_out.device)
    _call("pn2_three_interpolate_rows_grad", grad_out, B, int(c), int(m), n, ldg, int(col0),
          _ptr(grad_out), _ptr(idx), _ptr(weight), _ptr(out),
          alg_bytes=B * (4 * int(c) * n + 24 * n + 4 * int(c) * int(m)))
    return out


# ------------------------------------------------------------ TripletGCN primitives
def _check_index_range(index, n):
    if index.numel() and (int(index.min()) < 0 or int(index.max()) >= n):
        _fail("index out of range")


def gather_rows(x, index, out=None, col0=0, check=True):
    """x (N,H), index (E) i64 -> out (E,ldo)[:, col0:col0+H] (PyG __lift__)."""
    _f32(x, "x"); _i64(index, "index")
    _same_device((x, "x"), (index, "index"))
    N, H = x.shape
    E = index.numel()
    if check:
        _check_index_range(index, N)
    if out is None:
        out = torch.empty(E, H, dtype=torch.float32, device=x.device)
        col0 = 0
    else:
        _f32(out, "out")
    _call("pn2_gather_rows", x, E, H, N, out.size(1), int(col0), _ptr(x), _ptr(index), _ptr(out),
          alg_bytes=8 * E + 8 * E * H)
    return out


def scatter_add_rows(src, index, dim_size, h=None, col0=0, check=True):
    """src (E,lds)[:, col0:col0+h], index (E) i64 -> (dim_size,h); fp32 atomics."""
    _f32(src, "src"); _i64(index, "index")
    _same_device((src, "src"), (index, "index"))
    E, lds = src.shape
    h = lds if h is None else int(h)
    if check:
        _check_index_range(index, int(dim_size))
    out = torch.zeros(int(dim_size), h, dtype=torch.float32, device=src.device)
    _call("pn2_scatter_add_rows", src, E, h, int(dim_size), lds, int(col0), _ptr(src), _ptr(index), _ptr(out),
          alg_bytes=4 * E * h + 8 * E + 4 * int(dim_size) * h)
    return out


def segment_sum_rows(src, order, rowptr, dim_size, h=None, col0=0):
    """Deterministic scatter-add given a stable arg-sort `order` of the targets and
    its CSR `rowptr` (dim_size+1)."""
    _f32(src, "src"); _i64(order, "order"); _i64(rowptr, "rowptr")
    _same_device((src, "src"), (order, "order"), (rowptr, "rowptr"))
    E, lds = src.shape
    h = lds if h is None else int(h)
    out = torch.empty(int(dim_size), h, dtype=torch.float32, device=src.device)
    _call("pn2_segment_sum_rows", src, E, h, int(dim_size), lds, int(col0),
          _ptr(src), _ptr(order), _ptr(rowptr), _ptr(out), alg_bytes=4 * E * h + 16 * E + 4 * int(dim_size) * h)
    return out


def gather2_add_rows(q, p, ia, ib, cola, colb):
    """q (E,H) += p[ia, cola:cola+H] + p[ib, colb:colb+H]  (in place; returns q)."""
    _f32(q, "q"); _f32(p, "p"); _i64(ia, "ia"); _i64(ib, "ib")
    _same_device((q, "q"), (p, "p"), (ia, "ia"), (ib, "ib"))
    E, H = q.shape
    _call("pn2_gather2_add_rows", q, E, H, p.size(0), p.size(1), int(cola), int(colb), _ptr(p), _ptr(ia), _ptr(ib), _ptr(q),
          alg_bytes=16 * E + 16 * E * H)
    return q


def segment_sum2_rows(src, order, rowptr, dim_size, h, col0, col1):
    """out (dim_size,h) = CSR segment sum of src[:, col0:col0+h] + src[:, col1:col1+h] (deterministic, edge order)."""
    _f32(src, "src"); _i64(order, "order"); _i64(rowptr, "rowptr")
    _same_device((src, "src"), (order, "order"), (rowptr, "rowptr"))
    E, lds = src.shape
    out = torch.empty(int(dim_size), int(h), dtype=torch.float32, device=src.device)
    _call("pn2_segment_sum2_rows", src, E, int(h), int(dim_size), lds, int(col0), int(col1), _ptr(src), _ptr(order),
          _ptr(rowptr), _ptr(out), alg_bytes=8 * E * int(h) + 16 * E + 4 * int(dim_size) * int(h))
    return out


# ---- fused TripletGCN blocks (csrc/gcn_fused.hip): thin bindings — the autograd node that calls them
# (network_TripletGCN._FusedTripletLayer) validates dtypes / contiguity once per layer, not once per launch
def gcn_fused_supported(dn, de, dh, max_rows_per_scan):
    return bool(_lib.pn2_gcn_fused_supported(int(dn), int(de), int(dh), int(max_rows_per_scan)))


def gcn_linear(W, bias, ptr, S, A=None, triplet=None, bn=None, relu=False):
    """[ReLU] [BN per scan] (A W^T + bias) in one launch.  A (R, K) rows, or triplet = (x, e, dst, src): the virtual
    cat[x[dst], e, x[src]] (network_TripletGCN.py:46).  bn = (gamma, beta, eps) -> (out, ypre, mean, rstd), else out."""
    N, K = W.shape
    if triplet is not None:
        x, e, dst, src = triplet
        R, dn, de = e.size(0), x.size(1), e.size(1)
        ref, a_ptr, lda = e, None, 0
        tp = (_ptr(x), _ptr(e), _ptr(dst), _ptr(src), dn, de)
    else:
        R, lda = A.size(0), A.size(1)
        ref, a_ptr = A, _ptr(A)
        tp = (None, None, None, None, 0, 0)
    out = torch.empty(R, N, dtype=torch.float32, device=ref.device)
    if bn is not None:
        gamma, beta, eps = bn
        ypre = torch.empty(R, N, dtype=torch.float32, device=ref.device)
        stat = torch.empty(2, S, N, dtype=torch.float32, device=ref.device)
        _call("pn2_gcn_linear", ref, R, S, K, N, a_ptr, lda, *tp, _ptr(W), _ptr(bias), _ptr(ptr), _ptr(gamma), _ptr(beta),
              float(eps), int(bool(relu)), _ptr(ypre), _ptr(out), _ptr(stat[0]), _ptr(stat[1]), alg_flops=2 * R * K * N)
        return out, ypre, stat[0], stat[1]
    _call("pn2_gcn_linear", ref, R, S, K, N, a_ptr, lda, *tp, _ptr(W), _ptr(bias), _ptr(ptr), None, None, 0.0,
          int(bool(relu)), None, _ptr(out), None, None, alg_flops=2 * R * K * N)
    return out


def gcn_linear_grad_w(W_shape, ptr, S, dW, dbias, G=None, adjoint=None, bn=None, relu=False, ypre=None, A=None, triplet=None,
                      dgamma=None, dbeta=None):
    """Backward of a gcn_linear block up to the weights (include/pn2_hip.h): returns gz (R, N); dW / dbias / dgamma / dbeta
    are accumulated (zeroed by the caller).  adjoint = (gagg, gedge, dst, dh, de): the gradient is the adjoint of split +
    aggregate read in place.  bn = (ypre, mean, rstd, gamma, beta)."""
    N, K = W_shape
    if triplet is not None:
        x, e, dst, src = triplet
        R, dn, de = e.size(0), x.size(1), e.size(1)
        ref, a_ptr, lda = e, None, 0
        tp = [_ptr(x), _ptr(e), _ptr(dst), _ptr(src), dn, de]
    else:
        R, lda = A.size(0), A.size(1)
        ref, a_ptr = A, _ptr(A)
        tp = [None, None, None, None, 0, 0]
    if adjoint is not None:
        gagg, gedge, adst, dh, dE = adjoint
        gp = (None, _ptr(gagg), _ptr(gedge), int(dh), int(dE))
        if tp[2] is None:
            tp[2] = _ptr(adst)
    else:
        gp = (_ptr(G), None, None, 0, 0)
    gz = torch.empty(R, N, dtype=torch.float32, device=ref.device)
    if bn is not None:
        yp, mean, rstd, gamma, beta = bn
        bp = (_ptr(yp), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta))
    else:
        bp = (_ptr(ypre), None, None, None, None)
    _call("pn2_gcn_linear_grad_w", ref, R, S, K, N, *gp, *bp, int(bool(relu)), _ptr(ptr), a_ptr, lda, *tp, _ptr(gz), _ptr(dW),
          _ptr(dbias), _ptr(dgamma), _ptr(dbeta), alg_flops=2 * R * K * N)
    return gz


def gcn_linear_grad_x(gz, W, ptr, S, scatter=None):
    """gz W: (R, K) rows, or scatter = (gx, ge, dst, src, dn, de): gx (nodes, dn) += the x[dst] / x[src] column blocks
    (zero on entry), ge (R, de) = the middle block (adjoint of the triplet gather)."""
    R, N = gz.shape
    K = W.size(1)
    if scatter is not None:
        gx, ge, dst, src, dn, de = scatter
        _call("pn2_gcn_linear_grad_x", gz, R, S, K, N, _ptr(gz), _ptr(W), _ptr(ptr), None, _ptr(gx), _ptr(ge), _ptr(dst),
              _ptr(src), int(dn), int(de), alg_flops=2 * R * K * N)
        return None
    gin = torch.empty(R, K, dtype=torch.float32, device=gz.device)
    _call("pn2_gcn_linear_grad_x", gz, R, S, K, N, _ptr(gz), _ptr(W), _ptr(ptr), _ptr(gin), None, None, None, None, 0, 0,
          alg_flops=2 * R * K * N)
    return gin


def gcn_edge_slice(h, off, de, relu):
    """[ReLU] h[:, off:off+de] as a contiguous tensor (the new edge feature of a TripletGCN layer), one launch."""
    R, ld = h.shape
    out = torch.empty(R, int(de), dtype=torch.float32, device=h.device)
    _call("pn2_gcn_edge_slice", h, R, ld, int(off), int(de), int(bool(relu)), _ptr(h), _ptr(out), alg_bytes=8 * R * int(de))
    return out


_GCN_PARAMS = ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3", "g3", "be3", "W4", "b4")
_GCN_SAVED = ("h1", "h1p", "m1", "r1", "h2", "h2p", "m2", "r2", "agg", "e_out", "t", "tp", "m3", "r3", "out")


class GcnLayer(ctypes.Structure):
    """`pn2_gcn_layer` of include/pn2_hip.h, field for field."""
    _fields_ = ([("nodes", ctypes.c_longlong), ("edges", ctypes.c_longlong)]
                + [(k, _c_int) for k in ("S", "dn", "de", "dh", "relu_out")]
                + [(k, _c_f32) for k in ("eps1", "eps2", "eps3")]
                + [(k, _c_vp) for k in ("x", "e", "dst", "src", "order", "rowptr", "node_ptr", "edge_ptr")]
                + [(k, _c_vp) for k in _GCN_PARAMS] + [(k, _c_vp) for k in _GCN_SAVED]
                + [(k, _c_vp) for k in ("g_out", "g_e")] + [(k, _c_vp) for k in ("d" + k for k in _GCN_PARAMS)]
                + [(k, _c_vp) for k in ("gx", "ge", "work")])


def _gcn_saved_layout(nodes, edges, S, dn, de, dh):
    """Float offsets of the forward's saved pieces (all but e_out / out, which are tensors of their own) in ONE allocation."""
    wide = 2 * dh + de
    sizes = (edges * dh, edges * dh, S * dh, S * dh, edges * wide, edges * wide, S * wide, S * wide, nodes * dh, 0,
             nodes * dh, nodes * dh, S * dh, S * dh, 0)
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    return offs, total


def gcn_layer_forward(x, e, dst, src, order, rowptr, node_ptr, edge_ptr, S, relu_out, eps, params):
    """One TripletGCN layer (network_TripletGCN.py:40-58) in ONE C call (pn2_gcn_layer_forward: six launches).
    -> (out (nodes, dn), e_out (edges, de), saved): `saved` holds the pre-BatchNorm values and the per-scan statistics
    for gcn_layer_backward.  Thin binding: the autograd node validates dtypes / contiguity."""
    nodes, dn = x.shape
    edges, de = e.shape
    dh = params[8].size(1)
    offs, total = _gcn_saved_layout(nodes, edges, S, dn, de, dh)
    saved = torch.empty(total, dtype=torch.float32, device=x.device)
    out = torch.empty(nodes, dn, dtype=torch.float32, device=x.device)
    e_out = torch.empty(edges, de, dtype=torch.float32, device=x.device)
    base = saved.data_ptr()
    sp = [base + 4 * o for o in offs]
    sp[9], sp[14] = e_out.data_ptr(), out.data_ptr()
    L = GcnLayer(nodes, edges, S, dn, de, dh, int(bool(relu_out)), eps[0], eps[1], eps[2],
                 x.data_ptr(), e.data_ptr(), dst.data_ptr(), src.data_ptr(), order.data_ptr(), rowptr.data_ptr(),
                 node_ptr.data_ptr(), edge_ptr.data_ptr(), *[p.data_ptr() for p in params], *sp)
    wide = 2 * dh + de
    _call("pn2_gcn_layer_forward", x, ctypes.byref(L),
          alg_flops=2 * edges * ((2 * dn + de) * dh + dh * wide) + 2 * nodes * (dh * dh + dh * dn))
    return out, e_out, saved


def gcn_layer_backward(g_out, g_e, x, e, dst, src, order, rowptr, node_ptr, edge_ptr, S, relu_out, params, saved, out):
    """Backward of gcn_layer_forward in ONE C call (nine launches) -> (gx, ge, [dW1, db1, dgamma1, dbeta1, ..., dW4, db4])."""
    nodes, dn = x.shape
    edges, de = e.shape
    dh = params[8].size(1)
    offs, _ = _gcn_saved_layout(nodes, edges, S, dn, de, dh)
    f32 = torch.float32
    grads = zero_arena(x.device, [(tuple(p.shape), f32) for p in params] + [((nodes, dn), f32)])
    gx = grads.pop()
    ge = torch.empty_like(e)
    work = torch.empty(_lib.pn2_gcn_layer_backward_workspace_bytes(nodes, edges, dn, de, dh) // 4, dtype=f32, device=x.device)
    base = saved.data_ptr()
    sp = [base + 4 * o for o in offs]
    sp[9], sp[14] = None, out.data_ptr()
    L = GcnLayer(nodes, edges, S, dn, de, dh, int(bool(relu_out)), 0.0, 0.0, 0.0,
                 x.data_ptr(), e.data_ptr(), dst.data_ptr(), src.data_ptr(), order.data_ptr(), rowptr.data_ptr(),
                 node_ptr.data_ptr(), edge_ptr.data_ptr(), *[p.data_ptr() for p in params], *sp,
                 g_out.data_ptr(), g_e.data_ptr(), *[g.data_ptr() for g in grads], gx.data_ptr(), ge.data_ptr(), work.data_ptr())
    wide = 2 * dh + de
    _call("pn2_gcn_layer_backward", x, ctypes.byref(L),
          alg_flops=4 * edges * ((2 * dn + de) * dh + dh * wide) + 4 * nodes * (dh * dh + dh * dn))
    return gx, ge, grads


def segment_bn_rows(x, ptr, gamma, beta, eps, relu, h=None, col0=0):
    """x (R, ldx)[:, col0:col0+h], ptr (S+1) i64 -> (y (R,h), mean (S,h), rstd (S,h)): BatchNorm1d with the statistics of
    each row segment (scan) + optional ReLU."""
    _f32(x, "x"); _i64(ptr, "ptr"); _f32(gamma, "gamma"); _f32(beta, "beta")
    _same_device((x, "x"), (ptr, "ptr"), (gamma, "gamma"), (beta, "beta"))
    R, ldx = x.shape
    C = ldx if h is None else int(h)
    S = ptr.numel() - 1
    y = torch.empty(R, C, dtype=torch.float32, device=x.device)
    mean = torch.empty(S, C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(S, C, dtype=torch.float32, device=x.device)
    _call("pn2_segment_bn_rows", x, R, C, ldx, int(col0), S, _ptr(x), _ptr(ptr), _ptr(gamma), _ptr(beta), float(eps),
          int(bool(relu)), _ptr(y), _ptr(mean), _ptr(rstd), alg_bytes=8 * R * C + 8 * S * C)
    return y, mean, rstd


def segment_bn_rows_grad(grad_out, x, ptr, gamma, beta, mean, rstd, relu, col0=0, eps=None):
    """-> (grad_x (R,C), dgamma (C), dbeta (C)); `eps` is implied by the saved rstd (accepted for interface symmetry)."""
    _f32(grad_out, "grad_out"); _f32(x, "x"); _i64(ptr, "ptr")
    _same_device((grad_out, "grad_out"), (x, "x"), (ptr, "ptr"))
    R, C = grad_out.shape
    S = ptr.numel() - 1
    gx = torch.empty(R, C, dtype=torch.float32, device=x.device)
    dg = torch.empty(S, C, dtype=torch.float32, device=x.device)
    db = torch.empty(S, C, dtype=torch.float32, device=x.device)
    _call("pn2_segment_bn_rows_grad", x, R, C, x.size(1), int(col0), S, _ptr(grad_out), _ptr(x), _ptr(ptr), _ptr(gamma),
          _ptr(beta), _ptr(mean), _ptr(rstd), int(bool(relu)), _ptr(gx), _ptr(dg), _ptr(db),
          alg_bytes=12 * R * C + 16 * S * C)
    if S == 1:                                   # (one scan: its partial sums ARE the gradients — no reduction launches)
        return gx, dg.view(C), db.view(C)
    return gx, dg.sum(0), db.sum(0)


def segment_bn_running_update(mean, rstd, ptr, eps, momentum, running_mean, running_var, num_batches_tracked=None):
    """Running statistics after the S per-scan training batches whose (mean, rstd) (S,C) segment_bn_rows returned: the S
    momentum updates in scan order, unbiased variance, `num_batches_tracked` += S — one launch."""
    _f32(mean, "mean"); _f32(rstd, "rstd"); _i64(ptr, "ptr"); _f32(running_mean, "running_mean"); _f32(running_var, "running_var")
    S, C = mean.shape
    if ptr.numel() != S + 1 or running_mean.numel() != C or running_var.numel() != C:
        raise RuntimeError("segment_bn_running_update: mean / rstd (S,C), ptr (S+1), running statistics (C) expected")
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        _fail("segment_bn_running_update: num_batches_tracked must be int64")
    _call("pn2_segment_bn_running_update", mean, S, C, _ptr(mean), _ptr(rstd), _ptr(ptr), float(eps), float(momentum),
          _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked))


# ---------------------------------------------- fused shared-MLP kernels (A10)
PRO_NONE, PRO_BNRELU, PRO_GY, PRO_POOLG = 0, 1, 2, 3
EPI_NONE, EPI_STATS, EPI_MASK = 0, 1, 2


def mlp_gemm(X, W, pro=PRO_NONE, epi=EPI_NONE, X2=None, p=None, arg=None, gP=None, ns=0,
             stats=None, Yprev=None, e_fin=None, M=None):
    """Y (M,N) = pro(X) (M,K) @ W (N,K)^T with fused prologue / epilogue (include/pn2_hip.h).
    `p` = (p0, p1[, p2]) per-K vectors; `stats` (2,N) float64 is accumulated into."""
    _f32(W, "W")
    N, K = W.shape
    ref = X if X is not None else X2
    M = int(M if M is not None else ref.size(0))
    if X3_GEMM and X is not None and arg is None and M >= X3_MIN_ROWS and (epi != EPI_NONE or stats is None):
        # f32x3: epi 0 (no reductions) is epi 1 without a statistics buffer
        xepi = 1 if epi == EPI_NONE else int(epi)
        if (pro != PRO_GY or X3_DGRAD) and x3_gemm_supported(K, N, pro, xepi):
            return x3_gemm(X, W, int(pro), xepi, X2=X2, p=p, stats=stats, Yprev=Yprev, e_fin=e_fin, M=M)
    Y = torch.empty(M, N, dtype=torch.float32, device=W.device)
    p0 = p1 = p2 = None
    if p is not None:
        p0, p1 = p[0], p[1]
        p2 = p[2] if len(p) > 2 else None
    flops = 2 * M * N * K
    nbytes = 4 * (M * K * (2 if pro == PRO_GY else 1) + M * N * (2 if epi == EPI_MASK else 1) + N * K)   # POOLG reads y only
    _call("pn2_mlp_gemm", W, M, K, N, int(pro), int(epi), _ptr(X), _ptr(X2), _ptr(p0), _ptr(p1), _ptr(p2),
          _ptr(arg), _ptr(gP), int(ns), _ptr(W), _ptr(Y), _ptr(stats), _ptr(Yprev), _ptr(e_fin),
          alg_bytes=nbytes, alg_flops=flops, tag=(f"M{M},K{K},N{N},pro{int(pro)},epi{int(epi)}" if DETAIL_TAGS else None))
    return Y


def zero_arena(device, specs):
    """[(shape, dtype), ...] -> zero-filled tensors carved out of ONE allocation / ONE fill kernel.  The MLP
    autograd node needs ~6 small accumulators per layer (statistics, weight gradients); as separate torch.zeros
    calls they were ~80 four-microsecond launches per step."""
    offs, total = [], 0
    for shape, dtype in specs:
        n = 1
        for d in shape:
            n *= int(d)
        offs.append((total, n))
        total += (n * _ELEM[dtype] + 255) // 256 * 256
    buf = _zero_slab(torch.device(device), total)
    out = []
    for (shape, dtype), (o, n) in zip(specs, offs):
        out.append(buf[o:o + n * _ELEM[dtype]].view(dtype).view(*shape))
    return out


_ELEM = {torch.float32: 4, torch.float64: 8, torch.int32: 4, torch.int64: 8}
_SLAB_BYTES = 1 << 20
_slabs = {}        # (device, stream) -> [zero-filled uint8 slab, bytes handed out]
_slab_lock = threading.Lock()


def _zero_slab(device, nbytes):
    """`nbytes` of zeros on `device`: carved out of a 1 MB slab that is zero-filled ONCE (one fill kernel per ~50
    arenas instead of one per arena; every piece is handed out exactly once, the slab dies with its last view)."""
    # inside a hipGraph capture the fill must be a node of the graph (re-executed by every replay)
    if nbytes > _SLAB_BYTES // 4 or device.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    with _slab_lock:                               # forward (main thread) and backward (autograd thread) both carve
        ent = _slabs.get(key)
        if ent is None or ent[1] + nbytes > _SLAB_BYTES:
            ent = _slabs[key] = [torch.zeros(_SLAB_BYTES, dtype=torch.uint8, device=device), 0]
        o = ent[1]
        ent[1] = o + nbytes
        return ent[0][o:o + nbytes]


def mlp_wgrad(Yl, consts, X, gmode, amode, G=None, arg=None, gP=None, ns=0, a_fin=None, dW=None):
    """dW (N,K) = gy^T @ act, gy/act formed on the fly (include/pn2_hip.h).  `dW`: pre-zeroed output (optional)."""
    M, N = Yl.shape
    K = X.size(1)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_wgrad", Yl, M, N, K, int(gmode), int(amode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg),
          _ptr(gP), int(ns), _ptr(X), _ptr(a_fin), _ptr(dW),
          alg_bytes=4 * (M * N * (2 if gmode == PRO_GY else 1) + M * K + N * K), alg_flops=2 * M * N * K,
          tag=(f"M{M},N{N},K{K},g{int(gmode)},a{int(amode)}" if DETAIL_TAGS else None))
    return dW


def mlp_bwd_fused_supported(N, K):
    return bool(_lib.pn2_mlp_bwd_fused_supported(int(N), int(K)))


def mlp_bwd_fused(Yl, consts, W, Yprev, a_fin, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None):
    """dgrad + wgrad of one hidden layer in one pass -> (Gout (M,K), sums (2,K) f64, dW (N,K)).
    `sums` / `dW`: pre-zeroed accumulators (optional)."""
    M, N = Yl.shape
    K = Yprev.size(1)
    Gout = torch.empty(M, K, dtype=torch.float32, device=Yl.device)
    if sums is None:
        sums = torch.zeros(2, K, dtype=torch.float64, device=Yl.device)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_bwd_fused", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP),
          int(ns), _ptr(W), _ptr(Yprev), _ptr(a_fin), _ptr(Gout), _ptr(sums), _ptr(dW),
          alg_bytes=4 * (M * N * (2 if gmode == PRO_GY else 1) + 2 * M * K + N * K), alg_flops=4 * M * N * K,
          tag=(f"M{M},N{N},K{K},g{int(gmode)}" if DETAIL_TAGS else None))
    return Gout, sums, dW


def mlp_bwd_fused_fold_supported(N, K, K0):
    return bool(_lib.pn2_mlp_bwd_fused_fold_supported(int(N), int(K), int(K0)))


def mlp_bwd_fused_fold(Yl, consts, W, Yprev, a_fin, X, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None, P1=None):
    """As mlp_bwd_fused for the layer above the FIRST one, whose input rows X (M,K0) need no gradient: the input
    gradient is not stored, P1 (K,K0) = gz^T X is reduced instead -> (sums (2,K) f64, dW (N,K), P1)."""
    M, N = Yl.shape
    K, K0 = Yprev.size(1), X.size(1)
    if sums is None:
        sums = torch.zeros(2, K, dtype=torch.float64, device=Yl.device)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    if P1 is None:
        P1 = torch.zeros(K, K0, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_bwd_fused_fold", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP),
          int(ns), _ptr(W), _ptr(Yprev), _ptr(a_fin), _ptr(X), K0, _ptr(sums), _ptr(dW), _ptr(P1),
          alg_bytes=4 * (M * N * (2 if gmode == PRO_GY else 1) + M * K + M * K0 + N * K), alg_flops=4 * M * N * K,
          tag=(f"M{M},N{N},K{K},g{int(gmode)},fold{K0}" if DETAIL_TAGS else None))
    return sums, dW, P1


def mlp_gemm_first_supported(K0, K, N):
    return bool(_lib.pn2_mlp_gemm_first_supported(int(K0), int(K), int(N)))


def first_layer_stats(W0, gram, stats):
    """stats (2,N0) f64 = column sums of y_0 = X W0^T and of y_0^2, from gram = rows_gram(X) (no pass over y_0)."""
    N0, K0 = W0.shape
    _call("pn2_first_layer_stats", W0, N0, K0, _ptr(W0), _ptr(gram), _ptr(stats))
    return stats


def mlp_gemm_first(X0, W0, fin0, W, epi=EPI_NONE, stats=None):
    """Y (M,N) = relu(bn_0(X0 W0^T)) W^T: the second layer of a stack with the first one recomputed from its input rows
    X0 (M,K0 <= 8) — y_0 is never written (csrc/mlp_gemm.hip PRO_FIRST).  fin0 (4,N0) of pn2_bn_finalize."""
    _f32(X0, "X0"); _f32(W0, "W0"); _f32(W, "W"); _f32(fin0, "fin0")
    M, K0 = X0.shape
    N, K = W.shape
    if tuple(W0.shape) != (K, K0) or tuple(fin0.shape) != (4, K):
        raise RuntimeError("mlp_gemm_first: W0 (K,K0), fin0 (4,K), W (N,K) expected")
    if X3_GEMM and M >= X3_MIN_ROWS and (epi != EPI_NONE or stats is None) and _lib.pn2_x3_gemm_first_supported(K0, K, N):
        return x3_gemm_first(X0, W0, fin0, W, stats if epi != EPI_NONE else None)
    Y = torch.empty(M, N, dtype=torch.float32, device=W.device)
    _call("pn2_mlp_gemm_first", W, M, K0, K, N, int(epi), _ptr(X0), _ptr(W0), _ptr(fin0[2]), _ptr(fin0[3]), _ptr(W), _ptr(Y),
          _ptr(stats), alg_bytes=4 * (M * K0 + M * N + N * K), alg_flops=2 * M * N * K + 2 * M * K * K0,
          tag=(f"M{M},K0{K0},K{K},N{N},epi{int(epi)}" if DETAIL_TAGS else None))
    return Y


def x3_gemm_first(X0, W0, fin0, W, stats=None):
    """mlp_gemm_first on the split-bf16 product (csrc/x3_chain.hip: the eval level's chain on stored input rows, the first
    layer's BatchNorm folded into its fragments): Y (M, N) and, with `stats`, += the column sums of Y, Y^2."""
    M, K0 = X0.shape
    N, K = W.shape
    w0 = torch.empty((K // 32) * 3072, dtype=torch.uint8, device=W.device)
    _call("pn2_x3_pack_first", W0, K, K0, _ptr(W0), _ptr(fin0[2]), _ptr(fin0[3]), _ptr(w0))
    frags = x3_pack_weight(W.contiguous(), perm=True)      # the layer reads the first layer's accumulators: permuted contraction order
    Y = torch.empty(M, N, dtype=torch.float32, device=W.device)
    _call("pn2_x3_gemm_first", W, M, K0, K, N, _ptr(X0), _ptr(w0), _ptr(frags), _ptr(Y), _ptr(stats), _ptr(_x3_workspace(W.device)),
          alg_bytes=4 * (M * K0 + M * N + N * K), alg_flops=2 * M * N * K + 2 * M * K * K0, label="pn2_mlp_gemm_first",
          tag=(f"M{M},K0{K0},K{K},N{N},x3" if DETAIL_TAGS else None))
    return Y


def mlp_bwd_fused_fold_first(Yl, consts, W, W0, a_fin, X, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None, P1=None):
    """mlp_bwd_fused_fold when the forward was mlp_gemm_first: y_{l-1} is recomputed from X and W0 (K,K0)."""
    M, N = Yl.shape
    K, K0 = W0.shape
    if sums is None:
        sums = torch.zeros(2, K, dtype=torch.float64, device=Yl.device)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    if P1 is None:
        P1 = torch.zeros(K, K0, dtype=torch.float32, device=Yl.device)
    # (the f32x3 form of the two 64-deep products when the route is on — same outputs, y_0 still re-formed exactly)
    x3 = X3_GEMM and X3_BWD_FIRST and M >= X3_MIN_ROWS
    _call("pn2_x3_bwd_fold_first" if x3 else "pn2_mlp_bwd_fused_fold_first", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP),
          int(ns), _ptr(W), _ptr(W0), _ptr(a_fin), _ptr(X), K0, _ptr(sums), _ptr(dW), _ptr(P1),
          alg_bytes=4 * (M * N * (2 if gmode == PRO_GY else 1) + M * K0 + N * K), alg_flops=4 * M * N * K + 2 * M * K * K0,
          label="pn2_mlp_bwd_fused_fold_first", tag=(f"M{M},N{N},K{K},g{int(gmode)},first{K0}" + (",x3" if x3 else "") if DETAIL_TAGS else None))
    return sums, dW, P1


def rows_gram(X, gram=None):
    """gram (K0*K0 + K0,) f64 = [X^T X | column sums of X] for rows X (M,K0), K0 <= 8."""
    M, K0 = X.shape
    if gram is None:
        gram = torch.zeros(K0 * K0 + K0, dtype=torch.float64, device=X.device)
    _call("pn2_rows_gram", X, M, K0, _ptr(X), _ptr(gram), alg_bytes=4 * M * K0)
    return gram


def first_layer_dw(consts, P1, W0, gram):
    N, K0 = W0.shape
    dW0 = torch.empty(N, K0, dtype=torch.float32, device=W0.device)
    _call("pn2_first_layer_dw", W0, N, K0, _ptr(consts), _ptr(P1), _ptr(W0), _ptr(gram), _ptr(dW0))
    return dW0


def bn_running_update(fins, eps, decay, w, wu, running_mean, running_var, num_batches_tracked=None):
    """fins (S,4,C) of S scans -> running statistics after the S momentum updates (see include/pn2_hip.h)."""
    _f32(fins, "fins"); _f32(w, "w"); _f32(wu, "wu"); _f32(running_mean, "running_mean"); _f32(running_var, "running_var")
    S, four, C = fins.shape
    if four != 4 or w.numel() != S or wu.numel() != S or running_mean.numel() != C or running_var.numel() != C:
        raise RuntimeError("bn_running_update: fins (S,4,C), w / wu (S), running statistics (C) expected")
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        _fail("bn_running_update: num_batches_tracked must be int64")
    _call("pn2_bn_running_update", fins, S, C, _ptr(fins), float(eps), float(decay), _ptr(w), _ptr(wu), _ptr(running_mean),
          _ptr(running_var), _ptr(num_batches_tracked))


class SegTable:
    """Row ranges of the scans of a batched call with per-scan statistics: `rows` (host ints, rows of every scan in the
    stack's row tensors), `ptr` (S + 1 int64 offsets on the device).  Built once per (device, rows) signature — from host
    numbers, i.e. with a copy, which a stream capture does not allow: the first eager step creates it."""
    _CACHE = collections.OrderedDict()
    _MAX = 1024

    def __init__(self, device, rows):
        self.rows = tuple(int(r) for r in rows)
        self.nseg = len(self.rows)
        self.max_rows = max(self.rows) if self.rows else 0
        self.total = sum(self.rows)
        off = [0]
        for r in self.rows:
            off.append(off[-1] + r)
        self.ptr = torch.tensor(off, dtype=torch.int64, device=device)

    @classmethod
    def get(cls, device, rows):
        key = (device, tuple(int(r) for r in rows))
        hit = cls._CACHE.get(key)
        if hit is None:
            if len(cls._CACHE) >= cls._MAX:
                cls._CACHE.popitem(last=False)
            hit = cls._CACHE[key] = cls(device, rows)
        else:
            cls._CACHE.move_to_end(key)
        return hit


def bn_finalize_seg(stats, seg, gamma, beta, eps, momentum=0.0, running_mean=None, running_var=None,
                    num_batches_tracked=None, out=None):
    """stats (S,2,N) f64 of the scans of `seg` -> fin (S,4,N) = per scan [mean | rstd | scale | shift]; the running
    statistics (optional) receive the S momentum updates in scan order, `num_batches_tracked` (int64 scalar) += S."""
    S, two, N = stats.shape
    if S != seg.nseg or two != 2:
        raise RuntimeError("bn_finalize_seg: stats must be (S, 2, N) for the S scans of the segment table")
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        _fail("bn_finalize_seg: num_batches_tracked must be int64")
    fin = out if out is not None else torch.empty(S, 4, N, dtype=torch.float32, device=stats.device)
    _call("pn2_bn_finalize_seg", stats, S, N, _ptr(seg.ptr), _ptr(stats), _ptr(gamma), _ptr(beta), float(eps),
          float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(fin))
    return fin


def bn_bwd_consts_seg(sums, seg, gamma, fin, use_batch_stats, W=None, k0=0):
    """sums (S,2,N), fin (S,4,N) -> (consts (S,3,N), dgamma (N), dbeta (N) = sums over the scans[, Wt])."""
    S, two, N = sums.shape
    if S != seg.nseg or two != 2:
        raise RuntimeError("bn_bwd_consts_seg: sums must be (S, 2, N) for the S scans of the segment table")
    consts = torch.empty(S, 3, N, dtype=torch.float32, device=sums.device)
    dgamma = torch.empty(N, dtype=torch.float32, device=sums.device)
    dbeta = torch.empty(N, dtype=torch.float32, device=sums.device)
    Wt, K = None, 0
    if W is not None:
        K = W.size(1)
        Wt = torch.empty(K - int(k0), N, dtype=torch.float32, device=sums.device)
    _call("pn2_bn_bwd_consts_seg", sums, S, N, _ptr(seg.ptr), _ptr(sums), _ptr(gamma), _ptr(fin), int(bool(use_batch_stats)),
          _ptr(consts), _ptr(dgamma), _ptr(dbeta), _ptr(W), int(K), int(k0), _ptr(Wt))
    return (consts, dgamma, dbeta) if W is None else (consts, dgamma, dbeta, Wt)


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked=None, out=None):
    """fin (4,N) = [mean | rstd | scale | shift]; updates the running statistics in place and, if given, bumps the int64
    scalar `num_batches_tracked` (what _BatchNorm.forward does with a separate add_ kernel per layer).
    `out`: a contiguous (4,N) fp32 destination (e.g. one scan's block of an (S,4,N) buffer)."""
    N = stats.size(1)
    if out is None:
        fin = torch.empty(4, N, dtype=torch.float32, device=stats.device)
    else:
        _f32(out, "out")
        if tuple(out.shape) != (4, N):
            raise RuntimeError("bn_finalize: out must be a contiguous float (4, N) tensor")
        fin = out
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        _fail("bn_finalize: num_batches_tracked must be int64")
    _call("pn2_bn_finalize", stats, N, float(count), _ptr(stats), _ptr(gamma), _ptr(beta), float(eps),
          float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(fin))
    return fin


def bn_bwd_consts(sums, count, gamma, fin, use_batch_stats, want_param_grads=True, W=None, k0=0):
    """-> (consts (3,N), dgamma, dbeta[, Wt]).  With `W` (N,K) the kernel also writes Wt (K - k0, N) = W[:, k0:]^T, the
    weight layout the dgrad call of the layer takes."""
    N = sums.size(1)
    consts = torch.empty(3, N, dtype=torch.float32, device=sums.device)
    dgamma = torch.empty(N, dtype=torch.float32, device=sums.device) if want_param_grads else None
    dbeta = torch.empty(N, dtype=torch.float32, device=sums.device) if want_param_grads else None
    Wt, K = None, 0
    if W is not None:
        K = W.size(1)
        Wt = torch.empty(K - int(k0), N, dtype=torch.float32, device=sums.device)
    _call("pn2_bn_bwd_consts", sums, N, float(count), _ptr(sums), _ptr(gamma), _ptr(fin), int(bool(use_batch_stats)),
          _ptr(consts), _ptr(dgamma), _ptr(dbeta), _ptr(W), int(K), int(k0), _ptr(Wt))
    return (consts, dgamma, dbeta) if W is None else (consts, dgamma, dbeta, Wt)


def bn_relu_apply(y, fin):
    M, N = y.shape
    out = torch.empty_like(y)
    _call("pn2_bn_relu_apply", y, M, N, _ptr(y), _ptr(fin), _ptr(out), alg_bytes=8 * M * N)
    return out


def bn_relu_bwd_prep(y, gout, fin, sums=None):
    M, N = y.shape
    gpre = torch.empty_like(y)
    if sums is None:
        sums = torch.zeros(2, N, dtype=torch.float64, device=y.device)
    _call("pn2_bn_relu_bwd_prep", y, M, N, _ptr(y), _ptr(gout), _ptr(fin), _ptr(gpre), _ptr(sums), alg_bytes=12 * M * N)
    return gpre, sums


def bn_relu_rows_max(y, fin, ns):
    M, C = y.shape
    R = M // int(ns)
    out = torch.empty(R, C, dtype=torch.float32, device=y.device)
    arg = torch.empty(R, C, dtype=torch.int32, device=y.device)
    yraw = torch.empty(R, C, dtype=torch.float32, device=y.device)
    _call("pn2_bn_relu_rows_max", y, R, int(ns), C, _ptr(y), _ptr(fin), _ptr(out), _ptr(arg), _ptr(yraw),
          alg_bytes=4 * M * C + 12 * R * C)
    return out, arg, yraw


def pool_flip_rows(W, gamma):
    """(Wf, sgn): Wf = diag(sgn) W with sgn = -1 where gamma < 0 (pn2_pool_flip_rows)."""
    _f32(W, "W")
    N, K = W.shape
    Wf = torch.empty_like(W)
    sgn = torch.empty(N, dtype=torch.float32, device=W.device)
    _call("pn2_pool_flip_rows", W, N, K, _ptr(W), _ptr(gamma), _ptr(Wf), _ptr(sgn))
    return Wf, sgn


def pool_layer_supported(K, N, ns):
    """Shapes pn2_mlp_gemm_pool covers."""
    return ns in (16, 32, 64, 128) and N <= 320 and K <= 2048


def mlp_gemm_pool(X, Wf, sgn, ns, p=None, stats=None):
    """Pooled last layer without its (M, N) output: -> (pmax, parg) partial maxima (M / min(ns, 32), N) of
    pro(X) @ Wf^T; `stats` (2, N) float64 accumulates the column sums (include/pn2_hip.h)."""
    _f32(X, "X"); _f32(Wf, "Wf")
    N, K = Wf.shape
    M = X.size(0)
    if X3_GEMM and p is not None and M >= X3_MIN_ROWS and x3_gemm_supported(K, N, PRO_BNRELU, 3, ns):
        return x3_gemm(X, Wf, PRO_BNRELU, 3, p=p, stats=stats, sgn=sgn, ns=ns)
    psz = min(int(ns), 32)
    pmax = torch.empty(M // psz, N, dtype=torch.float32, device=X.device)
    parg = torch.empty(M // psz, N, dtype=torch.int32, device=X.device)
    p0, p1 = (None, None) if p is None else (p[0], p[1])
    _call("pn2_mlp_gemm_pool", X, M, K, N, PRO_NONE if p is None else PRO_BNRELU, _ptr(X), _ptr(p0), _ptr(p1), _ptr(Wf),
          _ptr(sgn), int(ns), _ptr(stats), _ptr(pmax), _ptr(parg),
          alg_bytes=4 * (M * K + N * K + 2 * (M // psz) * N), alg_flops=2 * M * N * K,
          tag=(f"M{M},K{K},N{N},ns{int(ns)}" if DETAIL_TAGS else None))
    return pmax, parg


def pool_finalize(pmax, parg, fin, sgn, ns):
    """-> (out (R,C), arg (R,C) int32, yraw (R,C)) like bn_relu_rows_max, from the partial maxima of mlp_gemm_pool."""
    P, C = pmax.shape
    psz = min(int(ns), 32)
    R = P * psz // int(ns)
    out = torch.empty(R, C, dtype=torch.float32, device=pmax.device)
    arg = torch.empty(R, C, dtype=torch.int32, device=pmax.device)
    yraw = torch.empty(R, C, dtype=torch.float32, device=pmax.device)
    _call("pn2_pool_finalize", pmax, R, C, int(ns), _ptr(pmax), _ptr(parg), _ptr(fin), _ptr(sgn), _ptr(out), _ptr(arg),
          _ptr(yraw), alg_bytes=8 * P * C + 12 * R * C)
    return out, arg, yraw


def pool_bwd_supported(N, K, ns):
    return bool(_lib.pn2_pool_bwd_supported(int(N), int(K), int(ns)))


def pool_bwd(Yp, fin_p, W, consts, arg, gPm, ns, sums):
    """Gram-form backward of the pooled last layer (pn2_pool_bwd): -> (Gout (M,K), dW (N,K)); `sums` (2,K) fp64
    accumulates the BatchNorm-backward column sums of the layer below."""
    _f32(Yp, "Yp"); _f32(W, "W")
    M, K = Yp.shape
    N = W.size(0)
    Gout = torch.empty(M, K, dtype=torch.float32, device=Yp.device)
    dW = torch.empty(N, K, dtype=torch.float32, device=Yp.device)
    nb = int(_lib.pn2_pool_bwd_workspace_bytes(M, N, K))
    ws = torch.empty(nb, dtype=torch.uint8, device=Yp.device)
    # (X3: the K = 64 kernel's matrix products on the f32x3 product; other shapes run the exact kernels behind the same entry)
    x3 = X3_GEMM and X3_POOL_BWD and M >= X3_MIN_ROWS and K == 64 and N <= 128
    _call("pn2_x3_pool_bwd" if x3 else "pn2_pool_bwd", Yp, M, N, K, int(ns), _ptr(Yp), _ptr(fin_p), _ptr(W), _ptr(consts), _ptr(arg), _ptr(gPm),
          _ptr(Gout), _ptr(sums), _ptr(dW), _ptr(ws), nb, alg_bytes=4 * (2 * M * K + 2 * (M // int(ns)) * N + N * K),
          alg_flops=4 * M * K * K, label="pn2_pool_bwd", tag=(f"M{M},N{N},K{K},ns{int(ns)}" + (",x3" if x3 else "") if DETAIL_TAGS else None))
    return Gout, dW


def pool_bwd_prep(yraw, pooled, gP, fin, sums=None, seg=None, ns=0):
    """`seg` (SegTable over the un-pooled rows, `ns` rows per pooled row): fin (S,4,C), sums (S,2,C) per scan."""
    R, C = pooled.shape
    gPm = torch.empty_like(pooled)
    if seg is not None:
        if sums is None:
            sums = torch.zeros(seg.nseg, 2, C, dtype=torch.float64, device=pooled.device)
        _call("pn2_pool_bwd_prep_seg", pooled, R, C, _ptr(yraw), _ptr(pooled), _ptr(gP), _ptr(fin), _ptr(gPm), _ptr(sums),
              _ptr(seg.ptr), seg.nseg, seg.max_rows, int(ns), alg_bytes=16 * R * C)
        return gPm, sums
    if sums is None:
        sums = torch.zeros(2, C, dtype=torch.float64, device=pooled.device)
    _call("pn2_pool_bwd_prep", pooled, R, C, _ptr(yraw), _ptr(pooled), _ptr(gP), _ptr(fin), _ptr(gPm),
          _ptr(sums), alg_bytes=16 * R * C)
    return gPm, sums


# ------------------------------------------- mixed precision: bf16 activations, fp32 weights / statistics (A10)
HAS_BF16_MLP = True
BF16 = torch.bfloat16


def _bf16(t, name):
    _check_cuda(t, name)
    if t.dtype != torch.bfloat16 or not t.is_contiguous():
        _fail(f"{name} must be a contiguous bfloat16 tensor")


def pad8(k):
    return (int(k) + 7) // 8 * 8


def group_concat_rows_bf16(xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius):
    """As group_concat_rows, rows written as bf16 with a pitch rounded up to 8 columns (zero pad): (B,m,ns,pad8(Cx+C))."""
    _i32(idx, "idx")
    B, m, ns = idx.shape
    N = xyz.size(1)
    C = 0
    if use_xyz:
        _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    if feats_rows is not None:
        _f32(feats_rows, "features")
        C = feats_rows.size(2)
    _same_device((xyz, "xyz"), (idx, "idx"), (new_xyz, "new_xyz"), (feats_rows, "features"))
    W = (3 if use_xyz else 0) + C
    ldo = pad8(W)
    out = torch.empty(B, m, ns, ldo, dtype=torch.bfloat16, device=xyz.device)
    _call("pn2_group_concat_rows_bf16", xyz, B, N, m, ns, C, int(bool(use_xyz)), int(bool(normalize)),
          float(radius if radius is not None else 1.0), ldo, _ptr(xyz), _ptr(new_xyz), _ptr(feats_rows), _ptr(idx), _ptr(out),
          alg_bytes=B * (4 * m * ns + (12 * N + 12 * m if use_xyz else 0) + 4 * C * N + 2 * ldo * m * ns))
    return out


def mlp_gemm_bf16(X, W, pro=PRO_NONE, epi=EPI_NONE, X2=None, p=None, arg=None, gP=None, ns=0, stats=None, Yprev=None,
                  e_fin=None, M=None, out_f32=False, seg=None):
    """Y (M, N) = pro(X) (M, K) @ W (N, K)^T on the bf16 MFMA path.  X: bf16 rows (pitch = X.size(1), a multiple of 8, may
    exceed K with zero pad columns) or fp32 rows (pro 0 only); W fp32; Y bf16 (fp32 when out_f32)."""
    _f32(W, "W")
    N, K = W.shape
    ref = X if X is not None else X2
    M = int(M if M is not None else ref.size(0))
    x_f32 = ref.dtype == torch.float32
    ldx = ref.size(1)
    Y = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=W.device)
    p0 = p1 = p2 = None
    if p is not None:
        p0, p1 = p[0], p[1]
        p2 = p[2] if len(p) > 2 else None
    eb = 4 if x_f32 else 2
    nbytes = (M * ldx * eb * (2 if pro == PRO_GY else 1) + M * N * (4 if out_f32 else 2) * (2 if epi == EPI_MASK else 1)
              + 4 * N * K)
    if seg is not None:
        # `p`: (S, K) VIEWS into the per-scan blocks — fin[:, 2], fin[:, 3] of an (S,4,K) finalize buffer (pitch 4K) or
        # consts[:, i] of the (S,3,K) constants (pitch 3K); stats (S,2,N), e_fin (S,4,N)
        pstride = 0
        if p0 is not None:
            if any(q is not None and (q.dim() != 2 or q.size(0) != seg.nseg or q.stride(1) != 1 or q.stride(0) != p0.stride(0))
                   for q in (p0, p1, p2)):
                raise RuntimeError("mlp_gemm_bf16(seg=...): p must be (S, K) views with one common scan pitch")
            pstride = int(p0.stride(0))
        _call("pn2_mlp_gemm_bf16_seg", W, M, K, N, int(pro), int(epi), int(x_f32), int(bool(out_f32)), ldx, N, _ptr(X),
              _ptr(X2), _ptr(p0), _ptr(p1), _ptr(p2), pstride, _ptr(arg), _ptr(gP), int(ns), _ptr(W), _ptr(Y), _ptr(stats),
              _ptr(Yprev), _ptr(e_fin), _ptr(seg.ptr), seg.nseg, seg.max_rows, alg_bytes=nbytes, alg_flops=2 * M * N * K,
              tag=(f"M{M},K{K},N{N},pro{int(pro)},epi{int(epi)},S{seg.nseg}" if DETAIL_TAGS else None))
        return Y
    _call("pn2_mlp_gemm_bf16", W, M, K, N, int(pro), int(epi), int(x_f32), int(bool(out_f32)), ldx, N, _ptr(X), _ptr(X2),
          _ptr(p0), _ptr(p1), _ptr(p2), _ptr(arg), _ptr(gP), int(ns), _ptr(W), _ptr(Y), _ptr(stats), _ptr(Yprev), _ptr(e_fin),
          alg_bytes=nbytes, alg_flops=2 * M * N * K,
          tag=(f"M{M},K{K},N{N},pro{int(pro)},epi{int(epi)}" if DETAIL_TAGS else None))
    return Y


def mlp_bwd_bf16_fold_supported(N, K, K0):
    return bool(_lib.pn2_mlp_bwd_bf16_fold_supported(int(N), int(K), int(K0)))


def mlp_bwd_bf16_fold(Yl, consts, Wt, Yprev, a_fin, X, K0, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None, P1=None):
    """mlp_bwd_bf16 for the layer above the first one, X (M, 8) bf16 its input rows without a gradient: no Gout,
    P1 (K, K0) += gz^T X instead -> (sums, dW, P1)."""
    M, N = Yl.shape
    K = Yprev.size(1)
    _call("pn2_mlp_bwd_bf16_fold", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP), int(ns),
          _ptr(Wt), _ptr(Yprev), _ptr(a_fin), _ptr(X), int(K0), _ptr(sums), _ptr(dW), _ptr(P1),
          alg_bytes=2 * (M * N * (2 if gmode == PRO_GY else 1) + M * K + 8 * M) + 4 * N * K, alg_flops=4 * M * N * K,
          tag=(f"M{M},N{N},K{K},g{int(gmode)},fold{int(K0)}" if DETAIL_TAGS else None))
    return sums, dW, P1


def mlp_gemm_first_bf16_supported(K0, K, N) -> bool:
    return bool(_lib.pn2_mlp_gemm_first_bf16_supported(int(K0), int(K), int(N)))


def mlp_gemm_first_bf16(X0, K0, W0, fin0, W, stats):
    """Y (M, N) bf16 = relu(bn_0(X0 W0^T)) W^T, X0 (M, 8) bf16 rows with K0 real columns: the second layer of a bf16 stack with
    the first one re-formed while the A tile is staged — y_0 is never stored (csrc/mlp_bf16.hip PRO_FIRST)."""
    _bf16(X0, "X0"); _f32(W0, "W0"); _f32(W, "W"); _f32(fin0, "fin0")
    M = X0.size(0)
    N, K = W.shape
    if X0.size(1) != 8 or tuple(W0.shape) != (K, int(K0)) or tuple(fin0.shape) != (4, K):
        raise RuntimeError("mlp_gemm_first_bf16: X0 (M,8) bf16, W0 (K,K0), fin0 (4,K), W (N,K) expected")
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=W.device)
    _call("pn2_mlp_gemm_first_bf16", W, M, int(K0), K, N, _ptr(X0), _ptr(W0), _ptr(fin0), _ptr(W), _ptr(Y), _ptr(stats),
          alg_bytes=16 * M + 2 * M * N + 4 * N * K, alg_flops=2 * M * N * K + 2 * M * K * int(K0),
          tag=(f"M{M},K0{int(K0)},K{K},N{N}" if DETAIL_TAGS else None))
    return Y


def mlp_bwd_bf16_fold_first(Yl, consts, Wt, W0, a_fin, X, K0, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None, P1=None):
    """mlp_bwd_bf16_fold when the forward was mlp_gemm_first_bf16: y_0 is re-formed from X and W0 (K, K0) -> (sums, dW, P1)."""
    M, N = Yl.shape
    K = W0.size(0)
    _call("pn2_mlp_bwd_bf16_fold_first", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP), int(ns),
          _ptr(Wt), _ptr(W0), _ptr(a_fin), _ptr(X), int(K0), _ptr(sums), _ptr(dW), _ptr(P1),
          alg_bytes=2 * (M * N * (2 if gmode == PRO_GY else 1) + 8 * M) + 4 * N * K, alg_flops=4 * M * N * K + 2 * M * K * int(K0),
          tag=(f"M{M},N{N},K{K},g{int(gmode)},first{int(K0)}" if DETAIL_TAGS else None))
    return sums, dW, P1


def rows_gram_bf16(X, K0, gram):
    """gram (K0*K0 + K0,) f64 += [X^T X | column sums] for bf16 rows X (M, 8) with K0 real columns."""
    _call("pn2_rows_gram_bf16", X, X.size(0), int(K0), _ptr(X), _ptr(gram), alg_bytes=16 * X.size(0))
    return gram


def mlp_wgrad_bf16(Yl, consts, X, gmode, amode, K, G=None, arg=None, gP=None, ns=0, a_fin=None, dW=None, seg=None):
    """dW (N, K) fp32 += gy^T @ act; Yl / G bf16 (M, N); X bf16 (M, ldx >= K) or fp32 rows (amode 0).
    `seg`: consts (S,3,N), a_fin (S,4,K) per scan; dW = the sum over the scans."""
    M, N = Yl.shape
    x_f32 = X.dtype == torch.float32
    if dW is None:
        dW = torch.zeros(N, int(K), dtype=torch.float32, device=Yl.device)
    if seg is not None:
        _call("pn2_mlp_wgrad_bf16_seg", Yl, M, N, int(K), int(gmode), int(amode), int(x_f32), X.size(1), _ptr(G), _ptr(Yl),
              _ptr(consts), _ptr(arg), _ptr(gP), int(ns), _ptr(X), _ptr(a_fin), _ptr(dW), _ptr(seg.ptr), seg.nseg,
              seg.max_rows,
              alg_bytes=2 * M * N * (2 if gmode == PRO_GY else 1) + M * X.size(1) * (4 if x_f32 else 2) + 4 * N * int(K),
              alg_flops=2 * M * N * int(K),
              tag=(f"M{M},N{N},K{K},g{int(gmode)},a{int(amode)},S{seg.nseg}" if DETAIL_TAGS else None))
        return dW
    _call("pn2_mlp_wgrad_bf16", Yl, M, N, int(K), int(gmode), int(amode), int(x_f32), X.size(1), _ptr(G), _ptr(Yl),
          _ptr(consts), _ptr(arg), _ptr(gP), int(ns), _ptr(X), _ptr(a_fin), _ptr(dW),
          alg_bytes=2 * M * N * (2 if gmode == PRO_GY else 1) + M * X.size(1) * (4 if x_f32 else 2) + 4 * N * int(K),
          alg_flops=2 * M * N * int(K), tag=(f"M{M},N{N},K{K},g{int(gmode)},a{int(amode)}" if DETAIL_TAGS else None))
    return dW


def mlp_bwd_bf16_supported(N, K):
    return bool(_lib.pn2_mlp_bwd_bf16_supported(int(N), int(K)))


def mlp_bwd_bf16(Yl, consts, Wt, Yprev, a_fin, gmode, G=None, arg=None, gP=None, ns=0, sums=None, dW=None, seg=None):
    """One-pass backward of a hidden layer on bf16 tensors -> (Gout (M,K) bf16, sums (2,K) f64, dW (N,K) f32).
    Wt = the layer's weights transposed (K, N) fp32; `sums` / `dW`: pre-zeroed accumulators (optional).
    `seg`: consts (S,3,N), a_fin (S,4,K), sums (S,2,K) per scan; dW = the sum over the scans."""
    M, N = Yl.shape
    K = Yprev.size(1)
    Gout = torch.empty(M, K, dtype=torch.bfloat16, device=Yl.device)
    if seg is not None:
        if sums is None:
            sums = torch.zeros(seg.nseg, 2, K, dtype=torch.float64, device=Yl.device)
        if dW is None:
            dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
        _call("pn2_mlp_bwd_bf16_seg", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP), int(ns),
              _ptr(Wt), _ptr(Yprev), _ptr(a_fin), _ptr(Gout), _ptr(sums), _ptr(dW), _ptr(seg.ptr), seg.nseg, seg.max_rows,
              alg_bytes=2 * (M * N * (2 if gmode == PRO_GY else 1) + 2 * M * K) + 4 * N * K, alg_flops=4 * M * N * K,
              tag=(f"M{M},N{N},K{K},g{int(gmode)},S{seg.nseg}" if DETAIL_TAGS else None))
        return Gout, sums, dW
    if sums is None:
        sums = torch.zeros(2, K, dtype=torch.float64, device=Yl.device)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yl.device)
    _call("pn2_mlp_bwd_bf16", Yl, M, N, K, int(gmode), _ptr(G), _ptr(Yl), _ptr(consts), _ptr(arg), _ptr(gP), int(ns),
          _ptr(Wt), _ptr(Yprev), _ptr(a_fin), _ptr(Gout), _ptr(sums), _ptr(dW),
          alg_bytes=2 * (M * N * (2 if gmode == PRO_GY else 1) + 2 * M * K) + 4 * N * K, alg_flops=4 * M * N * K,
          tag=(f"M{M},N{N},K{K},g{int(gmode)}" if DETAIL_TAGS else None))
    return Gout, sums, dW


def pool_layer_bf16_supported(K, N, ns) -> bool:
    """Shapes the pooled last layer of a bf16 stack can take without its output (forward AND backward)."""
    return bool(_lib.pn2_mlp_gemm_pool_bf16_supported(int(K), int(N), int(ns))) and bool(_lib.pn2_mlp_bwd_bf16_pool_supported(int(N), int(K)))


def mlp_gemm_pool_bf16(X, Wf, sgn, ns, p, stats):
    """bf16 counterpart of mlp_gemm_pool: X (M, K) bf16 = y_{L-1}, p = (scale, shift) of its BatchNorm -> (pmax, parg) partial
    maxima (M / min(ns, 32), N) of the fp32 accumulators of relu(bn(X)) @ Wf^T; nothing of the (M, N) output is stored."""
    _bf16(X, "X"); _f32(Wf, "Wf")
    N, K = Wf.shape
    M = X.size(0)
    psz = min(int(ns), 32)
    pmax = torch.empty(M // psz, N, dtype=torch.float32, device=X.device)
    parg = torch.empty(M // psz, N, dtype=torch.int32, device=X.device)
    _call("pn2_mlp_gemm_pool_bf16", X, M, K, N, X.size(1), _ptr(X), _ptr(p[0]), _ptr(p[1]), _ptr(Wf), _ptr(sgn), int(ns),
          _ptr(stats), _ptr(pmax), _ptr(parg), alg_bytes=2 * M * K + 4 * N * K + 8 * (M // psz) * N, alg_flops=2 * M * N * K,
          tag=(f"M{M},K{K},N{N},ns{int(ns)}" if DETAIL_TAGS else None))
    return pmax, parg


def mlp_bwd_bf16_pool(consts, Wt, Yprev, a_fin, arg, gP, ns, sums=None, dW=None):
    """mlp_bwd_bf16 (gmode PRO_POOLG) for a pooled last layer whose output was not stored: y_L is re-formed from Yprev inside
    the kernel -> (Gout (M, K) bf16, sums (2, K) f64, dW (N, K) f32)."""
    _bf16(Yprev, "Yprev")
    M, K = Yprev.shape
    N = Wt.size(1)
    Gout = torch.empty(M, K, dtype=torch.bfloat16, device=Yprev.device)
    if sums is None:
        sums = torch.zeros(2, K, dtype=torch.float64, device=Yprev.device)
    if dW is None:
        dW = torch.zeros(N, K, dtype=torch.float32, device=Yprev.device)
    _call("pn2_mlp_bwd_bf16_pool", Yprev, M, N, K, _ptr(consts), _ptr(arg), _ptr(gP), int(ns), _ptr(Wt), _ptr(Yprev), _ptr(a_fin),
          _ptr(Gout), _ptr(sums), _ptr(dW), alg_bytes=2 * (2 * M * K) + 4 * N * K + 8 * (M // int(ns)) * N, alg_flops=6 * M * N * K,
          tag=(f"M{M},N{N},K{K},pool" if DETAIL_TAGS else None))
    return Gout, sums, dW


def bn_relu_apply_bf16(y, fin):
    _bf16(y, "y")
    M, N = y.shape
    out = torch.empty(M, N, dtype=torch.float32, device=y.device)
    _call("pn2_bn_relu_apply_bf16", y, M, N, _ptr(y), _ptr(fin), _ptr(out), alg_bytes=6 * M * N)
    return out


def bn_relu_bwd_prep_bf16(y, gout, fin, sums=None):
    _bf16(y, "y"); _f32(gout, "gout")
    M, N = y.shape
    gpre = torch.empty_like(y)
    if sums is None:
        sums = torch.zeros(2, N, dtype=torch.float64, device=y.device)
    _call("pn2_bn_relu_bwd_prep_bf16", y, M, N, _ptr(y), _ptr(gout), _ptr(fin), _ptr(gpre), _ptr(sums), alg_bytes=8 * M * N)
    return gpre, sums


def bn_relu_rows_max_bf16(y, fin, ns, seg=None):
    _bf16(y, "y")
    M, C = y.shape
    R = M // int(ns)
    out = torch.empty(R, C, dtype=torch.float32, device=y.device)
    arg = torch.empty(R, C, dtype=torch.int32, device=y.device)
    yraw = torch.empty(R, C, dtype=torch.float32, device=y.device)
    if seg is not None:                               # fin (S,4,C): per-scan constants
        _call("pn2_bn_relu_rows_max_bf16_seg", y, R, int(ns), C, _ptr(y), _ptr(fin), _ptr(out), _ptr(arg), _ptr(yraw),
              _ptr(seg.ptr), seg.nseg, seg.max_rows, alg_bytes=2 * M * C + 12 * R * C)
        return out, arg, yraw
    _call("pn2_bn_relu_rows_max_bf16", y, R, int(ns), C, _ptr(y), _ptr(fin), _ptr(out), _ptr(arg), _ptr(yraw),
          alg_bytes=2 * M * C + 12 * R * C)
    return out, arg, yraw


# ------------------------------------------------- (f)3: crops of a fused scan (dataset/gpu_preparation.py drives these)
def prepare_scan_crops(points, masks, edges, n_obj, t_obj, t_rel, padding, seed):
    """points (P, ld) f32, masks (P) i32, edges (2, E) i32 -> (obj (n_obj, t_obj, ld), rel (E, t_rel, ld+1), boxes, sel)
    through the four pn2_prep_* kernels + one torch.cumsum (see include/pn2_hip.h)."""
    _f32(points, "points"); _i32(masks, "masks"); _i32(edges, "edges")
    _same_device((points, "points"), (masks, "masks"), (edges, "edges"))
    P, ld = points.shape
    E = edges.size(1)
    dev = points.device
    keys = torch.empty(max(n_obj, 1) * 6, dtype=torch.int32, device=dev)
    boxes = torch.empty(n_obj, 6, dtype=torch.float32, device=dev)
    _call("pn2_prep_object_boxes", points, P, ld, n_obj, float(padding), _ptr(points), _ptr(masks), _ptr(keys), _ptr(boxes),
          alg_bytes=P * (12 + 4))
    nch = int(_lib.pn2_prep_num_chunks(P))
    crops = n_obj + E
    counts = torch.empty(crops, nch, dtype=torch.int32, device=dev)
    _call("pn2_prep_chunk_counts", points, P, ld, n_obj, E, _ptr(points), _ptr(masks), _ptr(boxes), _ptr(edges), _ptr(counts),
          alg_bytes=crops * P * 16)
    prefix = torch.zeros(crops, nch + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, dim=1, out=prefix[:, 1:])
    slots = n_obj * t_obj + E * t_rel
    sel = torch.empty(slots, dtype=torch.int32, device=dev)
    _call("pn2_prep_select", points, P, ld, n_obj, E, int(t_obj), int(t_rel), int(seed) & 0xFFFFFFFF, _ptr(points), _ptr(masks),
          _ptr(boxes), _ptr(edges), _ptr(prefix), _ptr(sel), alg_bytes=slots * 4)
    obj = torch.empty(n_obj, t_obj, ld, dtype=torch.float32, device=dev)
    rel = torch.empty(E, t_rel, ld + 1, dtype=torch.float32, device=dev)
    _call("pn2_prep_gather_normalise", points, ld, n_obj, E, int(t_obj), int(t_rel), _ptr(points), _ptr(masks), _ptr(edges),
          _ptr(sel), _ptr(obj), _ptr(rel), alg_bytes=slots * (4 + 8 * ld))
    return obj, rel, boxes, sel, prefix[:, -1]


def prep_voxel_keys(pts, min_bound, size):
    """pts (n, ld >= 3) f32 rows, min_bound (3) f32 -> (n) int64 trace-slot keys of one voxel-ladder rung (pn2_prep_voxel_keys)."""
    _f32(pts, "pts"); _f32(min_bound, "min_bound")
    _same_device((pts, "pts"), (min_bound, "min_bound"))
    n, ld = pts.shape
    keys = torch.empty(n, dtype=torch.int64, device=pts.device)
    _call("pn2_prep_voxel_keys", pts, n, ld, _ptr(pts), _ptr(min_bound), float(size), _ptr(keys), alg_bytes=n * 20)
    return keys


def prep_object_boxes(points, masks, n_obj, padding):
    """(n_obj, 6) padded boxes [min | max] of the object members (pn2_prep_object_boxes)."""
    _f32(points, "points"); _i32(masks, "masks")
    P, ld = points.shape
    keys = torch.empty(max(n_obj, 1) * 6, dtype=torch.int32, device=points.device)
    boxes = torch.empty(n_obj, 6, dtype=torch.float32, device=points.device)
    _call("pn2_prep_object_boxes", points, P, ld, n_obj, float(padding), _ptr(points), _ptr(masks), _ptr(keys), _ptr(boxes),
          alg_bytes=P * (12 + 4))
    return boxes


def prep_gather_normalise(points, masks, edges, sel, n_obj, t_obj, t_rel):
    """Selected scan indices `sel` (n_obj*t_obj + E*t_rel) i32 -> (obj (n_obj, t_obj, ld), rel (E, t_rel, ld+1)): gather,
    mask channel, zero_mean (pn2_prep_gather_normalise)."""
    _f32(points, "points"); _i32(masks, "masks"); _i32(edges, "edges"); _i32(sel, "sel")
    _same_device((points, "points"), (masks, "masks"), (edges, "edges"), (sel, "sel"))
    P, ld = points.shape
    E = edges.size(1)
    if sel.numel() != n_obj * int(t_obj) + E * int(t_rel):
        raise RuntimeError("prep_gather_normalise: sel must hold n_obj * t_obj + E * t_rel indices")
    obj = torch.empty(n_obj, t_obj, ld, dtype=torch.float32, device=points.device)
    rel = torch.empty(E, t_rel, ld + 1, dtype=torch.float32, device=points.device)
    _call("pn2_prep_gather_normalise", points, ld, n_obj, E, int(t_obj), int(t_rel), _ptr(points), _ptr(masks), _ptr(edges),
          _ptr(sel), _ptr(obj), _ptr(rel), alg_bytes=sel.numel() * (4 + 8 * ld))
    return obj, rel


# ------------------------------------------- (f)4: Graphormer pre-processing (role_prediction/graphormer/algos.pyx)
def floyd_warshall(adjacency):
    """adjacency (B, n, n) int64 -> (dist (B,n,n), path (B,n,n)) int64; algos.pyx:11-54."""
    _i64(adjacency, "adjacency")
    _same_device((adjacency, "adjacency"))
    B, n, _ = adjacency.shape
    dist, path = torch.empty_like(adjacency), torch.empty_like(adjacency)
    _call("pn2_floyd_warshall", adjacency, B, n, _ptr(adjacency), _ptr(dist), _ptr(path), alg_bytes=24 * B * n * n)
    return dist, path


def gen_edge_input(max_dist, path, edge_feat):
    """path (B,n,n), edge_feat (B,n,n,F) int64 -> (B,n,n,max_dist,F) int64, -1 where there is no edge; algos.pyx:62-89."""
    _i64(path, "path"); _i64(edge_feat, "edge_feat")
    _same_device((path, "path"), (edge_feat, "edge_feat"))
    B, n, _ = path.shape
    F = edge_feat.size(-1)
    out = torch.full((B, n, n, int(max_dist), F), -1, dtype=torch.int64, device=path.device)
    _call("pn2_gen_edge_input", path, B, n, int(max_dist), F, _ptr(path), _ptr(edge_feat), _ptr(out),
          alg_bytes=8 * B * n * n * (1 + F + int(max_dist) * F))
    return out
