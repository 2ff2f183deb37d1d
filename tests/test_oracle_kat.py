"""Known-answer tests that pin the CPU oracle: hand-derived from
the reference kernels' semantics because the reference ships no golden vectors."""
import numpy as np
import torch

from oracle.oracle import OracleExt, opt_n_threads


def T(a, dtype=torch.float32):
    return torch.tensor(a, dtype=dtype)


def test_opt_n_threads_matches_reference_formula():
    # EXT/include/cuda_utils.h:15-19: clamp(2^floor(log2 n), 1, 512) via a double log ratio
    assert [opt_n_threads(n) for n in (1, 2, 3, 4, 7, 8, 255, 256, 511, 512, 513, 4000, 50000)] == \
        [1, 2, 2, 4, 4, 8, 128, 256, 256, 512, 512, 512, 512]


def test_fps_known_answer():
    p = T([[[1, 0, 0], [2, 0, 0], [4, 0, 0], [8, 0, 0]]])
    assert OracleExt.furthest_point_sampling(p, 3).tolist() == [[0, 3, 2]]


def test_fps_starts_at_zero_and_skips_near_origin_points():
    # point 0 may be near the origin: it still starts the sequence (sampling_gpu.cu:86-87);
    # points with |p|^2 <= 1e-3 are never *selected* (:100-101)
    p = T([[[0.01, 0, 0], [0.02, 0.01, 0], [1, 0, 0], [0, 1, 0], [0.03, 0, 0.005]]])
    sel = OracleExt.furthest_point_sampling(p, 5).tolist()[0]
    assert sel[0] == 0
    assert set(sel[1:]) <= {2, 3}          # only the two far points qualify; then they repeat
    assert sel[1] == 3 and sel[2] == 2      # d(0,3)=1.0001 > d(0,2)=0.9801


def test_fps_all_points_skipped_returns_zero():
    p = torch.full((2, 7, 3), 0.001)
    assert OracleExt.furthest_point_sampling(p, 4).tolist() == [[0, 0, 0, 0]] * 2


def test_fps_tie_break_is_bit_reversed_tid_then_k():
    # N = 1024 -> block 512.  Put exact duplicates of the far point at k = 1 (tid 1,
    # bitrev9 = 256), k = 2 (tid 2, bitrev9 = 128), k = 256 (tid 256, bitrev9 = 1)
    # and k = 768 (tid 256 again, larger k).  Winner: smallest bitrev -> tid 256,
    # then smaller k -> 256.
    N = 1024
    p = torch.zeros(1, N, 3)
    p[0, :, 0] = 0.5
    p[0, 0] = T([1.0, 0, 0])
    for k in (1, 2, 256, 768):
        p[0, k] = T([-1.0, 0.5, 0])
    assert OracleExt.furthest_point_sampling(p, 2).tolist() == [[0, 256]]
    # without the tid-256 copies the winner is tid 2 (bitrev 128) over tid 1 (bitrev 256)
    p[0, 256] = T([0.5, 0, 0]); p[0, 768] = T([0.5, 0, 0])
    assert OracleExt.furthest_point_sampling(p, 2).tolist() == [[0, 2]]


def test_fps_m_larger_than_distinct_points_repeats_by_rule():
    p = T([[[1, 0, 0], [2, 0, 0], [3, 0, 0]]])
    sel = OracleExt.furthest_point_sampling(p, 6).tolist()[0]
    assert sel[:3] == [0, 2, 1]
    # all running distances are now 0: ties -> block size 2 (N=3): tid = k%2, bitrev1(tid)=tid;
    # tid 0 holds k=0,2 -> smallest k = 0
    assert sel[3:] == [0, 0, 0]


def test_ball_query_padding_empty_and_strictness():
    xyz = torch.zeros(1, 12, 3)
    xyz[0, :, 0] = 10.0                     # far away
    xyz[0, 5] = T([0.1, 0, 0])
    xyz[0, 9] = T([0, 0.1, 0])
    xyz[0, 11] = T([0.5, 0, 0])             # exactly on the r = 0.5 sphere: d2 == r2 -> excluded
    new_xyz = T([[[0, 0, 0], [100, 100, 100]]])
    idx = OracleExt.ball_query(new_xyz, xyz, 0.5, 4)
    assert idx[0, 0].tolist() == [5, 9, 5, 5]
    assert idx[0, 1].tolist() == [0, 0, 0, 0]


def test_ball_query_takes_first_nsample_in_index_order():
    xyz = torch.zeros(1, 10, 3)
    idx = OracleExt.ball_query(torch.zeros(1, 1, 3), xyz, 1.0, 3)
    assert idx[0, 0].tolist() == [0, 1, 2]


def test_three_nn_ties_keep_earliest_index_and_short_known():
    known = T([[[1, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0]]])
    unknown = T([[[0, 0, 0]]])
    d2, idx = OracleExt.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [0, 1, 2]
    assert d2[0, 0].tolist() == [1.0, 1.0, 1.0]
    d2, idx = OracleExt.three_nn(unknown, known[:, :2].contiguous())
    assert idx[0, 0].tolist() == [0, 1, 0] and torch.isinf(d2[0, 0, 2])


def test_three_interpolate_reference_gradcheck_case():
    # the only numeric test in the reference tree: GF3D/pointnet2/pointnet2_test.py:18-30
    feats = T([[[1.0, 2.0, 3.0, 4.0], [-1.0, 0.5, 2.0, 8.0]]])
    idx = T([[[0, 1, 2], [1, 2, 3]]], torch.int32)
    w = T([[[1, 1, 1], [2, 2, 2]]])
    out = OracleExt.three_interpolate(feats, idx, w)
    assert out.tolist() == [[[6.0, 18.0], [1.5, 21.0]]]
    g = OracleExt.three_interpolate_grad(torch.ones(1, 2, 2), idx, w, 4)
    assert g.tolist() == [[[1.0, 3.0, 3.0, 2.0]] * 2]


def test_group_and_gather_and_grads():
    pts = torch.arange(2 * 3 * 5, dtype=torch.float32).view(2, 3, 5)
    idx = T([[[0, 4], [2, 2]], [[1, 1], [3, 0]]], torch.int32)
    out = OracleExt.group_points(pts, idx)
    assert out.shape == (2, 3, 2, 2)
    assert out[1, 2].tolist() == [[26.0, 26.0], [28.0, 25.0]]
    g = OracleExt.group_points_grad(torch.ones(2, 3, 2, 2), idx, 5)
    assert g[0, 0].tolist() == [1.0, 0.0, 2.0, 0.0, 1.0]
    gi = T([[0, 0, 3], [4, 1, 1]], torch.int32)
    assert OracleExt.gather_points(pts, gi)[1, 0].tolist() == [19.0, 16.0, 16.0]
    assert OracleExt.gather_points_grad(torch.ones(2, 3, 3), gi, 5)[0, 1].tolist() == [2.0, 0, 0, 1.0, 0]


def test_gcn_lift_and_scatter_follow_pyg_source_to_target():
    # scene_graph_helpers/model/pointnets/network_util.py:86-94 demo: edge_index=[[0,1,2],[2,1,0]]
    # => x_i = x[edge_index[1]] = x[[2,1,0]], x_j = x[edge_index[0]] = x[[0,1,2]]
    x = T([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]])
    ei = T([[0, 1, 2], [2, 1, 0]], torch.int64)
    assert OracleExt.gather_rows(x, ei[1].contiguous()).tolist() == [[3.0, 30.0], [2.0, 20.0], [1.0, 10.0]]
    assert OracleExt.gather_rows(x, ei[0].contiguous()).tolist() == x.tolist()
    msg = T([[1.0, 1.0], [2.0, 2.0], [4.0, 4.0]])
    agg = OracleExt.scatter_add_rows(msg, T([2, 2, 0], torch.int64), 3)
    assert agg.tolist() == [[4.0, 4.0], [0.0, 0.0], [3.0, 3.0]]
