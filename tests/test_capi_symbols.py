"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/pn2_hip.h declares (no compute calls here: there is no GPU)."""
import ctypes
import os
import re
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "pn2_hip.h")
LIB = os.path.join(REPO, "4d-or_amd", "libpn2_hip.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_nine_reference_ops():
    syms = declared_symbols()
    for op in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
               "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
               "group_points_grad"):                      # EXT/src/bindings.cpp:6-19
        assert f"pn2_{op}" in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.pn2_abi_version.restype = ctypes.c_int
    assert lib.pn2_abi_version() == 11
    lib.pn2_strerror.restype = ctypes.c_char_p
    lib.pn2_strerror.argtypes = [ctypes.c_int]
    assert lib.pn2_strerror(-1) and lib.pn2_strerror(12345)


def test_python_binding_covers_every_symbol_and_refuses_cpu_tensors():
    import torch
    from pointnet2_ops import _ext
    assert sorted(_ext.EXPORTED_SYMBOLS) == declared_symbols()
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(torch.zeros(1, 8, 3), 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.ball_query(torch.zeros(1, 3, 2).transpose(1, 2), torch.zeros(1, 4, 3), 0.1, 2)
    with pytest.raises(RuntimeError, match="int tensor"):
        _ext.gather_points(torch.zeros(1, 3, 4), torch.zeros(1, 2, dtype=torch.int64))


def test_ctypes_signatures_have_the_header_arity():
    """Every ctypes argtypes list (incl. the trailing stream) has as many entries as the C prototype has parameters: a
    missing entry makes ctypes pass the 64-bit stream handle as a 32-bit int (hipErrorInvalidValue at launch)."""
    from pointnet2_ops import _ext
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(pn2_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(1), m.group(2)
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        if name in _ext._SIGNATURES:
            assert len(_ext._SIGNATURES[name]) == n, name
            seen += 1
    assert seen == len(_ext._SIGNATURES)


def test_host_side_routing_predicates():
    """The shape predicates are plain host functions: which ball-query algorithm / which fused kernels a shape takes, and
    the workspace each algorithm asks for (include/pn2_hip.h: PN2_BQ_SCAN 0 | _CELLS 1 | _SLABS 2)."""
    lib = ctypes.CDLL(LIB)
    f32, i32, sz = ctypes.c_float, ctypes.c_int, ctypes.c_size_t
    lib.pn2_ball_query_auto.argtypes = [i32, i32, i32, f32, i32]
    lib.pn2_ball_query_auto.restype = i32
    lib.pn2_ball_query_algo_bytes.argtypes = [i32, i32, i32, i32, f32, i32]
    lib.pn2_ball_query_algo_bytes.restype = sz
    lib.pn2_ball_query_workspace_bytes.argtypes = [i32, i32, i32, f32, i32]
    lib.pn2_ball_query_workspace_bytes.restype = sz
    lib.pn2_ball_query_grid_bytes.argtypes = [i32, i32, i32]
    lib.pn2_ball_query_grid_bytes.restype = sz
    auto = lib.pn2_ball_query_auto
    # headline SA1: a scan would walk 50000 * 64 / 400 = 8000 points per centre -> slabs; SA2-SA4: short walks -> scan
    assert auto(32, 50000, 2048, 0.2, 64) == 2
    assert auto(32, 2048, 1024, 0.4, 32) == 0 and auto(32, 1024, 512, 0.8, 16) == 0 and auto(32, 512, 256, 1.2, 16) == 0
    # scene-graph encoders: sparse balls in 4000 / 8000-point clouds never exit early -> slabs; very crowded -> scan
    assert auto(72, 8000, 512, 0.1, 16) == 2 and auto(288, 4000, 512, 0.2, 32) == 2
    assert auto(72, 8000, 512, 0.4, 128) == 0
    # radii the cell edge cannot be sized from, empty problems
    for r in (0.0, -1.0, float("inf"), float("nan")):
        assert auto(4, 50000, 512, r, 32) == 0 and lib.pn2_ball_query_algo_bytes(2, 4, 50000, 512, r, 32) == 0
    assert auto(0, 50000, 512, 0.2, 32) == 0
    # workspaces: the scan has none; slabs = 16 B per point + a 16 KB table (+ 64 B) per slab of 2048 indices
    assert lib.pn2_ball_query_algo_bytes(0, 32, 50000, 2048, 0.2, 64) == 0
    assert lib.pn2_ball_query_algo_bytes(2, 32, 50000, 2048, 0.2, 64) == 32 * 50000 * 16 + 32 * 25 * (4096 + 16) * 4 + 256
    assert lib.pn2_ball_query_workspace_bytes(32, 50000, 2048, 0.2, 64) == lib.pn2_ball_query_algo_bytes(2, 32, 50000, 2048, 0.2, 64)
    assert lib.pn2_ball_query_algo_bytes(1, 32, 50000, 2048, 0.2, 64) == lib.pn2_ball_query_grid_bytes(32, 50000, 64) > 0
    assert lib.pn2_ball_query_algo_bytes(1, 32, 50000, 2048, 0.2, 300) == 0          # cell list: nsample <= 256
    # fused shared-MLP predicates
    lib.pn2_mlp_gemm_first_supported.argtypes = [i32, i32, i32]
    lib.pn2_mlp_bwd_fused_fold_supported.argtypes = [i32, i32, i32]
    lib.pn2_pool_bwd_supported.argtypes = [i32, i32, i32]
    assert lib.pn2_mlp_gemm_first_supported(6, 64, 64) and lib.pn2_mlp_gemm_first_supported(8, 128, 128)
    assert not lib.pn2_mlp_gemm_first_supported(9, 64, 64) and not lib.pn2_mlp_gemm_first_supported(6, 256, 64)
    assert lib.pn2_mlp_bwd_fused_fold_supported(64, 64, 6) and not lib.pn2_mlp_bwd_fused_fold_supported(128, 64, 6)
    assert lib.pn2_pool_bwd_supported(128, 64, 64) and lib.pn2_pool_bwd_supported(256, 128, 32)
    assert not lib.pn2_pool_bwd_supported(256, 128, 16)                               # SA3 / SA4: materialised path


def test_round4_host_side_fps_and_gcn():
    """Host functions of round 4, no GPU: the cluster FPS's hand-off area (several samples per hand-off: two parities x six
    fields x 64 sub-blobs of 8-byte granules per cloud) sits in front of the status word whatever the plan switches say; the
    measurement switch is scoped; the fused TripletGCN predicate (dimensions multiples of 32, scans of <= 128 rows)."""
    lib = ctypes.CDLL(LIB)
    i32, sz = ctypes.c_int, ctypes.c_size_t
    lib.pn2_fps_workspace_bytes.argtypes = [i32, i32, i32]
    lib.pn2_fps_workspace_bytes.restype = sz
    lib.pn2_fps_status_offset.argtypes = [i32, i32, i32]
    lib.pn2_fps_status_offset.restype = ctypes.c_longlong
    area = 2 * 6 * 64 * 8
    for B, N in ((32, 50000), (8, 20000), (16, 100000)):
        off = lib.pn2_fps_status_offset(B, N, 2048)
        assert off == B * area
        need = lib.pn2_fps_workspace_bytes(B, N, 2048)
        # hand-off area | status | B N floats (streaming fallback) | B N binned 16-byte records
        assert need >= B * area + 256 + B * N * 4 + B * N * 16
        for on in (0, 1):                       # the switches change the plan, never what a caller has to bring
            prev = lib.pn2_fps_get_multi()
            lib.pn2_fps_set_multi(on)
            try:
                assert lib.pn2_fps_workspace_bytes(B, N, 2048) == need and lib.pn2_fps_status_offset(B, N, 2048) == off
            finally:
                lib.pn2_fps_set_multi(prev)
    assert lib.pn2_fps_status_offset(32, 2048, 1024) == -1 and lib.pn2_fps_workspace_bytes(32, 2048, 1024) == 0   # one workgroup per cloud
    assert lib.pn2_fps_set_plan_override(5, 0, 1, 0, 0) == 0 and lib.pn2_fps_set_plan_override(6, 0, 0, 0, 0) != 0
    assert lib.pn2_fps_set_plan_override(-1, 0, 0, 0, 0) == 0
    lib.pn2_gcn_fused_supported.argtypes = [i32, i32, i32, i32]
    ok = lib.pn2_gcn_fused_supported
    assert ok(256, 256, 512, 110) and ok(256, 256, 512, 128) and ok(64, 64, 96, 2)
    assert not ok(256, 256, 512, 129) and not ok(250, 256, 512, 72) and not ok(256, 256, 16, 72)


def test_triplet_gcn_on_cpu_tensors_takes_the_unfused_path():
    """The fused per-scan layer is a GPU route: CPU tensors (the oracle backend of the parity tests) never reach it, and the
    ReLU between layers is applied by the layer itself either way (network_TripletGCN.py:76-78)."""
    import torch
    import oracle_ext
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    saved = gcn._ext
    gcn._ext = oracle_ext.OracleRowsExt
    try:
        torch.manual_seed(0)
        layer = gcn.TripletGCN(32, 32, 64).train()
        n = 4
        ei = torch.tensor([[a, b] for a in range(n) for b in range(n) if a != b]).t().contiguous()
        x, e = torch.randn(n, 32), torch.randn(n * (n - 1), 32)
        assert not layer._fused_ok(x, e, None)
        a_x, a_e = layer(x, e, ei)
        b_x, b_e = layer(x, e, ei, relu_out=True)
        assert torch.equal(b_x, a_x.relu()) and torch.equal(b_e, a_e.relu())
    finally:
        gcn._ext = saved


def test_segment_table_host_side():
    """The host side of the batched-scans entry points (include/pn2_hip.h "batched scans"): row offsets of the scans, the
    cache per (device, rows) signature, the routing predicate (bf16 node, pooled stack, every BatchNorm in training mode),
    and the argument checks of the launchers that need no GPU (null table, scan count, longest scan beyond M)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "4d-or_amd"))
    from pointnet2_ops import _ext as e
    from pointnet2_ops import fused_mlp
    from pointnet2_ops.pointnet2_modules import build_shared_mlp
    t = e.SegTable.get(torch.device("cpu"), [48, 0, 160, 16])
    assert t.ptr.tolist() == [0, 48, 48, 208, 224] and t.ptr.dtype == torch.int64
    assert (t.nseg, t.max_rows, t.total) == (4, 160, 224)
    assert e.SegTable.get(torch.device("cpu"), (48, 0, 160, 16)) is t
    mlp = build_shared_mlp([6, 64, 128], bn=True).train()
    layers = fused_mlp.parse_stack(mlp)
    assert fused_mlp.seg_table_ok(layers, 16, fused_mlp._FusedMLPBf16)
    assert not fused_mlp.seg_table_ok(layers, 16, fused_mlp._FusedMLP)          # fp32 stacks loop over the scans
    assert not fused_mlp.seg_table_ok(layers, 0, fused_mlp._FusedMLPBf16)       # un-pooled stack
    mlp.eval()
    assert not fused_mlp.seg_table_ok(fused_mlp.parse_stack(mlp), 16, fused_mlp._FusedMLPBf16)   # running statistics
    lib = ctypes.CDLL(LIB)
    ll, i32, vp = ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p
    lib.pn2_mlp_gemm_bf16_seg.argtypes = [ll] + [i32] * 8 + [vp] * 5 + [i32] + [vp] * 2 + [i32] + [vp] * 6 + [i32, ll, vp]
    base = [128, 64, 64, 0, 0, 0, 0, 64, 64] + [None] * 5 + [0, None, None, 0, None, None, None, None, None]
    assert lib.pn2_mlp_gemm_bf16_seg(*base, None, 2, 64, None) != 0                       # no table
    dummy = ctypes.cast(ctypes.create_string_buffer(64), vp)
    assert lib.pn2_mlp_gemm_bf16_seg(*base, dummy, 0, 64, None) != 0                      # no scans
    assert lib.pn2_mlp_gemm_bf16_seg(*base, dummy, 2, 129, None) != 0                     # longest scan beyond M
    lib.pn2_bn_finalize_seg.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp]
    assert lib.pn2_bn_finalize_seg(0, 64, dummy, dummy, None, None, 1e-5, 0.1, None, None, None, dummy, None) != 0
    assert lib.pn2_bn_finalize_seg(2, 64, dummy, dummy, None, None, 1e-5, 0.1, dummy, None, None, dummy, None) != 0  # mean w/o var


def test_gcn_layer_struct_mirrors_the_header_and_argument_checks_need_no_gpu():
    """pn2_gcn_layer (include/pn2_hip.h) against the ctypes mirror in _ext: same size, fields in the header's order; the
    entry points reject a null layer / unsupported dimensions and accept an empty layer before any launch; the backward's
    workspace holds its seven intermediate matrices in 256-byte pieces."""
    import re
    sys.path.insert(0, os.path.join(REPO, "4d-or_amd"))
    from pointnet2_ops import _ext
    header = open(os.path.join(REPO, "include", "pn2_hip.h")).read()
    body = re.search(r"typedef struct pn2_gcn_layer \{(.*?)\} pn2_gcn_layer;", header, re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(long long|int|float|void)\s*", "", decl)
        names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
    assert names == [f[0] for f in _ext.GcnLayer._fields_]
    assert ctypes.sizeof(_ext.GcnLayer) == 48 + 56 * ctypes.sizeof(ctypes.c_void_p)
    lib = ctypes.CDLL(LIB)
    for fn in (lib.pn2_gcn_layer_forward, lib.pn2_gcn_layer_backward):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert fn(None, None) == -2                                   # PN2_ENULL
        empty = _ext.GcnLayer(0, 0, 0, 256, 256, 512, 0)
        assert fn(ctypes.byref(empty), None) == 0                     # nothing to do: no pointer is looked at
        bad = _ext.GcnLayer(9, 72, 1, 250, 256, 512, 0)
        assert fn(ctypes.byref(bad), None) == -1                      # PN2_EINVAL: dn not a multiple of 32
        nul = _ext.GcnLayer(9, 72, 1, 256, 256, 512, 0)
        assert fn(ctypes.byref(nul), None) == -2                      # required pointers missing
    ws = lib.pn2_gcn_layer_backward_workspace_bytes
    ws.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    ws.restype = ctypes.c_size_t
    n, e, dn, de, dh = 9, 72, 256, 256, 512
    assert ws(n, e, dn, de, dh) == 4 * (n * dn + 3 * n * dh + e * (2 * dh + de) + 2 * e * dh)      # every piece already a multiple of 64 floats
    assert ws(3, 5, 32, 32, 32) == 4 * (128 + 3 * 128 + 512 + 2 * 192)                               # 96 -> 128, 480 -> 512, 160 -> 192 floats
    assert ws(-1, 5, 32, 32, 32) == 0
