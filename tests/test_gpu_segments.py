"""Per-scan BatchNorm statistics through the kernels' segment tables (csrc: pn2_*_seg, include/pn2_hip.h "batched scans"):
one launch per kernel for all scans of a block-diagonal batch == one call per scan — the arithmetic of the reference's
DataLoader(batch_size=1) steps (scene_graph_prediction/main.py:54-56)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mlp(widths, seed):
    from pointnet2_ops.pointnet2_modules import build_shared_mlp
    torch.manual_seed(seed)
    m = build_shared_mlp(widths, bn=True).cuda().train()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)
    return m


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("S,K,N,pro,epi", [(5, 64, 64, 1, 1), (3, 200, 128, 0, 1), (4, 64, 128, 1, 0), (7, 128, 256, 1, 1)])
def test_segmented_forward_gemm_equals_one_call_per_scan(S, K, N, pro, epi):
    """pn2_mlp_gemm_bf16_seg: every scan's output rows are BIT-equal to its own call (tiles start at the scan's first row),
    its statistics equal up to the order of the fp64 atomics (the kernel's virtual workgroups keep the fp32 partial sums of
    the scan's own call); finalize blocks and the pooled rows follow."""
    from pointnet2_ops import _ext as e
    g = torch.Generator().manual_seed(S * 1000 + K + N)
    ns = 16
    rows = [int(torch.randint(1, 40, (1,), generator=g)) * ns * 8 + (ns if s % 2 else 0) for s in range(S)]
    rows[1] = ns                                                   # a scan smaller than one row tile
    M = sum(rows)
    X = torch.randn(M, K, generator=g).to(BF).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    finp = torch.stack([torch.stack([torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5,
                                     torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3]) for _ in range(S)]
                       ).cuda().contiguous()
    seg = e.SegTable.get(X.device, rows)
    stats = torch.zeros(S, 2, N, dtype=torch.float64, device="cuda") if epi else None
    p = (finp[:, 2], finp[:, 3]) if pro else None
    Y = e.mlp_gemm_bf16(X, W, pro=pro, epi=epi, p=p, stats=stats, seg=seg)
    gamma, beta = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.2).cuda()
    r0 = 0
    fins = []
    for s, r in enumerate(rows):
        st = torch.zeros(2, N, dtype=torch.float64, device="cuda") if epi else None
        Ys = e.mlp_gemm_bf16(X[r0:r0 + r], W, pro=pro, epi=epi, p=None if not pro else (finp[s, 2], finp[s, 3]), stats=st)
        assert torch.equal(Y[r0:r0 + r], Ys), f"scan {s}"
        if epi:
            torch.testing.assert_close(stats[s], st, rtol=1e-13, atol=1e-11 * r)   # same fp32 partials, fp64 order only
            fins.append(e.bn_finalize(st, r, gamma, beta, 1e-5, 0.0, None, None))
        r0 += r
    if epi:
        rm, rv, nbt = torch.randn(N, device="cuda"), torch.rand(N, device="cuda") + 0.5, torch.zeros((), dtype=torch.int64, device="cuda")
        rm2, rv2, nbt2 = rm.clone(), rv.clone(), nbt.clone()
        for s, r in enumerate(rows):      # S calls of the single-scan finalize, in scan order
            st_s = stats[s].contiguous()
            e.bn_finalize(st_s, r, gamma, beta, 1e-5, 0.1, rm2, rv2, nbt2)
        fin = e.bn_finalize_seg(stats, seg, gamma, beta, 1e-5, 0.1, rm, rv, nbt)
        assert torch.equal(rm, rm2) and torch.equal(rv, rv2) and int(nbt) == int(nbt2) == S
        torch.testing.assert_close(fin, torch.stack(fins), rtol=1e-5, atol=1e-6)
        fin_exact = torch.stack(fins).contiguous()
        out, arg, yraw = e.bn_relu_rows_max_bf16(Y, fin_exact, ns, seg=seg)
        r0 = 0
        for s, r in enumerate(rows):
            o, a, yr = e.bn_relu_rows_max_bf16(Y[r0:r0 + r], fins[s], ns)
            g0, g1 = r0 // ns, (r0 + r) // ns
            assert torch.equal(out[g0:g1], o) and torch.equal(arg[g0:g1], a) and torch.equal(yraw[g0:g1], yr)
            r0 += r


@pytest.mark.parametrize("S,N,K,pooled", [(4, 64, 64, True), (3, 128, 64, True), (5, 128, 128, False), (3, 256, 256, True),
                                          (2, 256, 264, False)])
def test_segmented_backward_kernels_equal_one_call_per_scan(S, N, K, pooled):
    """pool_bwd_prep / bn_bwd_consts / one-pass backward (or wgrad + dgrad GEMM where the one-pass kernel has no instance)
    with a segment table == the same calls per scan; weight gradients = the sum over the scans."""
    from pointnet2_ops import _ext as e
    g = torch.Generator().manual_seed(S * 77 + N + K)
    ns = 32
    rows = [int(torch.randint(2, 30, (1,), generator=g)) * ns * 4 for _ in range(S)]
    M = sum(rows)
    seg = e.SegTable.get(torch.device("cuda", 0), rows)
    y = torch.randn(M, N, generator=g).to(BF).cuda()
    yprev = torch.randn(M, K, generator=g).to(BF).cuda()
    W = (torch.randn(N, K, generator=g) / N ** 0.5).cuda()
    mk = lambda C: torch.stack([torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5,
                                             torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3])
                                for _ in range(S)]).cuda().contiguous()
    fin_l, fin_p = mk(N), mk(K)
    gamma = (torch.rand(N, generator=g) + 0.5).cuda()
    if pooled:
        R = M // ns
        pooledv = torch.randn(R, N, generator=g).cuda()
        yraw = torch.randn(R, N, generator=g).cuda()
        gout = torch.randn(R, N, generator=g).cuda()
        arg = torch.randint(0, ns, (R, N), generator=g, dtype=torch.int32).cuda()
        gPm, sums = e.pool_bwd_prep(yraw, pooledv, gout, fin_l, seg=seg, ns=ns)
        gmode, G = e.PRO_POOLG, None
    else:
        G = torch.randn(M, N, generator=g).to(BF).cuda()
        sums = torch.randn(S, 2, N, generator=g, dtype=torch.float64).cuda()
        gmode, arg, gPm = e.PRO_GY, None, None
    consts, dgamma, dbeta, Wt = e.bn_bwd_consts_seg(sums, seg, gamma, fin_l, True, W=W, k0=0)
    one_pass = e.mlp_bwd_bf16_supported(N, K)
    if one_pass:
        Gout, s_in, dW = e.mlp_bwd_bf16(y, consts, Wt, yprev, fin_p, gmode, G=G, arg=arg, gP=gPm, ns=ns if pooled else 0, seg=seg)
    else:
        dW = e.mlp_wgrad_bf16(y, consts, yprev, gmode, e.PRO_BNRELU, K, G=G, arg=arg, gP=gPm, ns=ns if pooled else 0,
                              a_fin=fin_p, seg=seg)
        s_in = torch.zeros(S, 2, K, dtype=torch.float64, device="cuda")
        Gout = e.mlp_gemm_bf16(G, Wt, pro=gmode, epi=e.EPI_MASK, X2=y, p=(consts[:, 0], consts[:, 1], consts[:, 2]), arg=arg,
                               gP=gPm, ns=ns if pooled else 0, stats=s_in, Yprev=yprev, e_fin=fin_p, M=M, seg=seg)
    dW_sum = torch.zeros_like(dW)
    dg_sum, db_sum = torch.zeros_like(dgamma), torch.zeros_like(dbeta)
    r0 = 0
    for s, r in enumerate(rows):
        sl = slice(r0, r0 + r)
        gl = slice(r0 // ns, (r0 + r) // ns)
        if pooled:
            gPm_s, sums_s = e.pool_bwd_prep(yraw[gl], pooledv[gl], gout[gl], fin_l[s])
            assert torch.equal(gPm[gl], gPm_s)
            torch.testing.assert_close(sums[s], sums_s, rtol=1e-13, atol=1e-11 * r)
            a_s, G_s = arg[gl], None
        else:
            sums_s, gPm_s, a_s, G_s = sums[s].contiguous(), None, None, G[sl]
        c_s, dg_s, db_s, Wt_s = e.bn_bwd_consts(sums[s].contiguous(), r, gamma, fin_l[s], True, W=W, k0=0)
        torch.testing.assert_close(consts[s], c_s, rtol=1e-6, atol=1e-7)
        assert torch.equal(Wt, Wt_s)
        dg_sum += dg_s
        db_sum += db_s
        cs = consts[s].contiguous()
        if one_pass:
            Go, si, dWs = e.mlp_bwd_bf16(y[sl], cs, Wt, yprev[sl], fin_p[s], gmode, G=G_s, arg=a_s, gP=gPm_s, ns=ns if pooled else 0)
        else:
            dWs = e.mlp_wgrad_bf16(y[sl], cs, yprev[sl], gmode, e.PRO_BNRELU, K, G=G_s, arg=a_s, gP=gPm_s, ns=ns if pooled else 0,
                                   a_fin=fin_p[s])
            si = torch.zeros(2, K, dtype=torch.float64, device="cuda")
            Go = e.mlp_gemm_bf16(G_s, Wt, pro=gmode, epi=e.EPI_MASK, X2=y[sl], p=(cs[0], cs[1], cs[2]), arg=a_s, gP=gPm_s,
                                 ns=ns if pooled else 0, stats=si, Yprev=yprev[sl], e_fin=fin_p[s], M=r)
        assert torch.equal(Gout[sl], Go), f"scan {s}"
        torch.testing.assert_close(s_in[s], si, rtol=1e-13, atol=1e-11 * r)
        dW_sum += dWs
        r0 += r
    assert _rel(dW, dW_sum) <= 1e-5
    torch.testing.assert_close(dgamma, dg_sum, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dbeta, db_sum, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", ["grouped_xyz_rgb", "grouped_features", "group_all"])
def test_stack_with_segment_table_equals_the_per_scan_loop(case):
    """The autograd nodes: fused_group_mlp_pool(clouds_per_scan=...) / fused_shared_mlp(rows_per_scan=...) on the bf16 path
    through the segment tables == the per-scan loop (fused_mlp.SEG_TABLE = False): pooled outputs, every gradient, the
    running statistics after the S momentum updates."""
    from pointnet2_ops import _ext as e
    from pointnet2_ops import fused_mlp
    from pointnet2_ops import pointnet2_modules as pm
    from pointnet2_ops import pointnet2_utils as pu
    prev = fused_mlp.set_mlp_dtype("bf16")
    try:
        g = torch.Generator().manual_seed(11)
        sizes = [3, 5, 1, 4]
        B = sum(sizes)
        if case == "group_all":
            N, C = 128, 256
            mlp = _mlp([C + 3, 256, 256], 5)
            grouper = pu.GroupAll(use_xyz=True)
            new_xyz = None
        else:
            N, m, ns = 1500, 64, 16
            C = 3 if case == "grouped_xyz_rgb" else 61
            mlp = _mlp([C + 3, 64, 128], 6)
            grouper = pu.QueryAndGroup(0.3, ns, use_xyz=True)
        xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
        feats = torch.randn(B, N, C, generator=g).cuda()
        if case != "group_all":
            new_xyz = xyz[:, :m].contiguous()
        results = []
        for table in (True, False):
            fused_mlp.SEG_TABLE = table
            mm = copy.deepcopy(mlp)
            f = feats.clone().requires_grad_(case != "grouped_xyz_rgb")
            with pm.per_scan_statistics(sizes):
                out = pm.sa_scale_rows(grouper, mm, xyz, new_xyz, f)
            w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()
            (out * w).sum().backward()
            results.append((out.detach(), None if f.grad is None else f.grad, [p.grad for p in mm.parameters()],
                            [b.clone() for b in mm.buffers()]))
        (o1, g1, p1, b1), (o2, g2, p2, b2) = results
        assert _rel(o1, o2) <= 1e-6
        if g1 is not None:
            assert _rel(g1, g2) <= 1e-5
        for a, b in zip(p1, p2):
            assert _rel(a, b) <= 1e-5            # (fp32 atomics order in the weight gradients)
        for a, b in zip(b1, b2):
            if a.dtype == torch.int64:
                assert int(a) == int(b) == len(sizes)
            else:
                torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    finally:
        fused_mlp.SEG_TABLE = True
        fused_mlp.set_mlp_dtype(prev)


def test_head_batch_norm_per_scan_running_statistics_equal_sequential_batch_norm_calls():
    """scan_batch_norm on the GPU (pn2_segment_bn_rows + pn2_segment_bn_running_update) == S calls of
    F.batch_norm(training=True) on the scans' rows, in order: outputs, running mean / variance, num_batches_tracked."""
    import torch.nn.functional as F
    from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet import scan_batch_norm
    g = torch.Generator().manual_seed(5)
    rows = [9, 4, 72, 2, 30]
    C = 96
    x = (torch.randn(sum(rows), C, generator=g) * 2 + 0.5).cuda()
    ptr = torch.tensor([0] + list(torch.tensor(rows).cumsum(0)), dtype=torch.int64, device="cuda")
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(bn)
    y = scan_batch_norm(bn, x, ptr)
    r0, parts = 0, []
    for r in rows:
        parts.append(F.batch_norm(x[r0:r0 + r], ref.running_mean, ref.running_var, ref.weight, ref.bias, True, ref.momentum, ref.eps))
        ref.num_batches_tracked += 1
        r0 += r
    torch.testing.assert_close(y, torch.cat(parts), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=2e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == len(rows)


@pytest.mark.parametrize("bf16", [True, False])
def test_lifted_first_layer_with_a_segment_table_equals_single_scan_launches(bf16):
    """pn2_group_lift_rows_seg / _grad_seg (grid.y = scan): rows, per-scan column sums, per-point sums and the (3 N0 + 9)
    accumulator rows are BIT-EQUAL to one pn2_group_lift_rows[_bf16] / _grad launch per scan on that scan's clouds — scans of
    different sizes (the largest decides the grid, the others' surplus workgroups exit), sparse and heavy points."""
    from pointnet2_ops import _ext as e
    g = torch.Generator().manual_seed(21)
    clouds, N, m, ns, C, N0, r = [3, 7, 1, 5], 700, 96, 32, 64, 128, 0.45
    B = sum(clouds)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    new_xyz = xyz[:, :m].contiguous()
    idx = e.ball_query(new_xyz, xyz, r, ns)
    P = torch.randn(B, N, N0, generator=g).cuda()
    Wx = (torch.randn(N0, 3, generator=g) * 0.3).cuda()
    seg = e.SegTable.get(xyz.device, [c * m * ns for c in clouds])
    stats = torch.zeros(len(clouds), 2, N0, dtype=torch.float64, device="cuda")
    Y = e.group_lift_rows_seg(P, xyz, new_xyz, idx, Wx, True, r, stats, seg, out_bf16=bf16)
    inv = e.group_inverse_index(idx, N)
    G = torch.randn(Y.shape, generator=g).cuda().to(Y.dtype)
    consts = (torch.randn(len(clouds), 3, N0, generator=g) * 0.5).cuda().contiguous()
    acc = torch.zeros(len(clouds), 3 * N0 + 9, device="cuda")
    S = e.group_lift_rows_grad_seg(G, P, Wx, consts, xyz, new_xyz, inv, ns, True, r, acc, seg)
    c0 = 0
    for s_, nc in enumerate(clouds):
        c1, r0, r1 = c0 + nc, c0 * m * ns, (c0 + nc) * m * ns
        st = torch.zeros(2, N0, dtype=torch.float64, device="cuda")
        y1 = e.group_lift_rows(P[c0:c1].contiguous(), xyz[c0:c1].contiguous(), new_xyz[c0:c1].contiguous(),
                               idx[c0:c1].contiguous(), Wx, True, r, stats=st, out_bf16=bf16)
        assert torch.equal(Y[r0:r1], y1), s_
        torch.testing.assert_close(stats[s_], st, rtol=1e-14, atol=0)          # (fp64 atomics of identical fp32 partial sums)
        inv1 = e.group_inverse_index(idx[c0:c1].contiguous(), N)
        a1 = torch.zeros(3 * N0 + 9, device="cuda")
        s1 = e.group_lift_rows_grad(G[r0:r1].contiguous(), P[c0:c1].contiguous(), Wx, consts[s_].contiguous(), xyz[c0:c1].contiguous(),
                                    new_xyz[c0:c1].contiguous(), inv1, ns, True, r, a1)
        heavy = int(((inv1[0][1:] - inv1[0][:-1]) > 192).sum())
        if heavy == 0:                      # (heavy points are ADDED with atomics by sixteen waves each: not bit-reproducible)
            assert torch.equal(S[c0:c1], s1), s_
            assert torch.equal(acc[s_], a1), s_
        else:
            torch.testing.assert_close(S[c0:c1], s1, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(acc[s_], a1, rtol=1e-5, atol=1e-4)
        c0 = c1
