"""Per-kernel micro-benchmarks at BASELINE config-2 shapes.
Prints one JSON line per kernel: time, algorithmic bytes (SURVEY.md 8d formulas), achieved GB/s, `frac_of_8TBps` and,
beside it, north_star's `target` of 0.60 for the HBM-bound grouping / query kernels (FPS is latency-bound: no target)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]

import torch  # noqa: E402
from pointnet2_ops import _ext  # noqa: E402


def unit_ball(B, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)
    p = p - p.mean(dim=1, keepdim=True)
    return (p / p.norm(dim=2).amax(dim=1).view(B, 1, 1)).contiguous()


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, t, nbytes, target=0.60, **kw):
    row = dict(kernel=name, ms=round(t * 1e3, 4), alg_MB=round(nbytes / 1e6, 2), GBps=round(nbytes / t / 1e9, 1),
               frac_of_8TBps=round(nbytes / t / 8e12, 4))
    if target is not None:
        row["target"] = target                      # north_star: >= 60 % of the HBM roofline for ball-query + group
    print(json.dumps(dict(row, **kw)), flush=True)


def fps_sweep(dev):
    """Per-round cost of each FPS kernel variant (heuristic tuning)."""
    cases = [(32, 50000, 2048, [("coop", (1, 8)), ("coop", (2, 16)), ("coop", (4, 32)), ("coop", (2, 8)), ("coop", (4, 16)), ("coop", None)]),
             (8, 200000, 512, [("coop", 32), ("stream", None)]),
             (1, 20000, 512, [("resident", None), ("coop", 4), ("coop", 16)]),
             (72, 8000, 512, [("resident", None), ("coop", 2)]),
             (9, 4000, 512, [("resident", None), ("coop", 2), ("coop", 4)]),
             (32, 2048, 1024, [("resident", None), ("coop", 2)])]
    for B, N, m, variants in cases:
        x = unit_ball(B, N, 3).to(dev)
        for mode, g in variants:
            nc = None
            if isinstance(g, tuple):
                nc, g = g
            # variants are forced through the library's test hook (the PN2_FPS_* environment switches were removed in round 2)
            try:
                with _ext.fps_plan_override(mode, g=g or 0, nc=nc or 0):
                    t = timeit(lambda: _ext.furthest_point_sampling(x, m), iters=3, warm=1)
                report(f"fps_sweep B{B} N{N} m{m} {mode} NC{nc} G{g}", t, B * (12 * N + 4 * m),
                       us_per_round=round(t / (m - 1) * 1e6, 3))
            except RuntimeError as e:
                print("fps_sweep", B, N, mode, g, "->", e, flush=True)


def mlp_sweep(dev):
    """Per-shape time of the fused MFMA shared-MLP kernels (fwd / dgrad / wgrad), TF/s and GB/s."""
    B = int(os.environ.get("MB_B", 32))
    shapes = [  # (name, M, K, N)
        ("sa1.l1", B * 2048 * 64, 6, 64), ("sa1.l2", B * 2048 * 64, 64, 64), ("sa1.l3", B * 2048 * 64, 64, 128),
        ("sa2.l1", B * 1024 * 32, 131, 128), ("sa2.l2", B * 1024 * 32, 128, 128), ("sa2.l3", B * 1024 * 32, 128, 256),
        ("sa3.l1", B * 512 * 16, 259, 128), ("sa3.l3", B * 512 * 16, 128, 256),
        ("fp2.l1", B * 1024, 512, 256), ("fp2.l2", B * 1024, 256, 288)]
    for name, M, K, N in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.1
        p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.randn(K, device=dev) * 0.1)
        stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
        fl = 2.0 * M * K * N
        t = timeit(lambda: _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats), iters=5)
        report(f"mlp_fwd {name} M{M} K{K} N{N}", t, 4 * (M * K + M * N), TFps=round(fl / t / 1e12, 1))
        # dgrad: out[M,K] = gy[M,N] @ Wt[K,N]^T with GY prologue + MASK epilogue
        g = torch.randn(M, N, device=dev)
        y = torch.randn(M, N, device=dev)
        c = (torch.rand(N, device=dev), torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1)
        Wt = W.t().contiguous()
        fin = torch.rand(4, K, device=dev)
        s2 = torch.zeros(2, K, dtype=torch.float64, device=dev)
        t = timeit(lambda: _ext.mlp_gemm(g, Wt, pro=_ext.PRO_GY, epi=_ext.EPI_MASK, X2=y, p=c, stats=s2, Yprev=x,
                                         e_fin=fin), iters=5)
        report(f"mlp_dgrad {name}", t, 4 * (2 * M * N + 2 * M * K), TFps=round(fl / t / 1e12, 1))
        consts = torch.stack(c).contiguous()
        t = timeit(lambda: _ext.mlp_wgrad(y, consts, x, _ext.PRO_GY, _ext.PRO_BNRELU, G=g, a_fin=fin), iters=5)
        report(f"mlp_wgrad {name}", t, 4 * (2 * M * N + M * K), TFps=round(fl / t / 1e12, 1))
        del x, g, y


def main():
    B = int(os.environ.get("MB_B", 32))
    N = int(os.environ.get("MB_N", 50000))
    m, ns, r, C = 2048, 64, 0.2, 3
    dev = torch.device("cuda:0")
    if os.environ.get("MB_MLP") == "1":
        mlp_sweep(dev)
        return
    if os.environ.get("MB_FPS_SWEEP") == "1":
        fps_sweep(dev)
        return
    xyz = unit_ball(B, N).to(dev)
    feats = torch.rand(B, C, N, device=dev)
    feats_rows = feats.transpose(1, 2).contiguous()

    t = timeit(lambda: _ext.furthest_point_sampling(xyz, m), iters=3, warm=1)
    report("fps", t, B * (12 * N + 4 * m), target=None, B=B, N=N, m=m, us_per_round=round(t / (m - 1) * 1e6, 3))
    sel = _ext.furthest_point_sampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()

    for (n2, m2) in ((2048, 1024), (1024, 512), (512, 256)):
        x2 = xyz[:, :n2].contiguous()
        t = timeit(lambda: _ext.furthest_point_sampling(x2, m2), iters=5)
        report(f"fps_{n2}_{m2}", t, B * (12 * n2 + 4 * m2), target=None, us_per_round=round(t / (m2 - 1) * 1e6, 3))
    for (b2, n2, m2) in ((72, 8000, 512), (9, 4000, 512)):
        x2 = unit_ball(b2, n2, 1).to(dev)
        t = timeit(lambda: _ext.furthest_point_sampling(x2, m2), iters=5)
        report(f"fps_B{b2}_{n2}_{m2}", t, b2 * (12 * n2 + 4 * m2), target=None, us_per_round=round(t / (m2 - 1) * 1e6, 3))

    t = timeit(lambda: _ext.ball_query(new_xyz, xyz, r, ns))
    report("ball_query", t, B * (12 * N + 12 * m + 4 * m * ns), r=r, ns=ns)
    idx = _ext.ball_query(new_xyz, xyz, r, ns)
    x2 = unit_ball(72, 8000, 1).to(dev)
    nx2 = x2[:, :512].contiguous()
    t = timeit(lambda: _ext.ball_query(nx2, x2, 0.2, 32))
    report("ball_query_B72_8000_512_32", t, 72 * (12 * 8000 + 12 * 512 + 4 * 512 * 32))

    xyz_t = xyz.transpose(1, 2).contiguous()
    t = timeit(lambda: _ext.group_points(xyz_t, idx))
    report("group_points_C3", t, B * (4 * m * ns + 4 * C * N + 4 * C * m * ns))
    go = torch.rand(B, C, m, ns, device=dev)
    t = timeit(lambda: _ext.group_points_grad(go, idx, N))
    report("group_points_grad_C3", t, B * (4 * m * ns + 4 * C * m * ns + 4 * C * N))

    t_g = timeit(lambda: _ext.group_concat_rows(xyz, new_xyz, feats_rows, idx, True, True, r))
    report("group_concat_rows_C3", t_g, B * (4 * m * ns + 12 * N + 12 * m + 4 * C * N + 4 * (3 + C) * m * ns))
    # ball query + grouping as ONE kernel (pn2_ball_query_group) against the pair it replaces; SURVEY.md 8d: fused
    # (idx kept) = B (12 N + 12 m + 4 C N + 4 (3 + C) m ns + 4 m ns) = 156.7 MB at this shape
    fused_bytes = B * (12 * N + 12 * m + 4 * C * N + 4 * (3 + C) * m * ns + 4 * m * ns)
    t_q = timeit(lambda: _ext.ball_query(new_xyz, xyz, r, ns))
    report("ball_query+group_concat_rows_C3 (two kernels)", t_q + t_g, fused_bytes, r=r, ns=ns)
    for w in (1, 4, 0):
        t = timeit(lambda: _ext.ball_query_group(new_xyz, xyz, feats_rows, r, ns, True, True, slab_w=w))
        report(f"ball_query_group_fused_C3 slab_w={w or 'auto'}", t, fused_bytes, r=r, ns=ns,
               note="query + (xyz - centre) / r + colours in one pass, idx kept for the backward")
    i_f, r_f = _ext.ball_query_group(new_xyz, xyz, feats_rows, r, ns, True, True)
    assert torch.equal(i_f, idx) and torch.equal(r_f, _ext.group_concat_rows(xyz, new_xyz, feats_rows, idx, True, True, r))
    gr = torch.rand(B, m, ns, 3 + C, device=dev)
    t = timeit(lambda: _ext.group_rows_grad(gr, idx, N, C, 3))
    report("group_rows_grad_C3", t, B * (4 * m * ns + 4 * C * m * ns + 4 * C * N))

    # SA2-like: N=2048 feats C=128, m=1024, ns=32
    N2, m2, ns2, C2 = 2048, 1024, 32, 128
    x2 = xyz[:, :N2].contiguous()
    nx2 = x2[:, :m2].contiguous()
    f2 = torch.rand(B, N2, C2, device=dev)
    i2 = _ext.ball_query(nx2, x2, 0.4, ns2)
    t = timeit(lambda: _ext.ball_query(nx2, x2, 0.4, ns2))
    report("ball_query_sa2", t, B * (12 * N2 + 12 * m2 + 4 * m2 * ns2))
    t = timeit(lambda: _ext.group_concat_rows(x2, nx2, f2, i2, True, True, 0.4))
    report("group_concat_rows_sa2_C128", t, B * (4 * m2 * ns2 + 12 * N2 + 12 * m2 + 4 * C2 * N2 + 4 * (3 + C2) * m2 * ns2))
    g2 = torch.rand(B, m2, ns2, 3 + C2, device=dev)
    t = timeit(lambda: _ext.group_rows_grad(g2, i2, N2, C2, 3))
    report("group_rows_grad_sa2_C128", t, B * (4 * m2 * ns2 + 4 * C2 * m2 * ns2 + 4 * C2 * N2))
    h = torch.rand(B * m2, ns2, 256, device=dev)
    t = timeit(lambda: _ext.rows_max(h))
    report("rows_max_sa2", t, h.numel() * 4 + B * m2 * 256 * 8)
    o, a = _ext.rows_max(h)
    t = timeit(lambda: _ext.rows_max_grad(o, a, ns2))
    report("rows_max_grad_sa2", t, h.numel() * 4 + B * m2 * 256 * 8)

    u, k = xyz[:, :1024].contiguous(), xyz[:, 1024:1536].contiguous()
    t = timeit(lambda: _ext.three_nn(u, k))
    report("three_nn_1024_512", t, B * (12 * 1024 + 12 * 512 + 24 * 1024))
    # the literal gradients of FP's interpolation (B, 256, 1024) <- 512 known points and of gather_points (B, 3, 2048) -> 50k points
    _d, i3 = _ext.three_nn(u, k)
    w3 = torch.rand(B, 1024, 3, device=dev)
    go3 = torch.rand(B, 256, 1024, device=dev)
    t = timeit(lambda: _ext.three_interpolate_grad(go3, i3, w3, 512))
    report("three_interpolate_grad_C256", t, B * (4 * 256 * 1024 + 24 * 1024 + 4 * 256 * 512))
    gi = torch.randint(0, N, (B, m), device=dev, dtype=torch.int32)
    gg = torch.rand(B, C, m, device=dev)
    t = timeit(lambda: _ext.gather_points_grad(gg, gi, N))
    report("gather_points_grad_C3", t, B * (4 * m + 4 * C * m + 4 * C * N))


if __name__ == "__main__":
    main()
