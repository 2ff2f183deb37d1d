"""A/B of the layer above a lifted first layer, stored y0 vs re-formed y0 (csrc/mlp_gemm.hip PRO_LIFT / EPI_MASKL), per kernel,
at the headline's SA2 / SA3 / SA4 shapes.  python tools/lift_free_bench.py [out.jsonl]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch  # noqa: E402

from pointnet2_ops import _ext as e  # noqa: E402

dev = "cuda"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rows = []
for name, B, N, m, ns, C, r in [("SA2", 32, 2048, 1024, 32, 128, 0.4), ("SA3", 32, 1024, 512, 16, 256, 0.8), ("SA4", 32, 512, 256, 16, 256, 1.2)]:
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev) * torch.tensor([4.0, 4.0, 2.5], device=dev)
    new_xyz = xyz[:, :m].contiguous()
    idx = e.ball_query(new_xyz, xyz, r, ns)
    N0 = N1 = 128
    f = torch.randn(B, N, C, generator=g).to(dev)
    W = (torch.randn(N0, 3 + C, generator=g) * 0.1).to(dev)
    Wx = W[:, :3].contiguous()
    P = e.mlp_gemm(f.view(-1, C), W[:, 3:].contiguous()).view(B, N, N0)
    M = B * m * ns
    fin0 = torch.stack([torch.zeros(N0), torch.ones(N0), torch.ones(N0), torch.zeros(N0)]).to(dev).contiguous()
    W1 = (torch.randn(N1, N0, generator=g) * 0.1).to(dev)
    Wt = W1.t().contiguous()
    consts = (torch.randn(3, N1, generator=g) * 0.5).to(dev).contiguous()
    st = torch.zeros(2, N0, dtype=torch.float64, device=dev)
    st1 = torch.zeros(2, N1, dtype=torch.float64, device=dev)
    dW = torch.zeros(N1, N0, device=dev)
    G = torch.randn(M, N1, generator=g).to(dev)
    # stored route
    y0 = e.group_lift_rows(P, xyz, new_xyz, idx, Wx, True, r, stats=st)
    Y1 = e.mlp_gemm(y0, W1, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=(fin0[2], fin0[3]), stats=st1)
    t = {"level": name, "M": M}
    t["stored_lift_rows"] = timed(lambda: e.group_lift_rows(P, xyz, new_xyz, idx, Wx, True, r, stats=st))
    t["stored_gemm"] = timed(lambda: e.mlp_gemm(y0, W1, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=(fin0[2], fin0[3]), stats=st1))
    t["stored_wgrad"] = timed(lambda: e.mlp_wgrad(Y1, consts, y0, e.PRO_GY, e.PRO_BNRELU, G=G, a_fin=fin0, dW=dW))
    t["stored_dgrad"] = timed(lambda: e.mlp_gemm(G, Wt, pro=e.PRO_GY, epi=e.EPI_MASK, X2=Y1, p=(consts[0], consts[1], consts[2]),
                                                 stats=st, Yprev=y0, e_fin=fin0, M=M))
    # free route
    Pq, Q = e.lift_points(P, xyz, new_xyz, Wx, True, r)
    gidx = e.group_lift_stats(Pq, Q, idx, N, st)
    t["free_points"] = timed(lambda: e.lift_points(P, xyz, new_xyz, Wx, True, r))
    t["free_stats"] = timed(lambda: e.group_lift_stats(Pq, Q, idx, N, st))
    t["free_gemm"] = timed(lambda: e.mlp_gemm_lift(Pq, gidx, Q, ns, fin0, W1, st1))
    t["free_wgrad"] = timed(lambda: e.mlp_wgrad_lift(Y1, consts, G, Pq, gidx, Q, ns, fin0, dW=dW))
    t["free_dgrad"] = timed(lambda: e.mlp_dgrad_lift(G, Y1, consts, Wt, st, Pq, gidx, Q, ns, fin0))
    t["stored_total"] = sum(v for k, v in t.items() if k.startswith("stored_"))
    t["free_total"] = sum(v for k, v in t.items() if k.startswith("free_"))
    t = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}
    print(json.dumps(t), flush=True)
    rows.append(t)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        for t in rows:
            fh.write(json.dumps(t) + "\n")
