"""pn2_x3_gemm (split-bf16 product) against pn2_mlp_gemm / pn2_mlp_gemm_pool (exact fp32 MFMA) at the headline step's shapes:
time and error against a float64 product on sampled rows.  python tools/x3_gemm_bench.py [out.jsonl]"""
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


SHAPES = [("SA1 pooled last layer", 4194304, 64, 128, "pool", 64), ("SA2 pooled last layer", 1048576, 128, 256, "pool", 32),
          ("SA2 hidden layer", 1048576, 128, 128, "fwd", 0), ("SA2 input gradient", 1048576, 128, 128, "dgrad", 0),
          ("SA3 pooled last layer", 262144, 128, 256, "pool", 16), ("SA3 hidden layer", 262144, 128, 128, "fwd", 0),
          ("SA1 hidden layer (stored y0)", 4194304, 64, 64, "fwd", 0),
          ("SA1 second layer, first one re-formed (gemm_first, K0 6)", 4194304, 64, 64, "first", 6)]
rows = []
e.X3_GEMM = False
for name, M, K, N, kind, ns in SHAPES:
    g = torch.Generator().manual_seed(M % 1000 + K + N)
    X = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    p = ((torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev))
    st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    t = {"layer": name, "M": M, "K": K, "N": N, "kind": kind}
    pick = torch.randint(0, M, (256,), generator=g).to(dev)
    if kind == "first":
        K0 = ns
        X0 = torch.randn(M, K0, generator=g).to(dev)
        W0 = (torch.randn(K, K0, generator=g) / K0 ** 0.5).to(dev)
        fin0 = torch.stack([torch.zeros(K), torch.ones(K), p[0].cpu(), p[1].cpu()]).to(dev).contiguous()
        ex = lambda: e.mlp_gemm_first(X0, W0, fin0, W, epi=e.EPI_STATS, stats=st)
        x3 = lambda: e.x3_gemm_first(X0, W0, fin0, W, st)
        A64 = torch.relu((X0[pick].double() @ W0.double().t()) * fin0[2].double() + fin0[3].double())
        R64 = A64 @ W.double().t()
        scale = float((A64.abs() @ W.double().abs().t()).max())
        t["err_exact"] = float((ex()[pick].double() - R64).abs().max() / scale)
        t["err_f32x3"] = float((x3()[pick].double() - R64).abs().max() / scale)
    elif kind == "fwd":
        ex = lambda: e.mlp_gemm(X, W, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=st)
        x3 = lambda: e.x3_gemm(X, W, e.PRO_BNRELU, 1, p=p, stats=st)
        A64 = torch.relu(X[pick].double() * p[0].double() + p[1].double())
        R64 = A64 @ W.double().t()
        scale = float((A64.abs() @ W.double().abs().t()).max())
        t["err_exact"] = float((ex()[pick].double() - R64).abs().max() / scale)
        t["err_f32x3"] = float((x3()[pick].double() - R64).abs().max() / scale)
    elif kind == "pool":
        sgn = torch.ones(N, device=dev)
        ex = lambda: e.mlp_gemm_pool(X, W, sgn, ns, p=p, stats=st)
        x3 = lambda: e.x3_gemm(X, W, e.PRO_BNRELU, 3, p=p, stats=st, sgn=sgn, ns=ns)
        a, b = ex()[0], x3()[0]
        t["max_abs_diff_of_maxima"] = float((a - b).abs().max())
    else:
        Yl = torch.randn(M, K, generator=g).to(dev)
        c = (torch.randn(3, K, generator=g) * 0.5).to(dev)
        Yprev = torch.randn(M, N, generator=g).to(dev)
        e_fin = torch.stack([torch.zeros(N), torch.ones(N), torch.ones(N), torch.zeros(N)]).to(dev).contiguous()
        ex = lambda: e.mlp_gemm(X, W, pro=e.PRO_GY, epi=e.EPI_MASK, X2=Yl, p=(c[0], c[1], c[2]), stats=st, Yprev=Yprev, e_fin=e_fin)
        x3 = lambda: e.x3_gemm(X, W, e.PRO_GY, 2, X2=Yl, p=(c[0], c[1], c[2]), stats=st, Yprev=Yprev, e_fin=e_fin)
        t["max_abs_diff"] = float((ex() - x3()).abs().max())
    frags = e.x3_pack_weight(W, False)
    t["ms_exact"] = round(timed(ex), 4)
    t["ms_f32x3"] = round(timed(x3), 4)
    t["ms_weight_pack"] = round(timed(lambda: e.x3_pack_weight(W, False, out=frags)), 4)
    t["speedup"] = round(t["ms_exact"] / t["ms_f32x3"], 3)
    rows.append(t)
    print(json.dumps(t))
    del X
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        for t in rows:
            fh.write(json.dumps(t) + "\n")
