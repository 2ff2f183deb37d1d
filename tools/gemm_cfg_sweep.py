"""Sweep tile configurations of the forward BN+ReLU GEMM (needs the PN2_EXP_CFG experiment build)."""
import os, sys, json, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext
dev = torch.device("cuda:0")
B = 32
shapes = [("sa1.l2", B * 2048 * 64, 64, 64), ("sa1.l3", B * 2048 * 64, 64, 128), ("sa2.l2", B * 1024 * 32, 128, 128),
          ("sa2.l3", B * 1024 * 32, 128, 256)]
cfgs = {64: ["2,32,1", "2,16,1", "1,32,2", "2,8,1"], 128: ["2,32,2", "4,16,1", "4,32,1", "2,16,2", "4,8,1", "2,32,1", "2,16,1"],
        256: ["4,16,2", "2,32,2", "4,16,1", "2,16,2", "4,8,1"]}
grids = [256, 512, 768, 1024]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1)
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    ref = None
    for cfg in [None] + cfgs[N]:
        row = {}
        for g in grids:
            if cfg is None:
                os.environ.pop("PN2_GEMM_CFG", None)
                if g != 512: continue
            else:
                os.environ["PN2_GEMM_CFG"] = cfg
            os.environ["PN2_GEMM_GRID"] = str(g)
            for _ in range(2):
                y = _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
            torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            elif not torch.allclose(y, ref, atol=1e-3, rtol=1e-3): row["BAD"] = g
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6):
                _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
            e.record(); torch.cuda.synchronize()
            row[g] = round(s.elapsed_time(e) / 6 * 1e3)
        print(name, cfg, row, flush=True)
    del x
