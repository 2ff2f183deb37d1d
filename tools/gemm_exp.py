"""Time the forward BN+ReLU GEMM shapes of SA1/SA2 (decomposition experiments: PN2_HIP_LIB=variant)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext
dev = torch.device("cuda:0")
B = 32
shapes = [("sa1.l2", B * 2048 * 64, 64, 64), ("sa1.l3", B * 2048 * 64, 64, 128), ("sa2.l2", B * 1024 * 32, 128, 128),
          ("sa2.l3", B * 1024 * 32, 128, 256)]
res = {}
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1)
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    for _ in range(3):
        _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
    e.record(); torch.cuda.synchronize()
    res[name] = round(s.elapsed_time(e) / 10 * 1e3, 1)
    del x
print(os.environ.get("PN2_HIP_LIB", "product").split("/")[-1], json.dumps(res))
