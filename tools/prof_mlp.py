"""Run a few launches of selected fused-MLP shapes (for rocprofv3 --pmc passes)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext
dev = torch.device("cuda:0")
B = 32
shapes = [("sa1.l2", B * 2048 * 64, 64, 64), ("sa1.l3", B * 2048 * 64, 64, 128), ("sa2.l2", B * 1024 * 32, 128, 128),
          ("sa2.l3", B * 1024 * 32, 128, 256)]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1)
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    for _ in range(3):
        _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
    g = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev)
    consts = torch.rand(3, N, device=dev); fin = torch.rand(4, K, device=dev)
    for _ in range(3):
        _ext.mlp_wgrad(y, consts, x, _ext.PRO_GY, _ext.PRO_BNRELU, G=g, a_fin=fin)
    torch.cuda.synchronize()
    del x, g, y
