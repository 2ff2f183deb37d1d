import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext as e
e.X3_GEMM = True; e.X3_MIN_ROWS = 0
dev = "cuda"
for ns, K, N in ((16, 64, 128), (32, 64, 128), (64, 128, 256)):
    torch.manual_seed(ns * 1000 + K + N)
    R = 37 if ns <= 32 else 11
    M = R * ns
    x = torch.randn(M, K, device=dev)
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    gamma = torch.randn(N, device=dev)
    y = e.mlp_gemm(x, W, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=torch.zeros(2, N, dtype=torch.float64, device=dev))
    Wf, sgn = e.pool_flip_rows(W, gamma)
    y2 = e.mlp_gemm(x, Wf, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=torch.zeros(2, N, dtype=torch.float64, device=dev))
    print(ns, K, N, "flipped-weight product == -product:", bool(torch.equal(y2, y * sgn)), float((y2 - y * sgn).abs().max()))
    pmax, parg = e.mlp_gemm_pool(x, Wf, sgn, ns, p=p, stats=torch.zeros(2, N, dtype=torch.float64, device=dev))
    psz = min(ns, 32)
    ref = y2.view(M // psz, psz, N).max(1)
    print("   pmax == max of stored y2:", bool(torch.equal(pmax, ref.values)), float((pmax - ref.values).abs().max()),
          "arg same:", float((parg == ref.indices.int()).float().mean()))
