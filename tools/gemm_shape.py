import os, sys, torch
sys.path[:0] = ["4d-or_amd"]
from pointnet2_ops import _ext
M, K, N = [int(v) for v in sys.argv[1:4]]
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
p = (torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1)
def run():
    st = torch.zeros(2, N, dtype=torch.float64, device="cuda")
    return _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=st)
for _ in range(3): run()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
e[0].record()
for i in range(10):
    run(); e[i + 1].record()
torch.cuda.synchronize()
t = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(10))
print(f"M{M} K{K} N{N} cfg {os.environ.get('PN2_GEMM_CFG')} grid {os.environ.get('PN2_GEMM_GRID')}: median {t[5]:.0f} us min {t[0]:.0f}")
