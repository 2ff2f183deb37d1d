"""Can the cooperative FPS kernel (side stream) overlap the persistent MLP kernels (main stream)?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball
dev = torch.device("cuda:0")
xyz = unit_ball(32, 50000).to(dev)
M, K, N = 32 * 1024 * 32, 128, 128
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1)
stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
g = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev)
consts = torch.rand(3, N, device=dev); fin = torch.rand(4, K, device=dev)
def mlp_work(n=12):
    for _ in range(n):
        _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)
        _ext.mlp_wgrad(y, consts, x, _ext.PRO_GY, _ext.PRO_BNRELU, G=g, a_fin=fin)
def fps():
    return _ext.furthest_point_sampling(xyz, 2048)
os.environ["PN2_FPS_CHECK"] = "0"
for _ in range(2): mlp_work(); fps()
torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
t_mlp = timed(mlp_work); t_fps = timed(fps)
side = torch.cuda.Stream(priority=-1)
def both():
    with torch.cuda.stream(side):
        r = fps()
    mlp_work()
    torch.cuda.current_stream().wait_stream(side)
    return r
ts = [round(timed(both), 2) for _ in range(12)]
def both_rev():
    mlp_work(2)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream()) if False else None
        r = fps()
    mlp_work(10)
    torch.cuda.current_stream().wait_stream(side)
    return r
ts2 = [round(timed(both_rev), 2) for _ in range(12)]
print("fps first :", ts)
print("mlp first :", ts2)
t_both, t_both2 = ts[0], ts[1]
ref = fps(); torch.cuda.synchronize()
with torch.cuda.stream(side):
    r2 = fps()
mlp_work(); torch.cuda.synchronize()
print(f"mlp alone {t_mlp:.2f} ms, fps alone {t_fps:.2f} ms, concurrent {t_both:.2f} / {t_both2:.2f} ms (sum {t_mlp + t_fps:.2f}); fps result identical under overlap: {bool(torch.equal(ref, r2))}")
