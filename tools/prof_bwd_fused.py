"""A few launches of the one-pass backward kernel at the SA1 shapes (for rocprofv3 --pmc passes / timing)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext
dev = torch.device("cuda:0")
B = 32
for name, M, N, K, pool in [("sa1.l2", B * 2048 * 64, 64, 64, False), ("sa1.l3", B * 2048 * 64, 128, 64, True)]:
    y = torch.randn(M, N, device=dev); yp = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
    consts = torch.rand(3, N, device=dev); fin = torch.rand(4, K, device=dev)
    ns = 64
    if pool:
        arg = torch.randint(0, ns, (M // ns, N), device=dev, dtype=torch.int32); gP = torch.randn(M // ns, N, device=dev)
        kw = dict(arg=arg, gP=gP, ns=ns); mode = _ext.PRO_POOLG
    else:
        kw = dict(G=torch.randn(M, N, device=dev)); mode = _ext.PRO_GY
    for _ in range(2):
        _ext.mlp_bwd_fused(y, consts, W, yp, fin, mode, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        _ext.mlp_bwd_fused(y, consts, W, yp, fin, mode, **kw)
    e.record(); torch.cuda.synchronize()
    print(name, "fused us", round(s.elapsed_time(e) / 5 * 1e3, 1), flush=True)
    del y, yp
