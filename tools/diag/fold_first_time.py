"""pn2_mlp_bwd_fused_fold_first exact vs f32x3 at the SA1 shape (M = 32 x 2048 x 64 rows, 64 -> 64, K0 = 6)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "4d-or_amd"))
import torch
from pointnet2_ops import _ext as e
M, K0, N0, N1 = 4194304, 6, 64, 64
g = torch.Generator().manual_seed(1)
X0 = torch.randn(M, K0, generator=g).cuda(); W0 = (torch.randn(N0, K0, generator=g) * 0.5).cuda(); W1 = (torch.randn(N1, N0, generator=g) * 0.2).cuda()
fin0 = torch.stack([torch.zeros(N0), torch.ones(N0), torch.ones(N0), torch.zeros(N0)]).cuda().contiguous()
y1 = torch.randn(M, N1, generator=g).cuda(); G = torch.randn(M, N1, generator=g).cuda()
consts = (torch.randn(3, N1, generator=g) * 0.3).cuda().contiguous()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
out = {}
for x3 in (False, True):
    e.X3_GEMM = x3
    s = torch.zeros(2, N0, dtype=torch.float64, device="cuda"); dW = torch.zeros(N1, N0, device="cuda"); P1 = torch.zeros(N0, K0, device="cuda")
    out["ms_f32x3" if x3 else "ms_exact"] = round(timed(lambda: e.mlp_bwd_fused_fold_first(y1, consts, W1, W0, fin0, X0, e.PRO_GY, G=G, sums=s, dW=dW, P1=P1)), 4)
print(json.dumps({"kernel": "pn2_mlp_bwd_fused_fold_first", "M": M, **out}))
