"""pn2_mlp_bwd_bf16_fold (stored y_0) vs pn2_mlp_bwd_bf16_fold_first (y_0 re-formed) at the backbone's SA1 shape."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext as e
dev = "cuda"; BF = torch.bfloat16
M, N, K, K0 = 32 * 2048 * 64, 64, 64, 6
g = torch.Generator().manual_seed(0)
X = torch.zeros(M, 8); X[:, :K0] = torch.randn(M, K0, generator=g); X = X.to(BF).to(dev)
W0 = (torch.randn(K, K0, generator=g) * 0.5).to(dev)
W1 = (torch.randn(N, K, generator=g) / 8).to(dev); Wt = W1.t().contiguous()
y0 = e.mlp_gemm_bf16(X, W0, pro=e.PRO_NONE, epi=e.EPI_NONE)
y1 = torch.randn(M, N, generator=g).to(BF).to(dev)
G = torch.randn(M, N, generator=g).to(BF).to(dev)
c = (torch.randn(3, N, generator=g) * 0.3).to(dev).contiguous()
fin = torch.stack([torch.zeros(K), torch.ones(K), torch.ones(K), torch.zeros(K)]).to(dev).contiguous()
sums = torch.zeros(2, K, dtype=torch.float64, device=dev); dW = torch.zeros(N, K, device=dev); P1 = torch.zeros(K, K0, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
t1 = timed(lambda: e.mlp_bwd_bf16_fold(y1, c, Wt, y0, fin, X, K0, e.PRO_GY, G=G, sums=sums, dW=dW, P1=P1))
t2 = timed(lambda: e.mlp_bwd_bf16_fold_first(y1, c, Wt, W0, fin, X, K0, e.PRO_GY, G=G, sums=sums, dW=dW, P1=P1))
# pooled last layer (N = 128, K = 64, 64 rows per group): stored y_L vs re-formed
N2, ns = 128, 64
W2 = (torch.randn(N2, K, generator=g) / 8).to(dev); W2t = W2.t().contiguous()
c2 = (torch.randn(3, N2, generator=g) * 0.3).to(dev).contiguous()
y2 = torch.randn(M, N2, generator=g).to(BF).to(dev)
arg = torch.randint(0, ns, (M // ns, N2), generator=g, dtype=torch.int32).to(dev)
gP = torch.randn(M // ns, N2, generator=g).to(dev)
dW2 = torch.zeros(N2, K, device=dev)
t3 = timed(lambda: e.mlp_bwd_bf16(y2, c2, W2t, y0, fin, e.PRO_POOLG, arg=arg, gP=gP, ns=ns, sums=sums, dW=dW2))
t4 = timed(lambda: e.mlp_bwd_bf16_pool(c2, W2t, y0, fin, arg, gP, ns, sums=sums, dW=dW2))
print(json.dumps({"fold_ms": round(t1, 4), "fold_first_ms": round(t2, 4), "pool_bwd_stored_ms": round(t3, 4), "pool_bwd_reformed_ms": round(t4, 4)}))
