import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "4d-or_amd"))
from pointnet2_ops import _ext as e
g = torch.Generator().manual_seed(5)
M, N, ns, K = 64 * 700, 128, 64, 64
R = M // ns
yp = torch.randn(M, K, generator=g).cuda()
fin = torch.stack([torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5, torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3]).cuda().contiguous()
W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
consts = (torch.randn(3, N, generator=g) * 0.1).cuda().contiguous()
arg = torch.randint(0, ns, (R, N), generator=g, dtype=torch.int32).cuda()
gPm = (torch.randn(R, N, generator=g) * (torch.rand(R, N, generator=g) > 0.3)).cuda()
e.X3_GEMM = False
G0, dW0 = e.pool_bwd(yp, fin, W, consts, arg, gPm, ns, torch.zeros(2, K, dtype=torch.float64, device="cuda"))
e.X3_GEMM, e.X3_MIN_ROWS = True, 0
G1, dW1 = e.pool_bwd(yp, fin, W, consts, arg, gPm, ns, torch.zeros(2, K, dtype=torch.float64, device="cuda"))
D = (dW1 - dW0).double()
print("max |D|", float(D.abs().max()), "max |dW|", float(dW0.abs().max()))
A = (consts[1].double()[:, None] * W.double())          # diag(c2) W  (N x K)
E = torch.linalg.lstsq(A, D).solution                   # K x K
print("Gram hypothesis residual", float((A @ E - D).abs().max()))
for bi in range(2):
    for bj in range(2):
        print("block", bi, bj, float(E[32 * bi:32 * bi + 32, 32 * bj:32 * bj + 32].abs().max()))
z = torch.relu(yp.double() * fin[2].double() + fin[3].double())
Gm = z.t() @ z
print("true Gram block maxima", [float(Gm[32 * i:32 * i + 32, 32 * j:32 * j + 32].abs().max()) for i in range(2) for j in range(2)])
print("E/Gm ratio sample (1,1)", float((E[32:, 32:] / Gm[32:, 32:]).median()), "(0,1)", float((E[:32, 32:] / Gm[:32, 32:]).median()), "(0,0)", float((E[:32, :32] / Gm[:32, :32]).median()))
