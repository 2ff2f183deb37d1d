"""Per-phase cycle counts of pool_bwd64_kernel (argument "128": pool_bwd128_kernel; all waves share one role there) from
a library built with -DPB_PROF: csrc/exp/libpn2_PROF.so."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd")]
os.environ["PN2_HIP_LIB"] = os.path.join(REPO, "4d-or_amd/csrc/exp/libpn2_PROF.so")
import torch
from pointnet2_ops import _ext as e
M, K, N, ns = (1048576, 128, 256, 32) if "128" in sys.argv[1:] else (4194304, 64, 128, 64)
dev = torch.device("cuda:0")
R = M // ns
yp = torch.randn(M, K, device=dev)
fin = torch.stack([torch.zeros(K), torch.ones(K), torch.ones(K), torch.zeros(K)]).to(dev).contiguous()
W = torch.randn(N, K, device=dev) / K ** 0.5
consts = torch.randn(3, N, device=dev) * 0.01
arg = torch.randint(0, ns, (R, N), device=dev, dtype=torch.int32)
gPm = torch.randn(R, N, device=dev) * (torch.rand(R, N, device=dev) > 0.3)
sums = torch.zeros(2, K, dtype=torch.float64, device=dev)
Gout = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev)
nb = int(e._lib.pn2_pool_bwd_workspace_bytes(M, N, K))
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
for _ in range(2):
    e._call("pn2_pool_bwd", yp, M, N, K, ns, e._ptr(yp), e._ptr(fin), e._ptr(W), e._ptr(consts), e._ptr(arg), e._ptr(gPm),
            e._ptr(Gout), e._ptr(sums), e._ptr(dW), e._ptr(ws), nb)
torch.cuda.synchronize()
grid = 256
prof = Gout.view(-1)[:64 * 8 * 10 * 2].view(torch.int64).view(64, 8, 10).cpu().double()
names = ["A stage", "barrier1", "B sparse" if K == 64 else "B: S", "C mfma", "barrier2", "D epi", "barrier3", "E store",
         "-" if K == 64 else "B: T", "loop head"]
tiles = (M // 64) / grid
print("tiles per WG", tiles)
for role, sl in (("waves0-3 (aG+S)", slice(0, 4)), ("waves4-7 (gram+T)", slice(4, 8))):
    m = prof[:, sl].mean(dim=(0, 1)) / tiles
    print(role, " ".join(f"{n}:{v:7.0f}" for n, v in zip(names, m.tolist())), " total", float(m.sum()))
