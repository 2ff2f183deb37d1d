"""Times pn2_pool_bwd / pn2_mlp_gemm_pool alone at the headline shapes (PN2_HIP_LIB selects an experiment build)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd")]
import torch
from pointnet2_ops import _ext as e

SHAPES = [(4194304, 64, 128, 64), (1048576, 128, 256, 32), (262144, 128, 256, 16), (131072, 128, 256, 16)]
dev = torch.device("cuda:0")
for M, K, N, ns in SHAPES:
    torch.manual_seed(0)
    R = M // ns
    yp = torch.randn(M, K, device=dev)
    fin = torch.stack([torch.zeros(K), torch.ones(K), torch.ones(K), torch.zeros(K)]).to(dev).contiguous()
    W = torch.randn(N, K, device=dev) / K ** 0.5
    consts = torch.randn(3, N, device=dev) * 0.01
    arg = torch.randint(0, ns, (R, N), device=dev, dtype=torch.int32)
    gPm = torch.randn(R, N, device=dev) * (torch.rand(R, N, device=dev) > 0.3)
    sums = torch.zeros(2, K, dtype=torch.float64, device=dev)
    def run():
        return e.pool_bwd(yp, fin, W, consts, arg, gPm, ns, sums)
    e.X3_GEMM = len(sys.argv) > 1 and sys.argv[1] == "x3"
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        run()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    print(json.dumps({"op": "pool_bwd", "M": M, "K": K, "N": N, "ns": ns, "ms": round(ms, 4),
                      "TFLOPs_4MKK": round(4 * M * K * K / ms / 1e9, 1), "f32x3_route": bool(e.X3_GEMM and K == 64), "lib": os.path.basename(e.LIB_PATH)}))
