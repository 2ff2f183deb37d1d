"""Per-call times of the fused TripletGCN block entry points (csrc/gcn_fused.hip) at the layer's four shapes, against the
scan count.  A call from python costs the host 14-19 us (allocations + ctypes): rows below that are the HOST's rate, the device
durations of the small shapes are in profiles/r04_gcn_kernel_trace.md (rocprofv3 --kernel-trace over tools/gcn_time.py)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"))
from pointnet2_ops import _ext as e


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):              # best of three timing loops (a loop occasionally catches a ~50 ms host stall)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record(); torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / iters * 1e3)
    return best


def case(S, rows, K, N, bn=True):
    g = torch.Generator().manual_seed(0)
    ptr = torch.arange(S + 1, dtype=torch.int64) * rows
    R = S * rows
    A = torch.randn(R, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.zeros(N).cuda(); gamma = torch.ones(N).cuda(); beta = torch.zeros(N).cuda()
    G = torch.randn(R, N, generator=g).cuda(); ptr = ptr.cuda()
    out, ypre, mean, rstd = e.gcn_linear(W, b, ptr, S, A=A, bn=(gamma, beta, 1e-5), relu=True)
    dW = torch.zeros(N, K).cuda(); db = torch.zeros(N).cuda(); dg = torch.zeros(N).cuda(); dbe = torch.zeros(N).cuda()
    row = {"scans": S, "rows": rows, "K": K, "N": N}
    row["fwd_us"] = round(timeit(lambda: e.gcn_linear(W, b, ptr, S, A=A, bn=(gamma, beta, 1e-5), relu=True)), 1)
    row["grad_w_us"] = round(timeit(lambda: e.gcn_linear_grad_w((N, K), ptr, S, dW, db, G=G, bn=(ypre, mean, rstd, gamma, beta),
                                                                 relu=True, A=A, dgamma=dg, dbeta=dbe)), 1)
    gz = e.gcn_linear_grad_w((N, K), ptr, S, dW, db, G=G, bn=(ypre, mean, rstd, gamma, beta), relu=True, A=A, dgamma=dg, dbeta=dbe)
    row["grad_x_us"] = round(timeit(lambda: e.gcn_linear_grad_x(gz, W, ptr, S)), 1)
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    for S in (1, 8, 32):
        case(S, 72, 768, 512)      # nn1[0] (plain rows instead of the triplet gather)
        case(S, 72, 512, 1280)     # nn1[3]
        case(S, 9, 512, 512)       # nn2[0]
        case(S, 9, 512, 256)       # nn2[3]
