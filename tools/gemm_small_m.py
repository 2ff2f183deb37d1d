"""Launch time of the shared-MLP GEMM kernels against M (rows) at fixed K, N: where does the fixed per-launch cost (weight
staging, pipeline fill, persistent-grid tail) end and the streaming regime begin?  One JSON line per (kernel, K, N, M).
    python tools/gemm_small_m.py            (GPU box)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch  # noqa: E402
from pointnet2_ops import _ext as e  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3      # us


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    for K, N in ((64, 128), (128, 128), (256, 256)):
        W = torch.randn(N, K, generator=g).to(dev)
        p = (torch.rand(K, generator=g).to(dev) + 0.5, torch.randn(K, generator=g).to(dev) * 0.1)
        for M in (8192, 32768, 131072, 524288, 2097152):
            xb = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
            xf = xb.float()
            stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
            row = {"K": K, "N": N, "M": M}
            row["bf16_us"] = round(timeit(lambda: e.mlp_gemm_bf16(xb, W, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=stats)), 1)
            row["f32_us"] = round(timeit(lambda: e.mlp_gemm(xf, W, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=stats)), 1)
            row["bf16_ideal_us_at_6TBps"] = round(M * (K + N) * 2 / 6e12 * 1e6, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
