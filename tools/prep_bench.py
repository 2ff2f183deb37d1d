#!/usr/bin/env python
"""pn2_pool_bwd_prep / pn2_bn_relu_bwd_prep alone at the headline shapes: the vectorised kernel (round 5) against the
16-row-block kernel it replaces (forced by handing in a 4-byte-misaligned operand).  One JSON line per shape."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
from pointnet2_ops import _ext  # noqa: E402


def misaligned(t):
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    return v


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = "cuda"
    for name, R, C in [("SA1 pool", 65536, 128), ("SA2 pool", 32768, 256), ("SA3 pool", 16384, 256), ("SA4 pool", 8192, 256),
                       ("sgp rel SA1", 72 * 512 * 8, 128)]:
        yraw, pooled, gP = (torch.randn(R, C, device=dev) for _ in range(3))
        fin = torch.rand(4, C, device=dev) + 0.5
        sums = torch.zeros(2, C, dtype=torch.float64, device=dev)
        gPm = torch.empty_like(pooled)
        new = timeit(lambda: _ext.pool_bwd_prep(yraw, pooled, gP, fin, sums=sums))
        ym = misaligned(yraw)
        old = timeit(lambda: _ext.pool_bwd_prep(ym, pooled, gP, fin, sums=sums))
        mb = 16 * R * C / 1e6
        print(json.dumps({"kernel": "pn2_pool_bwd_prep", "shape": name, "R": R, "C": C, "alg_MB": round(mb, 1),
                          "vectorised_us": round(new, 1), "vectorised_GBps": round(mb / new * 1e3, 0),
                          "row_block_us": round(old, 1), "row_block_GBps": round(mb / old * 1e3, 0)}), flush=True)
    for name, M, N in [("FP1", 16384, 256), ("FP2", 32768, 288), ("FP2 first", 32768, 256)]:
        y, g = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
        fin = torch.rand(4, N, device=dev) + 0.5
        sums = torch.zeros(2, N, dtype=torch.float64, device=dev)
        new = timeit(lambda: _ext.bn_relu_bwd_prep(y, g, fin, sums=sums))
        ym = misaligned(y)
        old = timeit(lambda: _ext.bn_relu_bwd_prep(ym, g, fin, sums=sums))
        mb = 12 * M * N / 1e6
        print(json.dumps({"kernel": "pn2_bn_relu_bwd_prep", "shape": name, "M": M, "N": N, "alg_MB": round(mb, 1),
                          "vectorised_us": round(new, 1), "vectorised_GBps": round(mb / new * 1e3, 0),
                          "row_block_us": round(old, 1), "row_block_GBps": round(mb / old * 1e3, 0)}), flush=True)


if __name__ == "__main__":
    main()
