"""Ball query: the three algorithms of pn2_ball_query_algo (scan / per-cloud cell list / slab cell lists) side by side.

Clouds fill the unit ball (zero_mean normalisation), centres are a random subset.  ms per call (events, 20 calls after 3
warm-ups); "auto" marks the library's choice for the shape.  `python tools/bq_bench.py [headline|sgp|all]`."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/4d-or_amd")
from pointnet2_ops import _ext  # noqa: E402

SHAPES = {
    "headline": [(32, 50000, 2048, 0.2, 64), (32, 2048, 1024, 0.4, 32), (32, 1024, 512, 0.8, 16), (32, 512, 256, 1.2, 16)],
    "sgp": [(72, 8000, 512, 0.1, 16), (72, 8000, 512, 0.2, 32), (72, 8000, 512, 0.4, 128), (72, 512, 128, 0.2, 32),
            (72, 512, 128, 0.4, 64), (72, 512, 128, 0.8, 128), (288, 4000, 512, 0.1, 16), (288, 4000, 512, 0.2, 32),
            (288, 4000, 512, 0.4, 128)],
    "sweep": [(32, n, 2048, r, ns) for n in (4096, 8192, 16384, 50000) for (r, ns) in ((0.1, 16), (0.2, 32), (0.2, 64), (0.4, 64))],
}


def cloud(B, N, m, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    xyz = p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g) ** (1 / 3)
    return xyz[:, torch.randperm(N, generator=g)[:m]].contiguous().cuda(), xyz.cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    shapes = sum(SHAPES.values(), []) if which == "all" else SHAPES[which]
    print(f"{'B':>4} {'N':>6} {'m':>5} {'r':>5} {'ns':>4} | {'scan':>8} {'cells':>8} {'slabs':>8} | auto")
    for B, N, m, r, ns in shapes:
        c, x = cloud(B, N, m)
        row, ref = [], None
        for mode in (False, "cells", "slabs"):
            _ext.BALL_QUERY_GRID = mode
            if mode and not _ext._lib.pn2_ball_query_algo_bytes({"cells": 1, "slabs": 2}[mode], B, N, m, r, ns):
                row.append(float("nan"))
                continue
            out = _ext.ball_query(c, x, r, ns)
            ref = out if ref is None else ref
            assert torch.equal(out, ref), (mode, B, N, m, r, ns)
            row.append(timed(lambda: _ext.ball_query(c, x, r, ns)))
        _ext.BALL_QUERY_GRID = True
        auto = ("scan", "cells", "slabs")[_ext._lib.pn2_ball_query_auto(B, N, m, r, ns)]
        print(f"{B:>4} {N:>6} {m:>5} {r:>5} {ns:>4} | {row[0]:8.4f} {row[1]:8.4f} {row[2]:8.4f} | {auto}")


if __name__ == "__main__":
    main()
