"""Grouping kernels (pn2_group_concat_rows, its bf16 form, pn2_group_rows_grad) at the headline SA2-SA4 shapes (crowded
balls) and at the scene-graph encoder shapes (many small clouds, sparse balls).  One JSON line per (kernel, shape):
time, algorithmic bytes, GB/s.     python tools/group_bench.py            (GPU box)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]

import torch  # noqa: E402
from pointnet2_ops import _ext  # noqa: E402

SHAPES = [  # name, B, N, m, ns, C, radius (points uniform in the unit cube)
    ("headline SA2", 32, 2048, 1024, 32, 128, 0.25),
    ("headline SA3", 32, 1024, 512, 16, 256, 0.3),
    ("headline SA4", 32, 512, 256, 16, 256, 0.4),
    ("sgp8 rel L2 ns64", 576, 512, 128, 64, 192, 0.12),
    ("sgp8 rel L2 ns32", 576, 512, 128, 32, 192, 0.08),
    ("sgp8 obj L2 ns64", 72, 512, 128, 64, 192, 0.12),
]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = "cuda"
    for name, B, N, m, ns, C, r in SHAPES:
        g = torch.Generator().manual_seed(B + N)
        xyz = torch.rand(B, N, 3, generator=g).to(dev)
        new_xyz = xyz[:, :m].contiguous()
        idx = _ext.ball_query(new_xyz, xyz, r, ns)
        distinct = float((idx != idx[:, :, :1]).sum(-1).float().mean()) + 1
        feats = torch.randn(B, N, C, generator=g).to(dev)
        W = 3 + C
        rows = B * m * ns
        gbytes = rows * (4 + 4 * W) + B * N * (12 + 4 * C) + B * m * 12
        t = timeit(lambda: _ext.group_concat_rows(xyz, new_xyz, feats, idx, True, True, r))
        print(json.dumps(dict(kernel="group_concat_rows", shape=name, B=B, N=N, m=m, ns=ns, C=C, distinct_hits=round(distinct, 1),
                              ms=round(t * 1e3, 4), alg_MB=round(gbytes / 1e6, 1), GBps=round(gbytes / t / 1e9, 1))), flush=True)
        if _ext.HAS_BF16_MLP:
            ldo = (W + 7) // 8 * 8
            bbytes = rows * (4 + 2 * ldo) + B * N * (12 + 4 * C) + B * m * 12
            t = timeit(lambda: _ext.group_concat_rows_bf16(xyz, new_xyz, feats, idx, True, True, r))
            print(json.dumps(dict(kernel="group_concat_rows_bf16", shape=name, ms=round(t * 1e3, 4), alg_MB=round(bbytes / 1e6, 1),
                                  GBps=round(bbytes / t / 1e9, 1))), flush=True)
        go = torch.randn(B, m, ns, W, generator=g).to(dev)
        sbytes = rows * (4 + 4 * C) + 2 * B * N * 4 * C
        t = timeit(lambda: _ext.group_rows_grad(go, idx, N, C, 3))
        print(json.dumps(dict(kernel="group_rows_grad", shape=name, ms=round(t * 1e3, 4), alg_MB=round(sbytes / 1e6, 1),
                              GBps=round(sbytes / t / 1e9, 1))), flush=True)
        if _ext.HAS_BF16_MLP and C % 8 == 0:
            gob = torch.randn(B, m, ns, C, generator=g).to(torch.bfloat16).to(dev)
            bb = rows * (4 + 2 * C) + 2 * B * N * 4 * C
            t = timeit(lambda: _ext.group_rows_grad(gob, idx, N, C, 0))
            print(json.dumps(dict(kernel="group_rows_grad_bf16", shape=name, ms=round(t * 1e3, 4), alg_MB=round(bb / 1e6, 1),
                                  GBps=round(bb / t / 1e9, 1))), flush=True)
        t = timeit(lambda: _ext.group_inverse_index(idx, N))
        print(json.dumps(dict(kernel="group_inverse_index", shape=name, ms=round(t * 1e3, 4), rows=rows)), flush=True)
        inv = _ext.group_inverse_index(idx, N)
        t = timeit(lambda: _ext.group_rows_grad_csr(go, inv, N, C, 3))
        print(json.dumps(dict(kernel="group_rows_grad_csr", shape=name, ms=round(t * 1e3, 4), alg_MB=round(sbytes / 1e6, 1),
                              GBps=round(sbytes / t / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
