"""Device time of pn2_group_inverse_index by slice count (PN2_INVERSE_INDEX_SLICES) and of the radix route, at the level shapes
(100 calls inside one HIP graph: no host time in the figure)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "4d-or_amd"))
from pointnet2_ops import _ext

def timeit(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for B, N, m, ns in [(32, 50000, 2048, 64), (8, 50000, 2048, 64), (8, 2048, 1024, 32), (8, 1024, 512, 16), (32, 8000, 512, 32), (32, 4000, 512, 16), (8, 20000, 2048, 64), (8, 36864, 2048, 64)]:
    idx = torch.randint(0, N, (B, m, ns), dtype=torch.int32, device="cuda")
    row = {}
    for s in ["", "1", "2", "4", "8", "12", "16", "24", "32", "48", "64"]:
        if s: os.environ["PN2_INVERSE_INDEX_SLICES"] = s
        else: os.environ.pop("PN2_INVERSE_INDEX_SLICES", None)
        try:
            row[s or "default"] = round(timeit(lambda: _ext.group_inverse_index(idx, N)), 1)
        except Exception as ex:
            row[s] = str(ex)[:30]
    os.environ.pop("PN2_INVERSE_INDEX_SLICES", None)
    os.environ["PN2_INVERSE_INDEX_RADIX"] = "1"
    try:
        row["radix"] = round(timeit(lambda: _ext.group_inverse_index(idx, N)), 1)
    except Exception as ex:
        row["radix"] = str(ex)[:40]
    os.environ.pop("PN2_INVERSE_INDEX_RADIX")
    print((B, N, m, ns), row, flush=True)
