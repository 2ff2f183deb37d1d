"""Time FPS at the bench shapes (PN2_FPS_G / PN2_FPS_MODE overrides are read per call by the library)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball, timeit
dev = torch.device("cuda:0")
BS = [None, "256", "512", "1024"]
cases = [(32, 50000, 2048, [None]), (32, 2048, 1024, BS), (32, 1024, 512, BS), (32, 512, 256, BS),
         (72, 8000, 512, BS), (9, 4000, 512, BS), (32, 16384, 1024, BS), (8, 200000, 512, [None])]
for B, N, m, gs in cases:
    x = unit_ball(B, N, 3).to(dev)
    for g in gs:
        if g: os.environ["PN2_FPS_BS"] = g
        else: os.environ.pop("PN2_FPS_BS", None)
        os.environ["PN2_FPS_CHECK"] = "1"
        _ext.furthest_point_sampling(x, m)
        os.environ.pop("PN2_FPS_CHECK")
        t = timeit(lambda: _ext.furthest_point_sampling(x, m), iters=5, warm=1)
        print(json.dumps(dict(B=B, N=N, m=m, G=g, ms=round(t * 1e3, 3), us_per_round=round(t * 1e6 / m, 3))), flush=True)
