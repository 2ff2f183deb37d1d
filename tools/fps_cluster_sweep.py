"""FPS round time of the cluster kernels against cloud size and cluster shape (G workgroups of 512 / 1024 threads per
cloud, forced through _ext.fps_plan_override); PN2_FPS_BUCKETING=0 switches the spatial binning off for an A/B."""
import contextlib, os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball, timeit
dev = torch.device("cuda:0")
for N in (4096, 8192, 16384, 32768, 50000):
    x = unit_ball(32, N, 3).to(dev)
    for g, bs in ((2, 1024), (4, 512), (4, 1024)):
        try:
            with _ext.fps_plan_override("coop", g=g, nc=1, coop_bs=bs):
                t = timeit(lambda: _ext.furthest_point_sampling(x, 2048), iters=3, warm=1)
            print(json.dumps(dict(N=N, G=g, bs=bs, ms=round(t*1e3,3), us_per_round=round(t*1e6/2048,3))), flush=True)
        except RuntimeError as e:
            print(json.dumps(dict(N=N, G=g, bs=bs, error=str(e)[:60])), flush=True)
