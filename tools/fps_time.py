"""Time FPS at the bench shapes; workgroup widths of the resident kernel are forced through _ext.fps_plan_override."""
import contextlib, os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball, timeit
dev = torch.device("cuda:0")
BS = [None, 256, 512, 1024]
cases = [(32, 50000, 2048, [None]), (32, 2048, 1024, BS), (32, 1024, 512, BS), (32, 512, 256, BS),
         (72, 8000, 512, BS), (9, 4000, 512, BS), (32, 16384, 1024, BS), (8, 200000, 512, [None])]
for B, N, m, widths in cases:
    x = unit_ball(B, N, 3).to(dev)
    for bs in widths:
        ctx = _ext.fps_plan_override("resident", bs=bs) if bs else contextlib.nullcontext()
        try:
            with ctx:
                t = timeit(lambda: _ext.furthest_point_sampling(x, m), iters=5, warm=1)
            print(json.dumps(dict(B=B, N=N, m=m, bs=bs, ms=round(t * 1e3, 3), us_per_round=round(t * 1e6 / m, 3))), flush=True)
        except RuntimeError as e:
            print(json.dumps(dict(B=B, N=N, m=m, bs=bs, error=str(e)[:80])), flush=True)
