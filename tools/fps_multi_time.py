"""Cluster FPS at the headline shape: one sample per hand-off (round 3) against several (fps_multi_kernel, round 4),
alone and with the scheduling hint of the geometry prefetch."""
import contextlib, os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball, timeit
dev = torch.device("cuda:0")
shapes = [(32, 50000, 2048), (32, 50000, 512), (8, 50000, 2048), (32, 20000, 1024), (16, 100000, 2048), (72, 8000, 512)]
for B, N, m in shapes:
    x = unit_ball(B, N, 3).to(dev)
    ref = None
    for name, ctx in [("default", contextlib.nullcontext), ("multi sub=2", lambda: _ext.fps_plan_override("multi", nc=2)),
                      ("multi sub=1", lambda: _ext.fps_plan_override("multi", nc=1)),
                      ("one-sample cluster", lambda: _ext.fps_multi(False)),
                      ("one-sample, fewest CUs", lambda: contextlib.ExitStack()),
                      ("bucketed (1 wg / cloud)", lambda: _ext.fps_plan_override("bucketed"))]:
        try:
            with ctx() as st:
                if name.endswith("fewest CUs"):
                    st.enter_context(_ext.fps_multi(False)); st.enter_context(_ext.background_geometry(fewest=True))
                out = _ext.furthest_point_sampling(x, m)
                if ref is None: ref = out
                same = bool(torch.equal(out, ref))
                t = timeit(lambda: _ext.furthest_point_sampling(x, m), iters=5, warm=1)
            print(json.dumps(dict(B=B, N=N, m=m, variant=name, ms=round(t * 1e3, 3), us_per_sample=round(t * 1e6 / m, 3), same=same)), flush=True)
        except RuntimeError as e:
            print(json.dumps(dict(B=B, N=N, m=m, variant=name, error=str(e)[:100])), flush=True)
