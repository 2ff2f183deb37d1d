"""Forward + backward time of one fused shared-MLP stack (group rows -> 3 layers -> max over ns) against the number of
rows, fp32 and bf16: the fixed cost per stack call (what a per-scan call of the exact mode pays) next to the streaming
regime.      python tools/stack_small_m.py      (GPU box; one JSON line per (spec, M))"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch  # noqa: E402
from pointnet2_ops import fused_mlp, pointnet2_modules as pm  # noqa: E402


def timeit(fn, iters=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3


def main():
    specs = (((195, 128, 128, 256), 64), ((64, 64, 64, 128), 32), ((259, 256, 256, 256), 16))
    sizes = (16384, 65536, 262144, 1048576)
    if len(sys.argv) > 2:                      # one case (for a rocprofv3 --stats run): <spec index> <M>
        specs, sizes = (specs[int(sys.argv[1])],), (int(sys.argv[2]),)
    for spec, ns in specs:
        torch.manual_seed(0)
        mlp = pm.build_shared_mlp(list(spec)).cuda().train()
        for M in sizes:
            x = torch.randn(M, spec[0], device="cuda", requires_grad=True)
            row = {"spec": list(spec), "ns": ns, "M": M}
            for dt in ("f32", "bf16"):
                prev = fused_mlp.set_mlp_dtype(dt)

                def step():
                    out = fused_mlp.fused_shared_mlp(mlp, x, ns)
                    out.sum().backward()
                row[dt + "_us"] = round(timeit(step), 1)
                fused_mlp.set_mlp_dtype(prev)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
