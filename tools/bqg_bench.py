"""Ball query + grouping at the SA1 micro shape of SURVEY.md 8d (32 x 50k points, 2048 centres, r 0.2, 64 samples, C 3):
the fused kernel (pn2_ball_query_group) for both slab widths, the plain slab query, and the two-kernel pair, ms per call.
`python tools/bqg_bench.py [iters]`; a short loop for counter passes: `bash tools/pmc_kernel.sh bqg bq_slab_query -- python
tools/bqg_bench.py 3`."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch  # noqa: E402
from microbench import timeit, unit_ball  # noqa: E402
from pointnet2_ops import _ext  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N, m, r, ns, C = 32, 50000, 2048, 0.2, 64, 3
xyz = unit_ball(B, N).cuda()
feats = torch.rand(B, N, C, device="cuda")
sel = _ext.furthest_point_sampling(xyz, m)
new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = _ext.ball_query(new_xyz, xyz, r, ns)
fused = B * (12 * N + 12 * m + 4 * C * N + 4 * (3 + C) * m * ns + 4 * m * ns)
rows = [("ball_query (slabs, auto width)", lambda: _ext.ball_query(new_xyz, xyz, r, ns)),
        ("group_concat_rows", lambda: _ext.group_concat_rows(xyz, new_xyz, feats, idx, True, True, r))]
for w in (1, 4):
    rows.append((f"ball_query_group slab_w={w}", lambda w=w: _ext.ball_query_group(new_xyz, xyz, feats, r, ns, True, True, slab_w=w)))
for name, fn in rows:
    t = timeit(fn, iters=iters, warm=2)
    print(f"{name:34s} {t * 1e3:8.4f} ms   {fused / t / 1e9:8.1f} GB/s of the fused byte count ({fused / 1e6:.1f} MB) = {fused / t / 8e12:.3f} of 8 TB/s")
