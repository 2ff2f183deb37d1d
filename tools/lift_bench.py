"""First layer before the grouping (csrc/group_lift.hip) at the headline SA2 / SA3 / SA4 shapes: pn2_group_lift_rows and
pn2_group_lift_rows_grad alone, ms per call and GB/s of their algorithmic bytes; `python tools/lift_bench.py [iters]`.
Counter passes: `bash tools/pmc_kernel.sh lift group_lift -- python $GRAFT_REPO_ROOT/tools/lift_bench.py 3`."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch  # noqa: E402
from microbench import timeit, unit_ball  # noqa: E402
from pointnet2_ops import _ext as e  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32
for name, N, m, r, ns, C, N0 in (("SA2", 2048, 1024, 0.4, 32, 128, 128), ("SA3", 1024, 512, 0.8, 16, 256, 128),
                                 ("SA4", 512, 256, 1.2, 16, 256, 128)):
    xyz = unit_ball(B, N).cuda()
    new_xyz = xyz[:, :m].contiguous()
    idx = e.ball_query(new_xyz, xyz, r, ns)
    inv = e.group_inverse_index(idx, N)
    P = torch.randn(B, N, N0, device="cuda")
    Wx = torch.randn(N0, 3, device="cuda")
    stats = torch.zeros(2, N0, dtype=torch.float64, device="cuda")
    M = B * m * ns
    t = timeit(lambda: e.group_lift_rows(P, xyz, new_xyz, idx, Wx, True, r, stats=stats), iters=iters)
    nb = B * (4 * m * ns + 12 * N + 12 * m + 4 * N0 * N + 4 * N0 * m * ns)
    print(f"{name} group_lift_rows       {t * 1e3:8.4f} ms  {nb / t / 1e9:8.1f} GB/s ({nb / 1e6:.0f} MB)")
    G = torch.randn(M, N0, device="cuda")
    acc = torch.zeros(3 * N0 + 9, device="cuda")
    consts = torch.randn(3, N0, device="cuda")
    t = timeit(lambda: e.group_lift_rows_grad(G, P, Wx, consts, xyz, new_xyz, inv, ns, True, r, acc), iters=iters)
    nb = 8 * M + 4 * M * N0 + 4 * B * N * (2 * N0 + 4)
    print(f"{name} group_lift_rows_grad  {t * 1e3:8.4f} ms  {nb / t / 1e9:8.1f} GB/s ({nb / 1e6:.0f} MB)")
    gr = G.view(B, m, ns, N0)
    t = timeit(lambda: e.group_rows_grad_csr(gr, inv, N, N0, 0), iters=iters)
    print(f"{name} group_rows_grad_csr   {t * 1e3:8.4f} ms  (the plain per-point sum of the same rows)")
