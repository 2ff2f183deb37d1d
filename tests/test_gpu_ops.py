"""GPU parity: every C-ABI entry point of libpn2_hip.so against the CPU oracle on
the same seeded inputs.  Index outputs bit-exact; fp32 values bit-exact where the
arithmetic is a single pinned expression, 1e-4 where atomics reorder sums."""
import numpy as np
import pytest
import torch

import oracle_ext
from test_oracle_vs_naive import cloud

pytestmark = pytest.mark.gpu
O = oracle_ext.OracleRowsExt


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pointnet2_ops import _ext
    return _ext


def dev(t):
    return t.cuda()


def clouds(B, N, kind, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(np.stack([cloud(rng, N, kind) for _ in range(B)]))


# every resident-kernel template, the streaming kernel, tiny and non-power-of-two sizes
FPS_CASES = [(3, 1, 1), (2, 5, 5), (2, 37, 40), (3, 100, 30), (2, 255, 64), (2, 256, 64), (2, 511, 100),
             (3, 512, 128), (2, 1000, 128), (2, 1024, 256), (2, 2048, 512), (2, 4000, 512), (2, 4096, 64),
             (2, 8000, 512), (1, 16384, 300), (1, 20000, 256), (1, 24576, 64), (2, 30000, 200),
             (1, 123457, 96)]      # one whole-room cloud (the shape of compute_instance_labels.py:95,195): 16-workgroup cluster


@pytest.mark.parametrize("B,N,m", FPS_CASES)
@pytest.mark.parametrize("kind", ["uniform", "dup", "zero_tail", "grid"])
def test_fps_bit_exact(ext, B, N, m, kind):
    xyz = clouds(B, N, kind, seed=N * 31 + m)
    want = O.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,N,m", [(2, 50000, 300), (3, 40000, 64), (1, 123457, 96), (2, 20000, 100), (2, 4000, 128)])
@pytest.mark.parametrize("kind", ["uniform", "dup", "zero_tail"])
def test_fps_few_cus_hint_is_bit_exact(ext, B, N, m, kind):
    """PN2_FPS_FEW_CUS (geometry prefetched next to a training step: 1024-thread cluster workgroups on half as
    many CUs) only changes the schedule, never the indices."""
    xyz = clouds(B, N, kind, seed=N * 7 + m)
    want = O.furthest_point_sampling(xyz, m)
    for fewest in (False, True):          # PN2_FPS_FEW_CUS, + PN2_FPS_FEWEST_CUS (two 1024-thread workgroups per 50k cloud)
        with ext.background_geometry(fewest=fewest):
            got = ext.furthest_point_sampling(dev(xyz), m).cpu()
        assert torch.equal(got, want)


@pytest.mark.parametrize("mode,g", [("resident", None), ("stream", None), ("coop", 2), ("coop", 8), ("coop", 32), (None, None),
                                    ("bucketed", None), ("coop_plain", 2), ("coop_plain", 8)])
@pytest.mark.parametrize("B,N,m,kind", [(2, 9000, 300, "uniform"), (3, 20000, 150, "dup"), (1, 50000, 200, "zero_tail"),
                                        (32, 50000, 40, "uniform"), (5, 5000, 64, "grid")])
def test_fps_every_kernel_variant_agrees_with_oracle(ext, monkeypatch, mode, g, B, N, m, kind):
    """resident / streaming / cooperative (G workgroups per cloud; with and without the spatially binned points) /
    bucketed (one workgroup per cloud over the binned cloud) kernels are interchangeable: identical indices for
    identical input."""
    plain = mode == "coop_plain"
    mode = "coop" if plain else mode
    if mode == "resident" and N > 24576:
        pytest.skip("does not fit the register file of one workgroup")
    if mode == "coop" and (B * g > 256 or (N + g - 1) // g > 512 * 24):
        pytest.skip("cluster does not fit")
    xyz = clouds(B, N, kind, seed=N + B)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_bucketing(not plain), ext.fps_plan_override(mode=mode, g=g or 0):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()      # (a failed cluster hand-off asserts on the device)
    assert torch.equal(got, want)


@pytest.mark.parametrize("g", [1, 2, 4])
@pytest.mark.parametrize("B,N,m,kind", [(2, 100000, 48, "uniform"), (3, 60000, 64, "dup"), (1, 250000, 40, "zero_tail"),
                                        (64, 30000, 24, "uniform"), (2, 30000, 50, "grid"), (2, 20481, 32, "uniform")])
def test_fps_cluster_with_streamed_tail_agrees_with_oracle(ext, monkeypatch, g, B, N, m, kind):
    """Clouds beyond the register capacity of a cluster (BASELINE stress shape 64 x 200k): 20 points per thread stay
    in registers, the rest is streamed each round (csrc/fps.hip, TAIL) — with a tail, with a one-point tail, and
    with a cloud that fits entirely (empty tail)."""
    if B * g > 256:
        pytest.skip("cluster does not fit")
    xyz = clouds(B, N, kind, seed=N + B + g)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_plan_override(mode="hybrid", g=g):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,N,m,kind", [(2, 100000, 48, "uniform"), (3, 60000, 64, "dup"), (1, 250000, 40, "zero_tail"),
                                        (1, 300001, 24, "uniform"), (64, 30000, 24, "uniform"), (2, 30000, 50, "grid"),
                                        (2, 20481, 32, "uniform"), (3, 700, 40, "uniform"), (2, 64, 64, "dup")])
def test_fps_bucketed_kernel_agrees_with_oracle(ext, B, N, m, kind):
    """One workgroup per cloud over the spatially binned cloud (csrc/fps.hip, fps_bucketed_kernel): buckets of 64, 128,
    256 and 512 records (N up to 65536 S), ragged last buckets, clouds smaller than a workgroup, duplicates and lattice
    ties (the reference's rank decides), skipped points, every point sampled (m = N)."""
    xyz = clouds(B, N, kind, seed=N + B)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_plan_override(mode="bucketed"):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B", [130, 300])
def test_fps_many_large_clouds(ext, B):
    """More clouds than a cluster shape admits: 130 x 30k -> one 1024-thread workgroup per cloud with a streamed tail
    (chosen by the planner itself), 300 x 30k -> the streaming fallback; three clouds checked against the oracle."""
    xyz = clouds(3, 30000, "dup", seed=B).repeat((B + 2) // 3, 1, 1)[:B].contiguous()
    got = ext.furthest_point_sampling(dev(xyz), 40).cpu()
    want = O.furthest_point_sampling(xyz[:3], 40)
    for b in range(B):
        assert torch.equal(got[b], want[b % 3]), b


def test_fps_stress_shape_uses_the_whole_chip(ext):
    """64 x 200k -> 64 (the planner picks the streamed-tail cluster by itself): bit-exact on two of the clouds."""
    xyz = clouds(64, 200000, "uniform", seed=5)
    got = ext.furthest_point_sampling(dev(xyz), 64).cpu()
    for b in (0, 63):
        assert torch.equal(got[b:b + 1], O.furthest_point_sampling(xyz[b:b + 1], 64))


@pytest.mark.parametrize("nc,g", [(1, 8), (2, 16), (4, 32), (2, 4), (4, 8)])
@pytest.mark.parametrize("B,N,m,kind", [(32, 50000, 48, "uniform"), (4, 20000, 100, "dup"), (8, 9000, 64, "zero_tail")])
def test_fps_multi_cloud_clusters_agree_with_oracle(ext, monkeypatch, nc, g, B, N, m, kind):
    """NC clouds per cluster / G workgroups per cluster: same indices as the lane-accurate oracle."""
    if B % nc or (B // nc) * g > 512 or nc * ((N + g * 512 - 1) // (g * 512)) > 16:
        pytest.skip("configuration does not fit")
    xyz = clouds(B, N, kind, seed=N + B + nc)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_plan_override(mode="coop", g=g, nc=nc):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("sub", [2, 1])
@pytest.mark.parametrize("B,N,m,kind", [(2, 50000, 2048, "uniform"), (3, 50000, 700, "dup"), (2, 50000, 600, "zero_tail"),
                                        (2, 30000, 300, "grid"), (32, 50000, 64, "uniform"), (2, 16500, 400, "uniform"),
                                        (3, 20481, 257, "dup"), (2, 53248, 128, "uniform"), (1, 41000, 41000 // 16, "grid")])
def test_fps_several_samples_per_hand_off_agree_with_oracle(ext, sub, B, N, m, kind):
    """fps_multi_kernel (csrc/fps.hip, round 4): two 1024-thread workgroups per cloud exchange the arg-max candidates of
    all sub-blobs per hand-off and accept every further sample that is provably the reference's next one.  64 sub-blobs
    (two per wave) and 32 (one per wave); uniform clouds, duplicates (ties decided by the reference's rank), skipped
    points, lattice ties, long runs (2048 samples = hundreds of hand-offs with 1..8 accepted samples each)."""
    xyz = clouds(B, N, kind, seed=N + B + sub)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_plan_override(mode="multi", nc=sub):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,N,m,kind", [(2, 60000, 500, "uniform"), (2, 100000, 300, "dup"), (1, 106496, 200, "zero_tail"),
                                        (3, 53249, 300, "grid")])
def test_fps_several_samples_four_workgroups_per_cloud(ext, B, N, m, kind):
    """... and four workgroups per cloud (53k < N <= 106k points: one sub-blob per wave, 64 per cloud)."""
    xyz = clouds(B, N, kind, seed=N + B)
    want = O.furthest_point_sampling(xyz, m)
    with ext.fps_plan_override(mode="multi"):
        got = ext.furthest_point_sampling(dev(xyz), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n_nan", [7, 3000])
def test_fps_several_samples_with_nan_points(ext, n_nan):
    """NaN points take part in the reference's sampling (only |p|^2 <= 1e-3 is skipped, sampling_gpu.cu:100-101) and their
    running distance never changes (fminf semantics of min, :106): the sampling keeps returning the NaN point of the
    smallest rank.  A handful of them, and enough to fill whole sub-blobs of the binned cloud (no bounding box at all)."""
    xyz = clouds(2, 40000, "uniform", seed=n_nan)
    g = torch.Generator().manual_seed(n_nan)
    for b in range(2):
        xyz[b, torch.randperm(40000, generator=g)[:n_nan] + 0] = float("nan")
    xyz[0, 0] = torch.tensor([0.3, 0.2, 0.1])               # (a finite first sample in cloud 0, a NaN one possible in cloud 1)
    want = O.furthest_point_sampling(xyz, 40)
    for sub in (2, 1):
        with ext.fps_plan_override(mode="multi", nc=sub):
            got = ext.furthest_point_sampling(dev(xyz), 40).cpu()
        assert torch.equal(got, want), sub
    with ext.fps_multi(False), ext.background_geometry(fewest=True):       # the one-sample cluster over the binned cloud
        assert torch.equal(ext.furthest_point_sampling(dev(xyz), 40).cpu(), want)
    with ext.fps_plan_override(mode="bucketed"):
        assert torch.equal(ext.furthest_point_sampling(dev(xyz), 40).cpu(), want)


def test_fps_multi_is_the_default_for_cluster_sized_clouds_and_can_be_switched_off(ext):
    xyz = clouds(2, 50000, "uniform", seed=77)
    want = O.furthest_point_sampling(xyz, 300)
    for on in (True, False):
        with ext.fps_multi(on):
            assert torch.equal(ext.furthest_point_sampling(dev(xyz), 300).cpu(), want)
            with ext.background_geometry(fewest=True):
                assert torch.equal(ext.furthest_point_sampling(dev(xyz), 300).cpu(), want)
    # degenerate: no valid point at all -> index 0 for every sample (sampling_gpu.cu:90-91), one per hand-off
    z = torch.full((2, 20000, 3), 0.001)
    with ext.fps_plan_override(mode="multi"):
        assert torch.equal(ext.furthest_point_sampling(dev(z), 9).cpu(), torch.zeros(2, 9, dtype=torch.int32))


def test_fps_all_skipped_and_m_zero(ext):
    xyz = torch.full((2, 700, 3), 0.001)
    assert torch.equal(ext.furthest_point_sampling(dev(xyz), 5).cpu(), torch.zeros(2, 5, dtype=torch.int32))
    assert ext.furthest_point_sampling(dev(xyz), 0).shape == (2, 0)


BQ_CASES = [(2, 50, 7, 4, 0.5), (3, 300, 33, 16, 0.3), (2, 1000, 128, 64, 0.4), (2, 4000, 512, 32, 0.2),
            (1, 513, 5, 128, 0.9), (2, 64, 64, 2, 0.01), (2, 2048, 1024, 32, 0.4), (40, 700, 300, 16, 0.25)]


@pytest.mark.parametrize("B,N,m,ns,r", BQ_CASES)
@pytest.mark.parametrize("kind", ["uniform", "dup", "grid"])
def test_ball_query_bit_exact(ext, B, N, m, ns, r, kind):
    xyz = clouds(B, N, kind, seed=N + m)
    sel = torch.from_numpy(np.random.default_rng(5).integers(0, N, size=(B, m)))
    new_xyz = xyz[torch.arange(B)[:, None], sel].contiguous()
    if kind == "uniform":
        new_xyz = clouds(B, m, "uniform", seed=99)        # centres that are not cloud points
    want = O.ball_query(new_xyz, xyz, r, ns)
    got = ext.ball_query(dev(new_xyz), dev(xyz), r, ns).cpu()
    assert torch.equal(got, want)


def test_ball_query_empty_ball_rows_are_zero(ext):
    xyz = torch.rand(2, 100, 3)
    new_xyz = torch.full((2, 9, 3), 50.0)
    assert int(ext.ball_query(dev(new_xyz), dev(xyz), 0.1, 8).abs().sum()) == 0


@pytest.mark.parametrize("B,C,N,m,ns", [(2, 3, 100, 10, 4), (3, 7, 500, 64, 16), (2, 131, 300, 40, 8)])
def test_group_and_gather(ext, B, C, N, m, ns):
    g = torch.Generator().manual_seed(B * C)
    pts = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, m, ns), generator=g, dtype=torch.int32)
    assert torch.equal(ext.group_points(dev(pts), dev(idx)).cpu(), O.group_points(pts, idx))
    go = torch.randn(B, C, m, ns, generator=g)
    torch.testing.assert_close(ext.group_points_grad(dev(go), dev(idx), N).cpu(),
                               O.group_points_grad(go, idx, N), atol=1e-4, rtol=1e-4)
    gi = torch.randint(0, N, (B, m), generator=g, dtype=torch.int32)
    assert torch.equal(ext.gather_points(dev(pts), dev(gi)).cpu(), O.gather_points(pts, gi))
    gg = torch.randn(B, C, m, generator=g)
    torch.testing.assert_close(ext.gather_points_grad(dev(gg), dev(gi), N).cpu(),
                               O.gather_points_grad(gg, gi, N), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,n,m", [(2, 9, 1), (2, 20, 2), (3, 300, 3), (2, 1024, 512), (2, 700, 2500)])
@pytest.mark.parametrize("kind", ["uniform", "grid"])
def test_three_nn_bit_exact(ext, B, n, m, kind):
    u, k = clouds(B, n, kind, seed=n), clouds(B, m, kind, seed=m + 1)
    d2w, iw = O.three_nn(u, k)
    d2, idx = ext.three_nn(dev(u), dev(k))
    assert torch.equal(idx.cpu(), iw)
    assert torch.equal(d2.cpu(), d2w)


@pytest.mark.parametrize("B,C,m,n", [(1, 2, 4, 2), (2, 16, 50, 120), (2, 256, 256, 512)])
def test_three_interpolate(ext, B, C, m, n):
    g = torch.Generator().manual_seed(C)
    pts = torch.randn(B, C, m, generator=g)
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, generator=g)
    assert torch.equal(ext.three_interpolate(dev(pts), dev(idx), dev(w)).cpu(), O.three_interpolate(pts, idx, w))
    go = torch.randn(B, C, n, generator=g)
    torch.testing.assert_close(ext.three_interpolate_grad(dev(go), dev(idx), dev(w), m).cpu(),
                               O.three_interpolate_grad(go, idx, w, m), atol=1e-4, rtol=1e-4)
    rows = pts.transpose(1, 2).contiguous()
    got = ext.three_interpolate_rows(dev(rows), dev(idx), dev(w)).cpu()
    assert torch.equal(got, O.three_interpolate_rows(rows, idx, w))
    gr = go.transpose(1, 2).contiguous()
    torch.testing.assert_close(ext.three_interpolate_rows_grad(dev(gr), dev(idx), dev(w), m, C).cpu(),
                               O.three_interpolate_rows_grad(gr, idx, w, m, C), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,N,m,ns,C", [(2, 100, 10, 4, 3), (2, 400, 50, 16, 0), (3, 300, 20, 8, 128),
                                        # 16-byte batched variants (4 / 2 / 1 rows per wave instruction), ragged row counts
                                        (5, 257, 37, 11, 64), (3, 200, 7, 24, 20), (2, 512, 33, 9, 256), (2, 300, 19, 13, 132),
                                        (7, 64, 5, 8, 36), (2, 128, 9, 7, 128)])
@pytest.mark.parametrize("use_xyz,normalize", [(True, False), (True, True), (False, False)])
def test_group_concat_rows(ext, B, N, m, ns, C, use_xyz, normalize):
    if C == 0 and not use_xyz:
        pytest.skip("no channels")
    g = torch.Generator().manual_seed(N)
    xyz = torch.rand(B, N, 3, generator=g)
    new_xyz = torch.rand(B, m, 3, generator=g)
    feats = torch.randn(B, N, C, generator=g) if C else None
    idx = torch.randint(0, N, (B, m, ns), generator=g, dtype=torch.int32)
    if ns >= 8:                                   # ball-query style padding on half of the neighbourhoods (repeated first hit)
        idx[:, ::2, 3:] = idx[:, ::2, :1]
    want = O.group_concat_rows(xyz, new_xyz, feats, idx, use_xyz, normalize, 0.2)
    got = ext.group_concat_rows(dev(xyz), dev(new_xyz), None if feats is None else dev(feats), dev(idx),
                                use_xyz, normalize, 0.2).cpu()
    assert torch.equal(got, want)
    if C:
        go = torch.randn(*want.shape, generator=g)
        col0 = 3 if use_xyz else 0
        torch.testing.assert_close(ext.group_rows_grad(dev(go), dev(idx), N, C, col0).cpu(),
                                   O.group_rows_grad(go, idx, N, C, col0), atol=1e-4, rtol=1e-4)


# high fan-in scatters (every point receives many rows, half of them the repeated first hit of a padded neighbourhood)
@pytest.mark.parametrize("B,N,m,ns,C,col0", [(32, 512, 256, 16, 256, 3), (16, 1024, 512, 16, 131, 0), (4, 2048, 700, 32, 100, 3),
                                             (64, 300, 128, 16, 70, 0)])
def test_group_rows_grad_high_fan_in(ext, B, N, m, ns, C, col0):
    g = torch.Generator().manual_seed(N + C)
    idx = torch.randint(0, N, (B, m, ns), generator=g, dtype=torch.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]                              # ball-query style padding: repeated first hit
    go = torch.randn(B, m, ns, col0 + C, generator=g)
    want = O.group_rows_grad(go, idx, N, C, col0)
    got = ext.group_rows_grad(dev(go), dev(idx), N, C, col0).cpu()
    torch.testing.assert_close(got, want, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("R,ns,C", [(5, 1, 3), (100, 16, 64), (33, 64, 131)])
def test_rows_max(ext, R, ns, C):
    g = torch.Generator().manual_seed(R)
    x = torch.randn(R, ns, C, generator=g).round(decimals=1)      # plenty of ties
    out, arg = ext.rows_max(dev(x))
    wo, wa = O.rows_max(x)
    assert torch.equal(out.cpu(), wo) and torch.equal(arg.cpu(), wa)
    go = torch.randn(R, C, generator=g)
    assert torch.equal(ext.rows_max_grad(dev(go), arg, ns).cpu(), O.rows_max_grad(go, wa, ns))


@pytest.mark.parametrize("N,E,H", [(3, 3, 2), (9, 72, 256), (576, 4608, 512)])
def test_gcn_rows(ext, N, E, H):
    g = torch.Generator().manual_seed(E)
    x = torch.randn(N, H, generator=g)
    index = torch.randint(0, N, (E,), generator=g)
    assert torch.equal(ext.gather_rows(dev(x), dev(index)).cpu(), O.gather_rows(x, index))
    src = torch.randn(E, H, generator=g)
    want = O.scatter_add_rows(src, index, N)
    torch.testing.assert_close(ext.scatter_add_rows(dev(src), dev(index), N).cpu(), want, atol=1e-4, rtol=1e-4)
    order = torch.sort(index, stable=True).indices
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(index, minlength=N), 0)
    got = ext.segment_sum_rows(dev(src), dev(order), dev(rowptr), N).cpu()
    assert torch.equal(got, want)                                  # deterministic == sequential CPU order
    wide = torch.zeros(E, 3 * H + 5)
    ext_out = ext.gather_rows(dev(x), dev(index), out=dev(wide), col0=H + 5).cpu()
    assert torch.equal(ext_out[:, H + 5:2 * H + 5], O.gather_rows(x, index)) and ext_out[:, :H + 5].abs().sum() == 0


def test_errors_are_exceptions_not_exits(ext):
    with pytest.raises(RuntimeError):
        ext.gather_rows(dev(torch.zeros(3, 4)), dev(torch.tensor([0, 7])))     # index out of range
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.three_nn(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3))


# inverse neighbourhood index + CSR sum (csrc/group_csr.hip): the atomic-free, bit-reproducible form of the feature-gradient
# scatter (group_points_grad_kernel, src/group_points_gpu.cu:44-75)
@pytest.mark.parametrize("B,N,m,ns,C,col0", [(32, 512, 256, 16, 256, 3), (16, 1024, 512, 16, 131, 0), (4, 2048, 700, 32, 100, 3),
                                             (64, 300, 128, 16, 64, 0), (3, 97, 11, 5, 128, 3), (2, 50, 4, 9, 36, 1),
                                             (1, 5000, 40, 70, 192, 3), (2, 64, 64, 64, 7, 0)])
def test_group_rows_grad_csr(ext, B, N, m, ns, C, col0):
    g = torch.Generator().manual_seed(N + C)
    idx = torch.randint(0, N, (B, m, ns), generator=g, dtype=torch.int32)
    idx[:, ::2, ns // 2:] = idx[:, ::2, :1]                          # ball-query style padding on half of the neighbourhoods
    go = torch.randn(B, m, ns, col0 + C, generator=g)
    ptr, refs = ext.group_inverse_index(dev(idx), N)
    # the index itself: refs sorted by (b*N + idx[row], row), ptr = offsets of the points' row lists
    keys = (idx.long() + torch.arange(B).view(B, 1, 1) * N).flatten()
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(refs.cpu().long(), order)
    want_ptr = torch.zeros(B * N + 1, dtype=torch.long)
    want_ptr[1:] = torch.cumsum(torch.bincount(keys, minlength=B * N), 0)
    assert torch.equal(ptr.cpu().long(), want_ptr)
    want = O.group_rows_grad(go, idx, N, C, col0)
    got = ext.group_rows_grad_csr(dev(go), (ptr, refs), N, C, col0)
    torch.testing.assert_close(got.cpu(), want, atol=2e-4, rtol=1e-4)
    # fixed summation order: identical bits on a second run, and equal to a sequential sum in row order
    assert torch.equal(ext.group_rows_grad_csr(dev(go), (ptr, refs), N, C, col0), got)
    with pytest.raises(RuntimeError):
        ext.group_rows_grad_csr(dev(go), (ptr[:-1].contiguous(), refs), N, C, col0)


def test_group_rows_grad_csr_empty_and_unreferenced_points(ext):
    idx = torch.zeros(2, 3, 4, dtype=torch.int32)                     # every row gathers point 0 of its cloud
    go = torch.ones(2, 3, 4, 8)
    ptr, refs = ext.group_inverse_index(dev(idx), 10)
    got = ext.group_rows_grad_csr(dev(go), (ptr, refs), 10, 8, 0).cpu()
    assert torch.equal(got[:, 0], torch.full((2, 8), 12.0)) and float(got[:, 1:].abs().max()) == 0.0


def test_new_entry_points_reject_bad_arguments(ext):
    """C-ABI argument checks of the round-2 entry points: negative codes, no launch, no exit."""
    lib = ext._lib
    idx = torch.zeros(2, 3, 4, dtype=torch.int32).cuda()
    ptr = torch.empty(2 * 10 + 1, dtype=torch.int32).cuda()
    refs = torch.empty(24, dtype=torch.int32).cuda()
    need = int(lib.pn2_group_inverse_index_workspace_bytes(2, 10, 3, 4))
    assert need > 0
    ws = torch.empty(need + 512, dtype=torch.uint8).cuda()
    p = lambda t: t.data_ptr()
    assert lib.pn2_group_inverse_index(2, 10, 3, 4, p(idx), p(ptr), p(refs), p(ws), need - 1, None) == -4      # PN2_ENOSPC
    assert lib.pn2_group_inverse_index(2, 10, 3, 4, p(idx), p(ptr), p(refs), p(ws) + 4, need, None) == -1      # alignment
    assert lib.pn2_group_inverse_index(2, 10, 3, 4, p(idx), None, p(refs), p(ws), need, None) == -2            # PN2_ENULL
    assert lib.pn2_group_inverse_index(-1, 10, 3, 4, p(idx), p(ptr), p(refs), p(ws), need, None) == -1
    go = torch.zeros(2, 3, 4, 8).cuda()
    out = torch.empty(2, 10, 8).cuda()
    assert lib.pn2_group_rows_grad_csr(2, 10, 8, 4, 0, 24, p(go), p(ptr), p(refs), p(out), None) == -1         # ldg < col0 + C
    assert lib.pn2_group_rows_grad_csr(2, 10, 8, 8, 0, 24, p(go), None, p(refs), p(out), None) == -2
    gb = torch.zeros(2, 3, 4, 7, dtype=torch.bfloat16).cuda()
    assert lib.pn2_group_rows_grad_bf16(2, 10, 3, 4, 7, 7, 0, p(gb), p(idx), p(out), None) == -1                # odd C
    q, pp = torch.zeros(5, 6).cuda(), torch.zeros(3, 12).cuda()
    ia = torch.zeros(5, dtype=torch.int64).cuda()
    assert lib.pn2_gather2_add_rows(5, 6, 3, 12, 0, 6, p(pp), p(ia), p(ia), p(q), None) == -1                   # H % 4 != 0
    with pytest.raises(RuntimeError):
        ext.group_rows_grad_csr(go, (ptr, refs[:-1].contiguous()), 10, 8, 0)
