"""GPU cases added in round 5.

* the vectorised backward-prep kernels (`prep_vec_kernel`: pn2_pool_bwd_prep / pn2_bn_relu_bwd_prep / the segment-table
  form) against a float64 restatement, ragged shapes included, and segment table == one call per scan, bit for bit;
* a TripletGCN scan longer than the fused kernels cover comes back NaN (not a silent partial result); the FPS
  status word is asked for with the flags of the call; pre-grouped rows of a prefetched geometry are dropped when the
  features changed in between;
* VERDICT r04 parity hygiene (b): ONE oracle comparison at B = 32 for the XCD-aware paths — FPS 32 x 50k -> 256 and ball
  query 32 x 50k / 2048 / r 0.2 / ns 64, four of the 32 clouds each through the CPU oracle.
"""
import pytest
from conftest import assert_same_product
import torch

import oracle_ext
from pointnet2_ops import _ext
from pointnet2_ops import pointnet2_modules as pm
from pointnet2_ops import pointnet2_utils as pu

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _unit_ball(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)
    p = p - p.mean(dim=1, keepdim=True)
    return (p / p.norm(dim=2).amax(dim=1).view(B, 1, 1)).contiguous()


def _fin(C, seed):
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(C, generator=g) * 0.3
    rstd = torch.rand(C, generator=g) + 0.5
    scale = (torch.rand(C, generator=g) - 0.3) * rstd            # some negative gammas
    shift = torch.randn(C, generator=g) * 0.2 - mean * scale
    return torch.stack([mean, rstd, scale, shift]).contiguous().to(DEV)


# ------------------------------------------------------------------------------------------------ prep kernels
@pytest.mark.parametrize("R,C", [(65536, 128), (32768, 256), (1000, 288), (37, 64), (70001, 32), (513, 4), (300, 130)])
def test_pool_bwd_prep_matches_float64(R, C):
    g = torch.Generator().manual_seed(R + C)
    yraw = torch.randn(R, C, generator=g).to(DEV)
    pooled = torch.relu(torch.randn(R, C, generator=g)).to(DEV)          # about half the entries are 0 (ReLU'd maxima)
    gP = torch.randn(R, C, generator=g).to(DEV)
    fin = _fin(C, 3)
    gPm, sums = _ext.pool_bwd_prep(yraw, pooled, gP, fin)
    ref = torch.where(pooled > 0, gP, torch.zeros_like(gP))
    assert torch.equal(gPm, ref)
    yhat = (yraw.double() - fin[0].double()) * fin[1].double()
    torch.testing.assert_close(sums[0], ref.double().sum(0), rtol=1e-5, atol=1e-5 * R ** 0.5)
    torch.testing.assert_close(sums[1], (ref.double() * yhat).sum(0), rtol=1e-5, atol=1e-5 * R ** 0.5)
    # accumulates: a second call doubles the sums
    _, sums2 = _ext.pool_bwd_prep(yraw, pooled, gP, fin, sums=sums.clone())
    torch.testing.assert_close(sums2, 2 * sums, rtol=1e-12, atol=0)


@pytest.mark.parametrize("M,N", [(32768, 256), (32768, 288), (999, 128), (70001, 64), (50, 513)])
def test_bn_relu_bwd_prep_matches_float64(M, N):
    g = torch.Generator().manual_seed(M + N)
    y = torch.randn(M, N, generator=g).to(DEV)
    gout = torch.randn(M, N, generator=g).to(DEV)
    fin = _fin(N, 5)
    gpre, sums = _ext.bn_relu_bwd_prep(y, gout, fin)
    gate = torch.addcmul(fin[3], y, fin[2]) > 0                       # fma(y, scale, shift): one rounding, like the kernel
    gate64 = (y.double() * fin[2].double() + fin[3].double()) > 0
    agree = gate == gate64                                            # (entries within an ulp of the ReLU kink may differ)
    ref = torch.where(gate64, gout, torch.zeros_like(gout))
    assert float((gpre != ref)[agree].sum()) == 0 and float((~agree).float().mean()) < 1e-5
    yhat = (y.double() - fin[0].double()) * fin[1].double()
    torch.testing.assert_close(sums[0], gpre.double().sum(0), rtol=1e-5, atol=1e-5 * M ** 0.5)
    torch.testing.assert_close(sums[1], (gpre.double() * yhat).sum(0), rtol=1e-5, atol=1e-5 * M ** 0.5)


@pytest.mark.parametrize("C", [128, 288, 130])
def test_pool_bwd_prep_segment_table_equals_one_call_per_scan(C):
    ns = 16
    rows = [64 * ns, 1 * ns, 700 * ns, 33 * ns]                       # un-pooled rows per scan
    R = sum(rows) // ns
    g = torch.Generator().manual_seed(11)
    yraw = torch.randn(R, C, generator=g).to(DEV)
    pooled = torch.relu(torch.randn(R, C, generator=g)).to(DEV)
    gP = torch.randn(R, C, generator=g).to(DEV)
    fin = torch.stack([_fin(C, 20 + s) for s in range(len(rows))]).contiguous()
    seg = _ext.SegTable.get(torch.device(DEV, torch.cuda.current_device()), rows)
    gPm, sums = _ext.pool_bwd_prep(yraw, pooled, gP, fin, seg=seg, ns=ns)
    o = 0
    for s, r in enumerate(rows):
        n = r // ns
        a, b = _ext.pool_bwd_prep(yraw[o:o + n].contiguous(), pooled[o:o + n].contiguous(), gP[o:o + n].contiguous(), fin[s].contiguous())
        assert torch.equal(gPm[o:o + n], a)
        assert torch.equal(sums[s], b), f"scan {s}: the table's sums differ from the scan's own call"
        o += n


# ------------------------------------------------------------------------------------------------ ADVICE r04
def test_gcn_scan_longer_than_the_fused_kernels_cover_is_nan_not_partial():
    S, dn = 2, 64
    rows = [40, 200]                                                  # the second scan exceeds the 128 rows a workgroup covers
    ptr = torch.tensor([0, rows[0], sum(rows)], dtype=torch.int64, device=DEV)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(sum(rows), dn, generator=g).to(DEV)
    W = torch.randn(dn, dn, generator=g).to(DEV)
    bias = torch.zeros(dn, device=DEV)
    gamma, beta = torch.ones(dn, device=DEV), torch.zeros(dn, device=DEV)
    assert not _ext.gcn_fused_supported(dn, dn, dn, max(rows))
    out, ypre, mean, rstd = _ext.gcn_linear(W, bias, ptr, S, A=A, bn=(gamma, beta, 1e-5), relu=True)
    assert bool(torch.isfinite(out[:rows[0]]).all())                  # the scan that fits is computed
    assert bool(torch.isnan(out[rows[0]:]).all()) and bool(torch.isnan(ypre[rows[0]:]).all())


def test_fps_status_offset_follows_the_flags_of_the_call():
    lib = _ext._lib
    for B, N, m in [(32, 50000, 2048), (8, 20000, 512), (32, 2048, 1024), (4, 24000, 256), (64, 17000, 300)]:
        assert lib.pn2_fps_status_offset_ex(B, N, m, 0) == lib.pn2_fps_status_offset(B, N, m)
        for flags in (_ext.PN2_FPS_FEW_CUS, _ext.PN2_FPS_FEW_CUS | _ext.PN2_FPS_FEWEST_CUS):
            off = lib.pn2_fps_status_offset_ex(B, N, m, flags)
            assert off == -1 or off > 0                                 # a byte offset, or "this plan writes no word"
    assert lib.pn2_fps_status_offset_ex(4, 1000, 16, 0x40) == -1     # unknown flag bits
    # the flagged calls run clean whatever plan they pick (a spurious device assert would poison the context here)
    xyz = _unit_ball(6, 20000, 3).to(DEV)
    ref = _ext.furthest_point_sampling(xyz, 300)
    with _ext.background_geometry(fewest=True):
        assert torch.equal(_ext.furthest_point_sampling(xyz, 300), ref)
    with _ext.background_geometry(fewest=False):
        assert torch.equal(_ext.furthest_point_sampling(xyz, 300), ref)
    torch.cuda.synchronize()


def test_prefetched_rows_are_dropped_when_the_features_changed():
    torch.manual_seed(0)
    # (the headline's SA1 geometry: crowded balls in a large cloud, where the query kernel emits the grouped rows itself)
    sa = pm.PointnetSAModule(mlp=[3, 32, 64], npoint=512, radius=0.2, nsample=64).to(DEV).train()
    xyz = _unit_ball(2, 50000, 5).to(DEV)
    feats_rows = torch.rand(2, 50000, 3, device=DEV)
    geo = sa.sample_and_query(xyz, feats_rows=feats_rows)
    if geo["rows"][0] is None:
        pytest.skip("the library groups this shape with the two-kernel route: no pre-grouped rows to go stale")
    features = feats_rows.transpose(1, 2)                              # (B, C, N) view of the same storage
    assert geo["rows"][0] is not None and pm.rows_still_valid(geo, pu.as_rows(features))
    with torch.no_grad():
        _, same = sa(xyz, features, geometry=geo)
        _, plain = sa(xyz, features)
        torch.testing.assert_close(same, plain, atol=1e-5, rtol=1e-5)
        feats_rows.mul_(0.5)                                           # augmentation after the prefetch
        assert not pm.rows_still_valid(geo, pu.as_rows(features))
        _, stale_guarded = sa(xyz, features, geometry=geo)             # rows dropped: grouped from the CURRENT features
        _, fresh = sa(xyz, features)
    torch.testing.assert_close(stale_guarded, fresh, atol=1e-5, rtol=1e-5)
    assert float((fresh - plain).abs().max()) > 1e-3                   # (the change was visible in the result)


# ------------------------------------------------------------------------------------------------ B = 32 vs the oracle
def test_fps_at_32_clouds_matches_the_oracle_on_four_of_them():
    xyz = _unit_ball(32, 50000, 41)
    got = _ext.furthest_point_sampling(xyz.to(DEV), 256).cpu()
    with _ext.background_geometry(fewest=True):                        # the shape the bench's prefetch stream runs
        got_bg = _ext.furthest_point_sampling(xyz.to(DEV), 256).cpu()
    assert torch.equal(got, got_bg)
    pick = [0, 9, 18, 31]                                              # clouds of different XCDs / clusters
    ref = oracle_ext.OracleRowsExt.furthest_point_sampling(xyz[pick].contiguous(), 256)
    assert torch.equal(got[pick], ref)


def test_ball_query_at_32_clouds_matches_the_oracle_on_four_of_them():
    B, N, m, ns, r = 32, 50000, 2048, 64, 0.2
    xyz = _unit_ball(B, N, 43)
    sel = _ext.furthest_point_sampling(xyz.to(DEV), m).long()
    new_xyz = torch.gather(xyz.to(DEV), 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz.to(DEV), r, ns).cpu()
    pick = [0, 11, 20, 31]
    ref = oracle_ext.OracleRowsExt.ball_query(new_xyz[pick].cpu().contiguous(), xyz[pick].contiguous(), r, ns)
    assert torch.equal(idx[pick], ref)
    # the fused query + grouping entry (what the headline's SA1 runs) returns the same neighbourhoods and their rows
    if getattr(_ext, "ball_query_group", None) is not None:
        feats = torch.rand(B, N, 3, device=DEV)
        idx2, rows = _ext.ball_query_group(new_xyz, xyz.to(DEV), feats, r, ns, True, True)
        assert torch.equal(idx2.cpu()[pick], ref)
        want = _ext.group_concat_rows(xyz.to(DEV), new_xyz, feats, idx2, True, True, r)
        assert torch.equal(rows.view_as(want)[pick], want[pick])


# ------------------------------------------------------------------------------------------------ inverse index in one launch
@pytest.mark.parametrize("B,N,m,ns", [(32, 2048, 1024, 32), (32, 1024, 512, 16), (32, 512, 256, 16), (72, 8000, 512, 32),
                                      (9, 4000, 512, 16), (3, 36864, 100, 64), (2, 36865, 100, 64), (1, 1, 5, 3), (5, 7, 1, 1),
                                      (4, 300, 3, 500), (2, 2048, 1024, 64)])
def test_inverse_index_counting_sort_equals_a_stable_sort(B, N, m, ns):
    g = torch.Generator().manual_seed(B * 1000 + N + m + ns)
    idx = torch.randint(0, N, (B, m, ns), generator=g, dtype=torch.int32)
    idx[:, ::2, ns // 2:] = idx[:, ::2, :1]                           # ball-query style padding on half of the neighbourhoods
    if B > 1:
        idx[1] = 0                                                    # empty balls everywhere: ONE point holds every row of a cloud
    idx[0, :, 0] = idx[0, 0, 0]                                       # a heavy point
    ptr, refs = _ext.group_inverse_index(idx.to(DEV), N)
    keys = (idx.long() + torch.arange(B).view(B, 1, 1) * N).flatten()
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(refs.cpu().long(), order)
    want_ptr = torch.zeros(B * N + 1, dtype=torch.long)
    want_ptr[1:] = torch.cumsum(torch.bincount(keys, minlength=B * N), 0)
    assert torch.equal(ptr.cpu().long(), want_ptr)
    p2, r2 = _ext.group_inverse_index(idx.to(DEV), N)                 # deterministic
    assert torch.equal(p2, ptr) and torch.equal(r2, refs)


# ------------------------------------------------------------------------------------------------ interpolation gradient as a gather
@pytest.mark.parametrize("B,n,m,C,ldg,col0", [(32, 1024, 512, 256, 512, 0), (4, 512, 256, 256, 256, 0), (3, 100, 7, 64, 80, 12),
                                              (2, 33, 1, 8, 8, 0), (2, 300, 40, 288, 300, 4)])
def test_three_interpolate_rows_grad_csr_matches_the_atomic_form(B, n, m, C, ldg, col0):
    g = torch.Generator().manual_seed(n + m + C)
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(B, n, 3, generator=g).to(DEV)
    go = torch.randn(B, n, ldg, generator=g).to(DEV)
    inv = _ext.group_inverse_index(idx, m)
    got = _ext.three_interpolate_rows_grad(go, idx, w, m, C, col0=col0, inv=inv)
    want = _ext.three_interpolate_rows_grad(go, idx, w, m, C, col0=col0)
    ref = torch.zeros(B, m, C, dtype=torch.float64, device=DEV)
    src = (go[:, :, col0:col0 + C].double().unsqueeze(2) * w.double().unsqueeze(3)).reshape(B, n * 3, C)
    ref.scatter_add_(1, idx.long().reshape(B, n * 3, 1).expand(-1, -1, C), src)
    torch.testing.assert_close(got.double(), ref, atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(want.double(), ref, atol=1e-4, rtol=1e-5)
    assert torch.equal(_ext.three_interpolate_rows_grad(go, idx, w, m, C, col0=col0, inv=inv), got)      # fixed order


def test_fp_module_with_prefetched_interpolation_matches_the_oracle():
    torch.manual_seed(3)
    fp = pm.PointnetFPModule(mlp=[64 + 16, 32, 32]).to(DEV).train()
    unknown, known = _unit_ball(2, 400, 1).to(DEV), _unit_ball(2, 90, 2).to(DEV)
    uf = torch.randn(2, 16, 400, device=DEV, requires_grad=True)
    kf = torch.randn(2, 64, 90, device=DEV, requires_grad=True)
    interp = fp.interpolation(unknown, known)
    assert len(interp) == 3 and interp[2] is not None
    out = fp(unknown, known, uf, kf, interp=interp)
    out.square().mean().backward()
    g_u, g_k = uf.grad.clone(), kf.grad.clone()

    import copy
    fp_ref = copy.deepcopy(fp).cpu()
    for p_ in fp_ref.parameters():
        p_.grad = None
    uf2, kf2 = uf.detach().cpu().requires_grad_(True), kf.detach().cpu().requires_grad_(True)
    saved = pu._ext
    pu._ext = oracle_ext.OracleRowsExt
    try:
        out_ref = fp_ref(unknown.cpu(), known.cpu(), uf2, kf2)
        out_ref.square().mean().backward()
    finally:
        pu._ext = saved
    torch.testing.assert_close(out.detach().cpu(), out_ref.detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(g_u.cpu(), uf2.grad, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(g_k.cpu(), kf2.grad, atol=1e-4, rtol=1e-3)
    for a, b in zip(fp.parameters(), fp_ref.parameters()):
        torch.testing.assert_close(a.grad.cpu(), b.grad, atol=2e-4, rtol=1e-3)


def test_bf16_first_level_takes_the_rows_grouped_next_to_the_query():
    """bf16 node: a prefetched geometry of a level whose features are data carries the grouped bf16 rows; same result as
    grouping inside the stack (the same kernel on the same operands), and dropped when the arithmetic is fp32."""
    from pointnet2_ops import fused_mlp
    torch.manual_seed(1)
    sa = pm.PointnetSAModule(mlp=[3, 32, 64], npoint=256, radius=0.3, nsample=32).to(DEV).train()
    xyz = _unit_ball(3, 5000, 9).to(DEV)
    feats_rows = torch.rand(3, 5000, 3, device=DEV)
    features = feats_rows.transpose(1, 2)
    prev = fused_mlp.set_mlp_dtype(torch.bfloat16)
    try:
        geo = sa.sample_and_query(xyz, feats_rows=feats_rows)
        assert geo["rows"][0] is not None and geo["rows"][0].dtype == torch.bfloat16
        with torch.no_grad():
            _, a = sa(xyz, features, geometry=geo)
            _, b = sa(xyz, features)
        assert torch.equal(a, b)
        out = sa(xyz, features, geometry=geo)[1]
        out.square().mean().backward()                                   # the backward runs with the prefetched rows
        assert all(torch.isfinite(p_.grad).all() for p_ in sa.parameters())
        fused_mlp.set_mlp_dtype(torch.float32)
        with torch.no_grad():
            _, c = sa(xyz, features, geometry=geo)                       # bf16 rows are of no use to the fp32 node
            _, d = sa(xyz, features)
        torch.testing.assert_close(c, d, atol=1e-5, rtol=1e-5)          # (batch statistics through fp64 atomics: last-bit noise)
    finally:
        fused_mlp.set_mlp_dtype(prev)


# --------------------------------------------------------------- lifted first layer without its output (PRO_LIFT / EPI_MASKL)
def _lift_case(B, N, m, ns, C, N0, normalize, r, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = (_unit_ball(B, N, seed + 1) * 3.0 + 1.5).to(DEV)           # coordinates of a few metres, off-centre
    sel = torch.stack([torch.randperm(N, generator=g)[:m] for _ in range(B)]).to(DEV)
    new_xyz = xyz[torch.arange(B, device=DEV)[:, None], sel].contiguous()
    idx = _ext.ball_query(new_xyz, xyz, r, ns)
    f = torch.randn(B, N, C, generator=g).to(DEV)
    W = (torch.randn(N0, 3 + C, generator=g) * 0.2).to(DEV)
    P = torch.mm(f.view(-1, C), W[:, 3:].t()).view(B, N, N0).contiguous()
    return xyz, new_xyz, idx, f, W, P, g


@pytest.mark.parametrize("B,N,m,ns,C,N0,N1,normalize,r", [(2, 2048, 1024, 32, 128, 128, 128, True, 0.4 * 3),
                                                           (3, 700, 130, 16, 256, 128, 128, True, 0.4 * 3),
                                                           (2, 1000, 77, 64, 32, 64, 64, False, 0.4 * 3),
                                                           (1, 513, 5, 16, 20, 128, 96, True, 1.0),
                                                           (2, 1024, 512, 16, 256, 128, 128, True, 1.2 * 3)])
def test_lifted_layer_without_its_output_matches_the_stored_form(B, N, m, ns, C, N0, N1, normalize, r):
    """pn2_lift_points / pn2_group_lift_stats / pn2_mlp_gemm_lift / pn2_mlp_wgrad_lift / pn2_mlp_dgrad_lift: y0 = Pq[gidx] - Q
    equals W [rel | f[idx]] (float64) to 1e-5; the three consumers equal the library's own kernels run on the MATERIALISED
    y0 = Pq[gidx] - Q — bit for bit where nothing is summed with atomics (the GEMM outputs), to summation order elsewhere.
    Ragged tiles (M = 80), 16 / 32 / 64 rows per centre, a 64-wide lifted layer."""
    e = _ext
    xyz, new_xyz, idx, f, W, P, g = _lift_case(B, N, m, ns, C, N0, normalize, r, B * N + C)
    Wx = W[:, :3].contiguous()
    Pq, Q = e.lift_points(P, xyz, new_xyz, Wx, normalize, r)
    stats = torch.zeros(2, N0, dtype=torch.float64, device=DEV)
    gidx = e.group_lift_stats(Pq, Q, idx, N, stats)
    flat = (idx.long() + (torch.arange(B, device=DEV) * N).view(B, 1, 1)).view(-1)
    assert torch.equal(gidx.long(), flat)
    M = flat.numel()
    y0 = Pq[flat] - Q.repeat_interleave(ns, dim=0)                    # the same fp32 subtraction the kernels make
    rows = e.group_concat_rows(xyz, new_xyz, f, idx, True, normalize, r).view(-1, 3 + C)
    want = rows.double() @ W.double().t()
    assert float((y0.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    torch.testing.assert_close(stats[0], y0.double().sum(0), rtol=1e-6, atol=1e-6 * M)
    torch.testing.assert_close(stats[1], y0.double().square().sum(0), rtol=1e-6, atol=1e-6 * M)
    assert e.mlp_lift_supported(N0, N1, ns)
    # forward of the layer above
    fin0 = _fin(N0, 5)
    W1 = (torch.randn(N1, N0, generator=g) * 0.1).to(DEV)
    st_a = torch.zeros(2, N1, dtype=torch.float64, device=DEV)
    st_b = torch.zeros_like(st_a)
    Ya = e.mlp_gemm_lift(Pq, gidx, Q, ns, fin0, W1, st_a)
    Yb = e.mlp_gemm(y0, W1, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=(fin0[2], fin0[3]), stats=st_b)
    assert_same_product(Ya, Yb)
    torch.testing.assert_close(st_a, st_b, rtol=1e-6, atol=1e-6 * M)
    # weight gradient
    G = torch.randn(M, N1, generator=g).to(DEV)
    consts = (torch.randn(3, N1, generator=g) * 0.5).to(DEV).contiguous()
    dWa = e.mlp_wgrad_lift(Ya, consts, G, Pq, gidx, Q, ns, fin0)
    dWb = e.mlp_wgrad(Ya, consts, y0, e.PRO_GY, e.PRO_BNRELU, G=G, a_fin=fin0)
    torch.testing.assert_close(dWa, dWb, rtol=1e-4, atol=1e-4 * float(dWb.abs().max()))
    a0 = torch.relu(y0.double() * fin0[2].double() + fin0[3].double())
    gy = consts[0].double() * G.double() + consts[1].double() * Ya.double() + consts[2].double()
    dW_want = gy.t() @ a0
    assert float((dWa.double() - dW_want).abs().max()) < 1e-5 * float(dW_want.abs().max()) + 1e-6
    # input gradient + mask + BatchNorm-backward sums of the lifted layer
    Wt = W1.t().contiguous()                                           # (K, N) rows
    su_a = torch.zeros(2, N0, dtype=torch.float64, device=DEV)
    su_b = torch.zeros_like(su_a)
    Ga = e.mlp_dgrad_lift(G, Ya, consts, Wt, su_a, Pq, gidx, Q, ns, fin0)
    Gb = e.mlp_gemm(G, Wt, pro=e.PRO_GY, epi=e.EPI_MASK, X2=Ya, p=(consts[0], consts[1], consts[2]), stats=su_b, Yprev=y0,
                    e_fin=fin0, M=M)
    assert_same_product(Ga, Gb)
    torch.testing.assert_close(su_a, su_b, rtol=1e-6, atol=1e-6 * M)


def test_sa_level_without_the_lifted_layers_output_equals_the_stored_route():
    """One SA level (SA2 of the backbone: 131 -> 128 -> 128 -> 256, 32 rows per centre) with LIFT_FREE on and off: same features,
    feature gradients and parameter gradients up to fp32 rounding; the free route stores no (M, 128) first-layer output."""
    import copy
    from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from pointnet2_ops import fused_mlp
    from test_gpu_round4 import _Calls, _same_up_to_sparse_argmax_flips
    torch.manual_seed(11)
    sa = PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True,
                               normalize_xyz=True).cuda().train()
    xyz = _unit_ball(4, 2048, 51).to(DEV)
    feats = torch.randn(4, 128, 2048, generator=torch.Generator().manual_seed(52)).to(DEV)
    gout = torch.randn(4, 256, 1024, generator=torch.Generator().manual_seed(53)).to(DEV)

    def run(free):
        prev, fused_mlp.LIFT_FREE = fused_mlp.LIFT_FREE, free
        prev_rows, fused_mlp.LIFT_FREE_MIN_ROWS = fused_mlp.LIFT_FREE_MIN_ROWS, 0      # (131 072 rows here: below the default)
        try:
            m = copy.deepcopy(sa)
            f = feats.clone().requires_grad_(True)
            geo = m.sample_and_query(xyz, inverse_index=True)
            with _Calls(_ext, ["group_lift_rows", "group_lift_stats", "mlp_gemm_lift", "mlp_wgrad_lift", "mlp_dgrad_lift"]) as calls:
                _nx, nf, _i = m(xyz, f, geometry=geo)
                (nf * gout).sum().backward()
            want = (0, 1, 1, 1, 1) if free else (1, 0, 0, 0, 0)
            got = tuple(calls.count[k] for k in ("group_lift_rows", "group_lift_stats", "mlp_gemm_lift", "mlp_wgrad_lift", "mlp_dgrad_lift"))
            assert got == want, got
            return nf.detach(), f.grad, {n: p.grad for n, p in m.named_parameters()}
        finally:
            fused_mlp.LIFT_FREE, fused_mlp.LIFT_FREE_MIN_ROWS = prev, prev_rows

    nf_a, gf_a, g_a = run(True)
    nf_b, gf_b, g_b = run(False)
    torch.testing.assert_close(nf_a, nf_b, atol=2e-5, rtol=1e-5)
    _same_up_to_sparse_argmax_flips("d features", gf_a, gf_b)
    for k in g_b:
        assert float((g_a[k] - g_b[k]).norm()) <= 1e-2 * float(g_b[k].norm()) + 1e-6, k


# ------------------------------------------------------------------ bf16: pooled last layer without its output
@pytest.mark.parametrize("M,K,N,ns", [(64 * 300, 64, 128, 64), (32 * 257, 128, 128, 32), (16 * 111, 64, 64, 16), (128 * 9, 64, 128, 128),
                                      (16 * 5, 96, 128, 16)])
def test_bf16_pooled_last_layer_forward_takes_the_maxima_of_its_accumulators(M, K, N, ns):
    """pn2_mlp_gemm_pool_bf16 + pn2_pool_finalize against float64 on the operands the kernel sees (bf16 activations formed
    with the kernel's fma, bf16 weights): group maxima within fp32 accumulation noise, the arg-max row holds the maximum,
    among bit-identical rows (duplicated on purpose) the FIRST one wins; column sums; negative gammas (flipped weight rows),
    ragged last tiles, 16 .. 128 rows per group."""
    e = _ext
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=g).to(BF)
    Xg = X.view(M // ns, ns, K)
    Xg[:, 7] = Xg[:, 3]                       # rows 3, 7 and 11 of every group are the same row: ties
    Xg[:, 11] = Xg[:, 3]
    X = X.to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    finp = _fin(K, 7)
    gamma = (torch.rand(N, generator=g) - 0.3).to(DEV)
    assert e.pool_layer_bf16_supported(K, N, ns) == (K in (32, 64, 128))
    Wf, sgn = e.pool_flip_rows(W.contiguous(), gamma)
    stats = torch.zeros(2, N, dtype=torch.float64, device=DEV)
    pmax, parg = e.mlp_gemm_pool_bf16(X, Wf, sgn, ns, (finp[2], finp[3]), stats)
    # fma(x, scale, shift) in fp32 = the float64 expression rounded once (x is bf16: the product is exact in float64)
    pre = (X.double() * finp[2].double() + finp[3].double()).float()
    A = torch.relu(pre).to(BF).double()
    Y = A @ Wf.to(BF).double().t()                                           # flipped accumulators, float64
    psz = min(ns, 32)
    Yg = Y.view(M // psz, psz, N)
    want_max = Yg.max(dim=1)[0]
    scale = float(Y.abs().max())
    assert float((pmax.double() - want_max).abs().max()) <= 2e-6 * scale
    at_arg = Yg.gather(1, parg.long().unsqueeze(1)).squeeze(1)
    assert float((want_max - at_arg).max()) <= 4e-6 * scale
    sub = torch.arange(M // psz, device=DEV) % max(ns // psz, 1)              # partial group inside its group
    dup = (parg == 7) | (parg == 11)
    assert not bool((dup & (sub == 0).unsqueeze(1)).any()), "a later copy of a bit-identical row won the maximum"
    Yt = Y * sgn.double()
    torch.testing.assert_close(stats[0], Yt.sum(0), rtol=1e-5, atol=1e-5 * scale * M ** 0.5)
    torch.testing.assert_close(stats[1], Yt.square().sum(0), rtol=1e-5, atol=1e-5 * scale * scale * M ** 0.5)
    # pooled activations as bn_relu_rows_max would give them on the un-flipped values
    fin = e.bn_finalize(stats, M, gamma, torch.zeros_like(gamma), 1e-5, 0.0, None, None, None)
    out, arg, yraw = e.pool_finalize(pmax, parg, fin, sgn, ns)
    z = torch.relu(Yt * fin[2].double() + fin[3].double()).view(M // ns, ns, N).max(dim=1)[0]
    torch.testing.assert_close(out.double(), z, rtol=1e-4, atol=1e-4)
    assert not bool(((arg == 7) | (arg == 11)).any())


@pytest.mark.parametrize("M,N,K,ns", [(64 * 300, 128, 64, 64), (32 * 257, 128, 128, 32), (16 * 111, 64, 64, 16), (16 * 5, 64, 32, 16),
                                      (128 * 9, 128, 64, 128)])
def test_bf16_pooled_last_layer_backward_reforms_its_output(M, N, K, ns):
    """pn2_mlp_bwd_bf16_pool (y_L re-formed from y_{L-1} inside the kernel) against float64 formulas on the same bf16 operands,
    and against pn2_mlp_bwd_bf16 fed the STORED bf16 y_L: at least as close to the float64 result."""
    e = _ext
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    yprev = torch.randn(M, K, generator=g).to(BF).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    Wt = W.t().contiguous()
    c = torch.stack([torch.randn(N, generator=g), torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.05]).to(DEV)
    fin = torch.stack([torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5, torch.rand(K, generator=g) + 0.5,
                       torch.randn(K, generator=g) * 0.3]).to(DEV).contiguous()
    R = M // ns
    arg = torch.randint(0, ns, (R, N), generator=g, dtype=torch.int32).to(DEV)
    gP = torch.randn(R, N, generator=g).to(DEV)
    assert e.pool_layer_bf16_supported(K, N, ns)
    Gout, sums, dW = e.mlp_bwd_bf16_pool(c, Wt, yprev, fin, arg, gP, ns)
    # float64 on the operands the kernel sees: bf16 activations and weights, fp32 accumulation ~ float64 here
    act = torch.relu(yprev.float() * fin[2] + fin[3]).to(BF).double()
    Wb = W.to(BF).double()
    Y = act @ Wb.t()
    dense = torch.zeros(R, ns, N, dtype=torch.float64, device=DEV)
    dense.scatter_(1, arg.long().unsqueeze(1), gP.double().unsqueeze(1))
    gy = c[0].double() * dense.view(M, N) + c[1].double() * Y + c[2].double()
    mask = (yprev.float() * fin[2] + fin[3]) > 0
    G64 = (gy @ Wb) * mask
    dW64 = gy.t() @ act

    def err(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    e_g, e_w = err(Gout, G64), err(dW, dW64)
    assert e_g <= 1.0 / 64 and e_w <= 4e-3, (e_g, e_w)
    o = Gout.float().double()
    torch.testing.assert_close(sums[0], o.sum(0), rtol=1e-4, atol=1e-5 * M)
    torch.testing.assert_close(sums[1], (o * ((yprev.float() - fin[0]) * fin[1]).double()).sum(0), rtol=1e-4, atol=1e-5 * M)
    # the stored route on the rounded y_L
    G2, _s2, dW2 = e.mlp_bwd_bf16(Y.float().to(BF), c, Wt, yprev, fin, e.PRO_POOLG, arg=arg, gP=gP, ns=ns)
    assert e_g <= 1.5 * err(G2, G64) + 1e-3 and e_w <= 1.5 * err(dW2, dW64) + 1e-4, (e_g, err(G2, G64), e_w, err(dW2, dW64))


def test_bf16_stack_with_the_pooled_layer_not_stored_matches_the_stored_route():
    """The backbone's SA1 stack ([3+3, 64, 64, 128], 64 rows per centre) on the bf16 node with BF16_POOL on and off: pooled
    features within bf16 rounding of each other (the un-stored route is the one that rounds LESS), gradients in norm; the route
    taken is checked by the entry points called."""
    import copy
    from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from pointnet2_ops import fused_mlp
    from test_gpu_round4 import _Calls
    torch.manual_seed(13)
    sa = PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).cuda().train()
    xyz = _unit_ball(3, 20000, 61).to(DEV)
    feats = torch.randn(3, 3, 20000, generator=torch.Generator().manual_seed(62)).to(DEV)
    gout = torch.randn(3, 128, 512, generator=torch.Generator().manual_seed(63)).to(DEV)

    def run(dtype, pool):
        prev_d = fused_mlp.set_mlp_dtype(dtype)
        prev, fused_mlp.BF16_POOL = fused_mlp.BF16_POOL, pool
        try:
            m = copy.deepcopy(sa)
            with _Calls(_ext, ["mlp_gemm_pool_bf16", "mlp_bwd_bf16_pool", "bn_relu_rows_max_bf16"]) as calls:
                _nx, nf, _i = m(xyz, feats)
                (nf * gout).sum().backward()
            if dtype == torch.bfloat16:
                want = (1, 1, 0) if pool else (0, 0, 1)
                assert tuple(calls.count[k] for k in ("mlp_gemm_pool_bf16", "mlp_bwd_bf16_pool", "bn_relu_rows_max_bf16")) == want, calls.count
            return nf.detach(), {n: p.grad for n, p in m.named_parameters()}
        finally:
            fused_mlp.BF16_POOL = prev
            fused_mlp.set_mlp_dtype(prev_d)

    ref, gref = run(torch.float32, False)
    a, ga = run(torch.bfloat16, True)
    b, gb = run(torch.bfloat16, False)
    ea = float((a - ref).abs().max() / ref.abs().max())
    eb = float((b - ref).abs().max() / ref.abs().max())
    print(f"\n[bf16 pooled layer] forward rel-max vs fp32: not stored {ea:.3e}, stored {eb:.3e}", end="")
    assert ea <= 2e-2 and ea <= 1.5 * eb + 1e-3
    for k in gref:
        na = float((ga[k] - gref[k]).norm() / (gref[k].norm() + 1e-12))
        nb = float((gb[k] - gref[k]).norm() / (gref[k].norm() + 1e-12))
        print(f"\n[bf16 pooled layer] d{k}: rel-L2 vs fp32 not stored {na:.3e}, stored {nb:.3e}", end="")
        assert na <= max(1.5 * nb, 5e-2), k


# ------------------------------------------------------------------ bf16: first layer without its output
@pytest.mark.parametrize("M,K0,K,N", [(64 * 333, 6, 64, 64), (5000, 3, 64, 128), (777, 8, 32, 32), (128 * 20 + 5, 4, 128, 64)])
def test_bf16_second_layer_reforms_the_first_from_its_input_rows(M, K0, K, N):
    """pn2_mlp_gemm_first_bf16 against float64 on the operands the kernel sees, and against the stored route (first layer's
    GEMM -> bf16 y_0 -> second layer): the re-formed route rounds once less and must be at least as close."""
    e = _ext
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(M + K0 + K + N)
    X = torch.zeros(M, 8)
    X[:, :K0] = torch.randn(M, K0, generator=g)
    X = X.to(BF).to(DEV)
    W0 = (torch.randn(K, K0, generator=g) * 0.5).to(DEV)
    W1 = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    fin0 = _fin(K, 9)
    assert e.mlp_gemm_first_bf16_supported(K0, K, N)
    st = torch.zeros(2, N, dtype=torch.float64, device=DEV)
    Y = e.mlp_gemm_first_bf16(X, K0, W0, fin0, W1, st)
    y0 = X[:, :K0].double() @ W0.to(BF).double().t()          # (bf16 operands on the matrix pipe, like the first layer's own GEMM)
    a = torch.relu(y0 * fin0[2].double() + fin0[3].double()).float().to(BF).double()
    want = a @ W1.to(BF).double().t()
    scale = float(want.abs().max())
    err = float((Y.double() - want).abs().max())
    assert err <= scale / 128, (err, scale)
    torch.testing.assert_close(st[0], Y.double().sum(0), rtol=1e-5, atol=1e-4 * M ** 0.5)
    torch.testing.assert_close(st[1], Y.double().square().sum(0), rtol=1e-5, atol=1e-4 * M ** 0.5)
    # stored route
    y0s = e.mlp_gemm_bf16(X, W0, pro=e.PRO_NONE, epi=e.EPI_NONE)
    Y2 = e.mlp_gemm_bf16(y0s, W1, pro=e.PRO_BNRELU, epi=e.EPI_NONE, p=(fin0[2], fin0[3]))
    err2 = float((Y2.double() - want).abs().max())
    assert err <= 1.5 * err2 + scale / 512, (err, err2)


def test_bf16_stack_without_first_and_last_layer_outputs_matches_fp32():
    """The backbone's SA1 stack on the bf16 node with BF16_FIRST / BF16_POOL on and off against the fp32 node: the routes are
    checked by the entry points called; the un-stored routes are at least as close to fp32 as the stored ones."""
    import copy
    from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from pointnet2_ops import fused_mlp
    from test_gpu_round4 import _Calls
    torch.manual_seed(17)
    sa = PointnetSAModuleVotes(npoint=512, radius=0.3, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).cuda().train()
    xyz = _unit_ball(3, 20000, 71).to(DEV)
    feats = torch.randn(3, 3, 20000, generator=torch.Generator().manual_seed(72)).to(DEV)
    gout = torch.randn(3, 128, 512, generator=torch.Generator().manual_seed(73)).to(DEV)
    names = ["mlp_gemm_first_bf16", "mlp_bwd_bf16_fold_first", "mlp_bwd_bf16_fold", "mlp_gemm_pool_bf16", "mlp_bwd_bf16_pool"]

    def run(dtype, free):
        prev_d = fused_mlp.set_mlp_dtype(dtype)
        prev = (fused_mlp.BF16_FIRST, fused_mlp.BF16_POOL)
        fused_mlp.BF16_FIRST = fused_mlp.BF16_POOL = free
        try:
            m = copy.deepcopy(sa)
            with _Calls(_ext, names) as calls:
                _nx, nf, _i = m(xyz, feats)
                (nf * gout).sum().backward()
            if dtype == torch.bfloat16:
                want = (1, 1, 0, 1, 1) if free else (0, 0, 1, 0, 0)
                assert tuple(calls.count[k] for k in names) == want, calls.count
            return nf.detach(), {n: p.grad for n, p in m.named_parameters()}, {n: b.clone() for n, b in m.named_buffers()}
        finally:
            fused_mlp.BF16_FIRST, fused_mlp.BF16_POOL = prev
            fused_mlp.set_mlp_dtype(prev_d)

    ref, gref, bref = run(torch.float32, False)
    a, ga, ba = run(torch.bfloat16, True)
    b, gb, _bb = run(torch.bfloat16, False)
    ea = float((a - ref).abs().max() / ref.abs().max())
    eb = float((b - ref).abs().max() / ref.abs().max())
    print(f"\n[bf16 SA1, nothing stored at its ends] forward rel-max vs fp32: {ea:.3e} (stored {eb:.3e})", end="")
    assert ea <= 2e-2 and ea <= 1.5 * eb + 1e-3
    for k in gref:
        na = float((ga[k] - gref[k]).norm() / (gref[k].norm() + 1e-12))
        nb = float((gb[k] - gref[k]).norm() / (gref[k].norm() + 1e-12))
        print(f"\n[bf16 SA1] d{k}: rel-L2 vs fp32 {na:.3e} (stored {nb:.3e})", end="")
        assert na <= max(1.5 * nb, 5e-2), k
    for k in bref:
        if "running" in k:
            assert float((ba[k] - bref[k]).abs().max()) <= 2e-2 * float(bref[k].abs().max()) + 1e-3, k
