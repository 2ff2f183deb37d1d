"""GPU parity cases added in round 2 (VERDICT r01 "next" item 2 and the ops added since):

* BASELINE configs[0] (one 20k-point cloud through the MSG object encoder) on the HIP path against the fixture the
  imported reference python layer produced;
* the SA/FP backbone at the north-star tolerance: eval mode end to end at 1e-4, train mode LEVEL BY LEVEL with every
  level fed the oracle's inputs at 1e-4 (the achieved maxima are printed);
* avg / rbf pooling, the classification heads and sample_uniformly against the reference fixtures / the oracle;
* the stress shape of configs[4]: 64 x 200k points through the full stack (finite, FPS bit-exact on two clouds) and a
  3-layer TripletGCN over 64 block-diagonal scenes against the oracle backend.
"""
import copy
import os

import numpy as np
import pytest
import torch

import oracle_ext
from pointnet2_ops import pointnet2_modules as pm
from pointnet2_ops import pointnet2_utils as pu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_config0_20k_point_cloud_through_the_hip_encoder():
    """BASELINE.json configs[0] on the product path (the CPU twin is tests/test_golden.py::test_config1_plumbing_case)."""
    from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet2 import PointNetfeat
    z = load("msg_encoder.npz")
    torch.manual_seed(23)
    enc = PointNetfeat(input_dim=6, out_size=256, input_dropout=0.0).eval().cuda()
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1, 20000, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(1, 20000, 1, generator=g).pow(1 / 3)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).max()
    pc = torch.cat([p, torch.rand(1, 20000, 3, generator=g)], dim=2)
    assert abs(float(pc.double().sum()) - float(z["cfg1/pc_checksum"][0])) < 1e-9
    with torch.no_grad():
        y = enc(pc.transpose(1, 2).contiguous().cuda())
    err = float(np.abs(y.cpu().numpy() - z["cfg1/y"]).max())
    print(f"\n[configs[0]] max |hip - reference layer on oracle| = {err:.3e}")
    np.testing.assert_allclose(y.cpu().numpy(), z["cfg1/y"], atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ backbone tolerances
def _backbone_pair(seed, B, N):
    from external_src.group_free_3D.models.backbone_module import Pointnet2Backbone
    torch.manual_seed(seed)
    net = Pointnet2Backbone(input_feature_dim=3)
    g = torch.Generator().manual_seed(seed + 1)
    pc = torch.rand(B, N, 6, generator=g) * 2 - 1
    return net, pc


def _with_backend(backend, fn):
    saved = pu._ext
    pu._ext = backend
    try:
        return fn()
    finally:
        pu._ext = saved


def test_backbone_eval_mode_end_to_end_within_1e4():
    """Six levels, eval mode (running statistics): HIP path vs the oracle backend at the north-star 1e-4."""
    from pointnet2_ops import _ext
    net, pc = _backbone_pair(71, 2, 3000)
    # non-trivial running statistics: a few train-mode steps on the oracle backend
    net.train()
    _with_backend(oracle_ext.OracleRowsExt, lambda: [net(pc) for _ in range(2)])
    net.eval()
    with torch.no_grad():
        ref = _with_backend(oracle_ext.OracleRowsExt, lambda: net(pc))
        got = _with_backend(_ext, lambda: copy.deepcopy(net).cuda()(pc.cuda()))
    worst = 0.0
    for k in ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features"):
        a, b = got[k].cpu(), ref[k]
        e = float((a - b).abs().max())
        worst = max(worst, e)
        print(f"\n[eval backbone] {k}: max abs err {e:.3e} (max |ref| {float(b.abs().max()):.3f})", end="")
        torch.testing.assert_close(a, b, atol=1e-4, rtol=1e-4)
    for k in ("sa1_inds", "sa2_inds", "sa4_xyz"):
        assert torch.equal(got[k].cpu(), ref[k])
    print(f"\n[eval backbone] worst level error {worst:.3e}")


def test_backbone_train_mode_level_by_level_within_1e4():
    """Train mode (batch statistics), every level fed the ORACLE's inputs: the per-level error of the HIP path stays
    within 1e-4, so what the end-to-end train-mode comparison (tests/test_gpu_model.py, 1e-3) adds on top is the
    re-normalisation of upstream rounding noise by the following BatchNorms, not a kernel error."""
    from pointnet2_ops import _ext
    net, pc = _backbone_pair(73, 2, 3000)
    net.train()
    xyz, feats = net._break_up_pc(pc)
    levels = []                                          # (module name, inputs, reference output)

    def ref_pass():
        x, f = xyz, feats
        ep = {}
        for i in (1, 2, 3, 4):
            mod = getattr(net, f"sa{i}")
            nx, nf, inds = mod(x, f)
            levels.append((f"sa{i}", (x, f), nf.detach().contiguous(), nx))
            ep[i] = (nx, nf.detach().contiguous())
            x, f = nx, nf.detach().contiguous()
        f1 = net.fp1(ep[3][0], ep[4][0], ep[3][1], ep[4][1]).detach().contiguous()
        levels.append(("fp1", (ep[3][0], ep[4][0], ep[3][1], ep[4][1]), f1, None))
        f2 = net.fp2(ep[2][0], ep[3][0], ep[2][1], f1).detach().contiguous()
        levels.append(("fp2", (ep[2][0], ep[3][0], ep[2][1], f1), f2, None))

    with torch.no_grad():
        _with_backend(oracle_ext.OracleRowsExt, ref_pass)
        gpu = copy.deepcopy(net).cuda().train()
        worst = 0.0
        for name, inputs, want, want_xyz in levels:
            mod = getattr(gpu, name)
            out = _with_backend(_ext, lambda: mod(*[t.cuda() for t in inputs]))
            got = (out[1] if isinstance(out, tuple) else out).cpu()
            if want_xyz is not None:
                assert torch.equal(out[0].cpu(), want_xyz), name           # FPS / gather bit-exact
            e = float((got - want).abs().max())
            worst = max(worst, e)
            print(f"\n[train level] {name}: max abs err {e:.3e} (max |ref| {float(want.abs().max()):.3f})", end="")
            torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-4)
    print(f"\n[train level] worst level error {worst:.3e}")


# ------------------------------------------------------------------------------------------- reference-fixture checks
@pytest.mark.parametrize("pooling,norm,sigma", [("max", True, None), ("avg", False, None), ("rbf", True, None), ("rbf", False, 0.11)])
def test_votes_pooling_modes_on_gpu(pooling, norm, sigma):
    from test_golden import _votes_case
    z = load("votes_pooling.npz")
    tag, nx, nf, inds, gf, gw = _votes_case(z, pooling, norm, sigma, device="cuda")
    assert np.array_equal(inds.cpu().numpy(), z[f"{tag}/inds"])
    assert np.array_equal(nx.detach().cpu().numpy(), z[f"{tag}/new_xyz"])
    np.testing.assert_allclose(nf.detach().cpu().numpy(), z[f"{tag}/new_features"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(gf.cpu().numpy(), z[f"{tag}/grad_features"], atol=1e-4, rtol=1e-3)
    want = z[f"{tag}/grad_w0"]
    assert float(np.abs(gw.cpu().numpy() - want).max()) <= 1e-3 * float(np.abs(want).max()) + 1e-5


def test_heads_on_gpu():
    from test_golden import test_heads_match_reference
    test_heads_match_reference(device="cuda")


@pytest.mark.parametrize("B,N,m,ns,r", [(2, 900, 40, 16, 0.3), (3, 5000, 333, 64, 0.12), (2, 400, 50, 100, 0.5), (1, 300, 7, 5, 0.01)])
def test_sample_uniformly_kernel_matches_oracle(B, N, m, ns, r):
    """pn2_ball_query_unique_resample vs the CPU restatement (torch.unique + the same counter-based draws): indices and
    unique counts bit-exact, including rows longer than a wave, empty balls and full rows."""
    from pointnet2_ops import _ext
    g = torch.Generator().manual_seed(B * 1000 + m)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    new_xyz = xyz[:, :m].contiguous()
    new_xyz[:, -1] += 10.0                                               # an empty ball
    idx_ref = oracle_ext.OracleRowsExt.ball_query(new_xyz, xyz, r, ns)
    idx = _ext.ball_query(new_xyz.cuda(), xyz.cuda(), r, ns)
    assert torch.equal(idx.cpu(), idx_ref)
    cnt_ref = oracle_ext.OracleRowsExt.ball_query_unique_resample(idx_ref, 12345)
    cnt = _ext.ball_query_unique_resample(idx, 12345)
    assert torch.equal(cnt.cpu(), cnt_ref)
    assert torch.equal(idx.cpu(), idx_ref)


def test_query_and_group_sample_uniformly_on_gpu():
    z = load("sample_uniformly.npz")
    pc = torch.from_numpy(z["pc"]).cuda()
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    qg = pu.QueryAndGroup(0.3, 16, use_xyz=True, ret_grouped_xyz=True, sample_uniformly=True, ret_unique_cnt=True)
    torch.manual_seed(62)
    grouped, grouped_xyz, cnt = qg(xyz, xyz[:, :40].contiguous(), feats)
    assert np.array_equal(cnt.cpu().numpy(), z["unique_cnt"])
    got, want = grouped.cpu().numpy(), z["grouped"]
    for b in range(cnt.size(0)):
        for r in range(cnt.size(1)):
            n = int(cnt[b, r])
            assert np.array_equal(got[b, :, r, :n], want[b, :, r, :n])


# ------------------------------------------------------------------------------------------------- configs[4] stress
def test_stress_shape_full_stack_64x200k():
    """BASELINE configs[4]: 64 clouds x 200k points through the whole SA/FP stack, forward + backward: finite
    everywhere, and the first-level FPS indices of two clouds equal the oracle's."""
    from external_src.group_free_3D.models.backbone_module import Pointnet2Backbone
    torch.manual_seed(0)
    net = Pointnet2Backbone(input_feature_dim=3).cuda().train()
    g = torch.Generator().manual_seed(11)
    p = torch.randn(64, 200000, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(64, 200000, 1, generator=g).pow(1 / 3)
    pc = torch.cat([p, torch.rand(64, 200000, 3, generator=g)], dim=2)
    ep = net(pc.cuda())
    out = ep["fp2_features"]
    assert out.shape == (64, 288, 1024)
    out.square().mean().backward()
    assert bool(torch.isfinite(out).all())
    assert all(torch.isfinite(q.grad).all() for q in net.parameters())
    for b in (0, 63):
        want = oracle_ext.OracleRowsExt.furthest_point_sampling(p[b:b + 1].contiguous(), 96)
        assert torch.equal(ep["sa1_inds"][b:b + 1, :96].cpu(), want)


@pytest.mark.parametrize("E,N,H,ldp,cola,colb", [(72, 9, 512, 1024, 0, 512), (4608, 576, 512, 1024, 0, 512), (7, 3, 4, 16, 8, 4),
                                                 (300, 40, 1280, 2560, 1280, 0)])
def test_lifted_product_gather_and_two_window_segment_sum_are_bit_exact(E, N, H, ldp, cola, colb):
    from pointnet2_ops import _ext
    g = torch.Generator().manual_seed(E + H)
    p = torch.randn(N, ldp, generator=g)
    q = torch.randn(E, H, generator=g)
    ia, ib = torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)
    want = oracle_ext.OracleRowsExt.gather2_add_rows(q.clone(), p, ia, ib, cola, colb)
    got = _ext.gather2_add_rows(q.cuda(), p.cuda(), ia.cuda(), ib.cuda(), cola, colb)
    assert torch.equal(got.cpu(), want)
    src = torch.randn(E, ldp + 8, generator=g)
    order = torch.sort(ia, stable=True).indices
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(ia, minlength=N), 0)
    want = oracle_ext.OracleRowsExt.segment_sum2_rows(src, order, rowptr, N, H, cola, colb + 8)
    got = _ext.segment_sum2_rows(src.cuda(), order.cuda(), rowptr.cuda(), N, H, cola, colb + 8)
    assert torch.equal(got.cpu(), want)
    with pytest.raises(RuntimeError):
        _ext.gather2_add_rows(q.cuda(), p.cuda(), ia.cuda(), ib.cuda(), cola, ldp)      # window past the row


@pytest.mark.parametrize("route", ["fused", "lifted", "concat"])
def test_three_layer_gcn_on_64_block_diagonal_scenes_matches_oracle(route, monkeypatch):
    """configs[4]'s "3-hop GNN" over 64 scenes batched block-diagonally (per-scene BatchNorm statistics,
    network_TripletGCN.py:20): HIP kernels vs the oracle backend, forward and gradients — the fused per-scan layer kernels
    (csrc/gcn_fused.hip, the default for scans of <= 128 rows), and the unfused path with the first Linear of the triplet
    MLP lifted after the product resp. in the literal concat form."""
    from pointnet2_ops import _ext
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    lifted = route == "lifted"
    monkeypatch.setattr(gcn, "FUSED_LAYER", route == "fused")
    assert gcn.FUSED_MAX_SCANS >= 64                       # configs[4]'s 64 scenes stay on the fused kernels by default (round 6)
    monkeypatch.setattr(gcn, "LIFT_MIN_EDGES", 0 if lifted else 1 << 60)
    torch.manual_seed(5)
    model = gcn.TripletGCNModel(num_layers=3, dim_node=256, dim_edge=256, dim_hidden=512).train()
    g = torch.Generator().manual_seed(6)
    n_objs = [int(v) for v in torch.randint(4, 12, (64,), generator=g)]
    nodes, edges, node_ptr, edge_ptr = [], [], [0], [0]
    for n in n_objs:
        ei = torch.tensor([[a, b] for a in range(n) for b in range(n) if a != b]).t() + node_ptr[-1]
        edges.append(ei)
        node_ptr.append(node_ptr[-1] + n)
        edge_ptr.append(edge_ptr[-1] + ei.size(1))
    ei = torch.cat(edges, dim=1).contiguous()
    x = torch.randn(node_ptr[-1], 256, generator=g)
    e = torch.randn(edge_ptr[-1], 256, generator=g)
    scenes = gcn.SceneBatch(torch.tensor(node_ptr), torch.tensor(edge_ptr))

    def run(dev, backend):
        saved = gcn._ext
        gcn._ext = backend
        try:
            m = copy.deepcopy(model).to(dev)
            xx, ee = x.detach().clone().to(dev).requires_grad_(True), e.detach().clone().to(dev).requires_grad_(True)
            ox, oe = m(xx, ee, ei.to(dev), scenes=scenes.to(dev))
            (ox.square().mean() + oe.square().mean()).backward()
            return ox.detach().cpu(), oe.detach().cpu(), xx.grad.cpu(), ee.grad.cpu(), [q.grad.cpu() for q in m.parameters()]
        finally:
            gcn._ext = saved

    ref = run("cpu", oracle_ext.OracleRowsExt)
    calls = {"n": 0}
    if route == "fused":
        real = _ext.gcn_layer_forward
        monkeypatch.setattr(_ext, "gcn_layer_forward", lambda *a_, **k_: (calls.__setitem__("n", calls["n"] + 1), real(*a_, **k_))[1])
    got = run("cuda", _ext)
    assert calls["n"] == (3 if route == "fused" else 0)        # the 3 layers went through the fused kernels (one C call each)
    for a, b, name in zip(got[:4], ref[:4], ("nodes", "edges", "grad nodes", "grad edges")):
        err = float((a - b).abs().max())
        print(f"\n[gcn x64 {route}] {name}: max abs err {err:.3e} (max |ref| {float(b.abs().max()):.3f})", end="")
        assert err <= 2e-4 * max(1.0, float(b.abs().max())), name
    # parameter gradients: BatchNorm over the 4..11 node rows / 12..110 edge rows of ONE scan is ill-conditioned — the same
    # model in fp32 vs fp64 on the CPU (pure torch) already differs by 4.5e-4 in norm per parameter (tools/gcn_conditioning.py);
    # two fp32 implementations are compared at 1e-2.  Linear biases in front of a BatchNorm have an exactly-zero gradient
    # (1e-18 noise on both sides) and are skipped.
    top = max(float(b.norm()) for b in ref[4])
    worst = 0.0
    for a, b in zip(got[4], ref[4]):
        if float(b.norm()) > 1e-6 * top:
            worst = max(worst, float((a - b).norm() / b.norm()))
    print(f"\n[gcn x64] worst parameter-gradient rel-L2 {worst:.3e}", end="")
    assert worst <= 1e-2


# ------------------------------------------------------------------------------------------------- batched scans
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("C,col0,ldx", [(512, 0, 512), (1280, 0, 1280), (100, 28, 128), (3, 1, 7)])
def test_segment_bn_kernel_matches_per_scan_torch_batch_norm(relu, C, col0, ldx):
    from pointnet2_ops import _ext
    g = torch.Generator().manual_seed(C + col0)
    sizes = [12, 72, 2, 110, 30, 1 + 1, 56]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    R = int(ptr[-1])
    x = torch.randn(R, ldx, generator=g) * 3 + 1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    go = torch.randn(R, C, generator=g)
    y_ref, mean_ref, rstd_ref = oracle_ext.OracleRowsExt.segment_bn_rows(x, ptr, gamma, beta, 1e-5, relu, h=C, col0=col0)
    gx_ref, dg_ref, db_ref = oracle_ext.OracleRowsExt.segment_bn_rows_grad(go, x, ptr, gamma, beta, mean_ref, rstd_ref, relu,
                                                                           col0=col0, eps=1e-5)
    d = "cuda"
    y, mean, rstd = _ext.segment_bn_rows(x.to(d), ptr.to(d), gamma.to(d), beta.to(d), 1e-5, relu, h=C, col0=col0)
    gx, dg, db = _ext.segment_bn_rows_grad(go.to(d), x.to(d), ptr.to(d), gamma.to(d), beta.to(d), mean, rstd, relu, col0=col0)
    torch.testing.assert_close(y.cpu(), y_ref, atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(mean.cpu(), mean_ref, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(gx.cpu(), gx_ref, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(dg.cpu(), dg_ref, atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(db.cpu(), db_ref, atol=1e-3, rtol=1e-4)


def test_batched_scans_on_gpu_equal_single_scan_steps():
    """a block-diagonal batch of scans through the HIP path == the single-scan results (forward
    <= 1e-4; loss; gradients of the mean per-scan loss), eval-mode encoders, per-scan GCN BatchNorm."""
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    torch.manual_seed(0)
    m = SGPNModelWrapper(config_loader("no_gt.json"), 12, 15, torch.rand(12) + 0.5, torch.rand(15) + 0.5,
                         RELATION_NAMES).cuda().eval()
    scans = [synthetic_scan(n, 1024, 2048, seed=i, scan_id=f"s{i}") for i, n in enumerate([5, 9, 4, 7, 6, 9, 8, 5])]
    batch = to_device(collate_scans(scans), "cuda")
    obj, rel = m(batch)
    loss = m.loss(obj, rel, batch)
    loss.backward()
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    outs, total = [], 0.0
    for s in scans:
        sd = to_device(s, "cuda")
        o, r = m(sd)
        l = m.loss(o, r, sd) / len(scans)
        l.backward()
        outs.append((o.detach(), r.detach()))
        total += float(l.detach())
    e_obj = float((obj.detach() - torch.cat([o for o, _ in outs])).abs().max())
    e_rel = float((rel.detach() - torch.cat([r for _, r in outs])).abs().max())
    print(f"\n[batched scans] log-prob max abs err: objects {e_obj:.3e}, relations {e_rel:.3e}; loss {float(loss):.6f} vs {total:.6f}")
    assert e_obj <= 1e-4 and e_rel <= 1e-4
    assert abs(float(loss.detach()) - total) < 1e-5
    # gradients: both sides are the same fp32 kernels with different summation orders (rocBLAS picks other GEMM splits
    # for 56 rows than for 5, atomics reorder); the per-scan BatchNorm over 4..9 node rows amplifies that (see the
    # conditioning note in the GCN test above), so parameters are compared in norm
    top = max(float(p.grad.norm()) for p in m.parameters() if p.grad is not None)
    errs = sorted(((float((got[n] - p.grad).norm() / p.grad.norm()), n, float(p.grad.norm())) for n, p in m.named_parameters()
                   if p.grad is not None and float(p.grad.norm()) > 1e-4 * top), reverse=True)
    for e_, n, nrm in errs[:4]:
        print(f"[batched scans] grad rel-L2 {e_:.3e}  |g| {nrm:.3e}  {n}")
    assert errs[0][0] <= 6e-2
    assert m.predict_step(batch) == [m.predict_step(to_device(s, "cuda")) for s in scans]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_batched_training_on_gpu_with_per_scan_statistics_equals_single_scan_steps(dtype):
    """TRAINING mode on the HIP path: S scans per step with `per_scan_statistics` (default) == S single-scan steps of the
    reference's loop (main.py:54-56): log-probabilities, mean loss, averaged gradients, running statistics after the S
    momentum updates.  Dropout off (its random stream differs between one call and S calls).  bf16: the same comparison
    with the mixed-precision stacks on BOTH sides (identical per-scan arithmetic, so the bounds stay the fp32 ones)."""
    from pointnet2_ops import fused_mlp
    prev = fused_mlp.set_mlp_dtype(dtype)
    try:
        _batched_training_equals_single_scan_steps()
    finally:
        fused_mlp.set_mlp_dtype(prev)


def _batched_training_equals_single_scan_steps():
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    torch.manual_seed(0)
    m = SGPNModelWrapper(config_loader("no_gt.json"), 12, 15, torch.rand(12) + 0.5, torch.rand(15) + 0.5,
                         RELATION_NAMES).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    single = copy.deepcopy(m)
    scans = [synthetic_scan(n, 1024, 2048, seed=i, scan_id=f"s{i}") for i, n in enumerate([5, 9, 4, 7, 6, 9, 8, 5])]
    batch = to_device(collate_scans(scans), "cuda")
    obj, rel = m(batch)
    loss = m.loss(obj, rel, batch)
    loss.backward()
    outs, total = [], 0.0
    for s in scans:
        sd = to_device(s, "cuda")
        o, r = single(sd)
        l = single.loss(o, r, sd) / len(scans)
        l.backward()
        outs.append((o.detach(), r.detach()))
        total += float(l.detach())
    e_obj = float((obj.detach() - torch.cat([o for o, _ in outs])).abs().max())
    e_rel = float((rel.detach() - torch.cat([r for _, r in outs])).abs().max())
    print(f"\n[batched training] log-prob max abs err: objects {e_obj:.3e}, relations {e_rel:.3e}; loss {float(loss):.6f} vs {total:.6f}")
    assert e_obj <= 2e-4 and e_rel <= 2e-4            # train-mode BatchNorm over 4-9 rows on top of the eval-mode 1e-4
    assert abs(float(loss.detach()) - total) < 1e-5
    want = dict(single.named_parameters())
    top = max(float(p.grad.norm()) for p in single.parameters() if p.grad is not None)
    errs = sorted(((float((p.grad - want[n].grad).norm() / want[n].grad.norm()), n) for n, p in m.named_parameters()
                   if p.grad is not None and float(want[n].grad.norm()) > 1e-4 * top), reverse=True)
    print(f"[batched training] worst gradient rel-L2 {errs[0][0]:.3e} ({errs[0][1]})")
    assert errs[0][0] <= 6e-2                         # same conditioning-aware bound as the eval-mode test above
    stats = dict(single.named_buffers())
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            torch.testing.assert_close(b, stats[n], atol=1e-4, rtol=1e-3)
        elif n.endswith("num_batches_tracked"):
            assert int(b) == int(stats[n])
    m.per_scan_statistics = False                      # the whole-batch mode is a different BatchNorm
    o2, _ = m(batch)
    assert float((o2.detach() - obj.detach()).abs().max()) > 1e-3


# ------------------------------------------------------------------------------------------------- cell-list ball query
def _bq_cloud(B, N, kind, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "ball":
        p = torch.randn(B, N, 3, generator=g)
        p = p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)
    elif kind == "plane":                       # degenerate extent along z: one cell layer
        p = torch.rand(B, N, 3, generator=g) * 2 - 1
        p[..., 2] = 0.25
    elif kind == "dup":                         # many exact duplicates
        p = (torch.rand(B, N // 8, 3, generator=g) * 2 - 1).repeat(1, 8, 1)
    elif kind == "clustered":                   # crowded balls: exercises the index-order fallback inside the kernel
        p = torch.randn(B, N, 3, generator=g) * 0.05
        p[:, ::7] = torch.rand(B, (N + 6) // 7, 3, generator=g) * 2 - 1
    else:
        p = torch.rand(B, N, 3, generator=g) * 2 - 1
    return p.contiguous()


@pytest.mark.parametrize("B,N,m,ns,r,kind", [
    (3, 4000, 512, 16, 0.1, "ball"), (2, 8000, 512, 32, 0.2, "ball"), (2, 50000, 1024, 64, 0.2, "ball"),
    (2, 2048, 300, 16, 0.05, "cube"), (2, 5000, 200, 64, 0.3, "plane"), (2, 4096, 256, 32, 0.15, "dup"),
    (2, 6000, 333, 64, 0.2, "clustered"), (1, 20000, 100, 128, 0.4, "ball"), (2, 3000, 64, 8, 5.0, "cube"),
    (2, 3000, 64, 8, 1e-4, "cube"), (1, 2500, 50, 200, 0.5, "cube")])
def test_cell_list_ball_query_is_bit_exact(B, N, m, ns, r, kind):
    """pn2_ball_query_ws (cells of edge >= r, 27-cell candidate lists, rank sort by index) == the reference semantics
    (oracle) == the index-order scan kernel, incl. centres outside the cloud, empty balls, duplicates, a radius larger
    than the cloud, crowded balls (in-kernel fallback) and nsample beyond the collection cap."""
    from pointnet2_ops import _ext
    xyz = _bq_cloud(B, N, kind, seed=N + m)
    g = torch.Generator().manual_seed(7)
    sel = torch.randint(0, N, (B, m), generator=g)
    new_xyz = xyz[torch.arange(B)[:, None], sel].clone()
    new_xyz[:, ::5] += torch.randn(B, (m + 4) // 5, 3, generator=g) * r            # centres that are not cloud points
    new_xyz[:, -1] = 40.0                                                         # far outside the bounding box
    new_xyz = new_xyz.contiguous()
    want = oracle_ext.OracleRowsExt.ball_query(new_xyz, xyz, r, ns)
    results = {}
    for mode in ("force", "slabs", False, True):        # cell list / slab cell lists forced, index-order scan, the library's choice
        _ext.BALL_QUERY_GRID = mode
        try:
            results[mode] = _ext.ball_query(new_xyz.cuda(), xyz.cuda(), r, ns).cpu()
        finally:
            _ext.BALL_QUERY_GRID = True
    for mode, got in results.items():
        assert torch.equal(got, want), mode


# ------------------------------------------------------------------------------------------------- GPU data preparation
@pytest.mark.parametrize("n_obj,P,t_obj,t_rel", [(4, 60000, 1000, 2000), (3, 20000, 4000, 8000), (2, 5000, 300, 700)])
def test_gpu_scan_preparation_matches_the_numpy_restatement(n_obj, P, t_obj, t_rel):
    """crops / boxes / filters / mask channel / zero_mean of data_preparation_utils.py on the GPU vs
    the numpy restatement (tests/prep_oracle.py): boxes and member counts exact, selected indices bit-exact, the
    normalised clouds within 1e-5, the batch layout of collate_fn."""
    import prep_oracle
    from scene_graph_prediction.scene_graph_helpers.dataset import gpu_preparation as gp
    pts, masks = gp.synthetic_fused_scan(n_obj, P, seed=n_obj, device="cuda")
    names = ["Patient", "instrument_table", "human_1", "anesthesia_equipment"][:n_obj]
    batch = gp.prepare_scan(pts, masks, n_obj, t_obj, t_rel, padding=0.2, seed=77, object_names=names)
    obj, rel, boxes, sel, counts, edges = prep_oracle.prepare(pts.cpu().numpy(), masks.cpu().numpy(), n_obj, t_obj, t_rel, 0.2, 77)
    assert np.array_equal(batch["edge_indices"].cpu().numpy(), edges)
    np.testing.assert_array_equal(batch["prep"]["boxes"].cpu().numpy(), boxes)
    np.testing.assert_array_equal(batch["prep"]["members"].cpu().numpy(), counts)
    np.testing.assert_array_equal(batch["prep"]["selection"].cpu().numpy(), sel)
    assert batch["obj_points"].shape == (n_obj, 6, t_obj) and batch["rel_points"].shape == (n_obj * (n_obj - 1), 7, t_rel)
    np.testing.assert_allclose(batch["obj_points"].permute(0, 2, 1).cpu().numpy(), obj, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(batch["rel_points"].permute(0, 2, 1).cpu().numpy(), rel, atol=1e-5, rtol=1e-5)
    # zero_mean: centred, unit sphere; mask channel in {0, 1, 2}; one-hot = two ones per edge
    xyz = batch["rel_points"][:, :3]
    assert float(xyz.mean(dim=2).abs().max()) < 1e-4 and abs(float(xyz.norm(dim=1).max()) - 1.0) < 1e-5
    assert set(batch["rel_points"][:, 6].unique().tolist()) <= {0.0, 1.0, 2.0}
    assert bool((batch["relation_objects_one_hot"].sum(1) == 2).all())
    # the under-populated object (2500 points < target in the first two cases) was up-sampled with replacement
    if t_obj > 2500:
        first = batch["prep"]["selection"][:t_obj]
        assert len(torch.unique(first)) < t_obj


def test_prepared_scan_runs_through_the_model():
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset import gpu_preparation as gp
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    torch.manual_seed(0)
    model = SGPNModelWrapper(config_loader("no_gt.json"), 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)),
                             RELATION_NAMES).cuda().eval()
    pts, masks = gp.synthetic_fused_scan(5, 120000, seed=1, device="cuda")
    names = ["Patient", "operating_table", "human_0", "instrument", "secondary_table"]
    batch = gp.prepare_scan(pts, masks, 5, 4000, 8000, seed=3, object_names=names, scan_id="prep_000001")
    scan_id, triples = model.predict_step(batch)
    assert scan_id == "prep_000001" and all(len(t) == 3 for t in triples)


# ------------------------------------------------------------------------------------------------- (f)4
@pytest.mark.parametrize("n,density,seed", [(1, 0.5, 0), (2, 1.0, 1), (5, 0.3, 2), (13, 0.15, 3), (14, 0.12, 4), (20, 0.1, 5),
                                            (31, 0.05, 6), (40, 0.04, 7), (9, 0.0, 8)])
def test_graphormer_algos_match_the_compiled_reference(n, density, seed):
    """floyd_warshall / gen_edge_input (role_prediction/graphormer/algos.pyx:11-89) on the GPU == the reference's own
    module: hop distances, intermediate vertices (incl. vertex 12 colliding with the MAX_DIST marker for n > 12 and
    vertex 0 never expanded) and the edge features along every path, all int64, bit for bit; also batched.  The expected
    values are a fixture generated in the build container from the reference's compiled Cython module
    (tests/golden/make_algos_golden.py -> tests/golden/algos.npz); nothing built from the reference is imported here."""
    import role_prediction.graphormer.algos as algos
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "algos.npz"))
    key = f"n{n}_s{seed}"
    adj, feat = fx[key + "_adj"], fx[key + "_feat"]
    Mr, Pr = fx[key + "_M"], fx[key + "_P"]
    M, P = algos.floyd_warshall(adj)
    assert M.dtype == np.int64 and np.array_equal(M, Mr) and np.array_equal(P, Pr)
    md = int(np.amax(Mr)) if n else 0
    if md > 0:
        Er = fx[key + "_E"]
        E = algos.gen_edge_input(md, P, feat)
        assert E.shape == Er.shape and np.array_equal(E, Er)
    # a batch of graphs in one launch
    adjs = torch.from_numpy(np.stack([adj, adj.T, np.zeros_like(adj)])).cuda()
    Mb, Pb = algos.floyd_warshall(adjs)
    for b, tag in enumerate(("", "_T", "_Z")):
        assert np.array_equal(Mb[b].cpu().numpy(), fx[key + tag + "_M"]) and np.array_equal(Pb[b].cpu().numpy(), fx[key + tag + "_P"])


def test_instance_label_fps_call_shape():
    """compute_instance_labels.py:95,195: furthest_point_sample on ONE whole-object cloud (1, n, 3) -> 200 samples
    (NPOINTS), n = a few 1e5 points in metric units: bit-exact against the oracle."""
    from pointnet2_ops import pointnet2_utils as pu2
    g = torch.Generator().manual_seed(12)
    pts = torch.rand(1, 180000, 3, generator=g) * torch.tensor([2000.0, 900.0, 600.0])       # millimetres, like the scans
    want = oracle_ext.OracleRowsExt.furthest_point_sampling(pts, 200)
    got = pu2.furthest_point_sample(pts.cuda(), 200)[0].cpu()
    assert torch.equal(got, want[0])


@pytest.mark.parametrize("momentum", [0.1, 0.5])
def test_running_statistics_kernel_equals_sequential_batch_norm_calls(momentum):
    """pn2_bn_running_update: the S momentum updates of S single-scan steps in one launch == S training-mode calls of
    torch's BatchNorm on the scans, in order (running_mean, running_var with the unbiased variance, num_batches_tracked)."""
    from pointnet2_ops import fused_mlp
    g = torch.Generator().manual_seed(11)
    C, rows = 200, [40, 7, 130, 2, 65, 1000]
    xs = [torch.randn(n, C, generator=g) * (1 + i) + i for i, n in enumerate(rows)]
    ref = torch.nn.BatchNorm2d(C, momentum=momentum).train()
    ref.running_mean.normal_(generator=g); ref.running_var.uniform_(0.5, 2.0, generator=g)
    bn = copy.deepcopy(ref).cuda()
    for x in xs:
        ref(x.t().reshape(1, C, -1, 1))
    F = torch.stack([torch.stack([x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + bn.eps), torch.zeros(C), torch.zeros(C)])
                     for x in xs]).cuda()
    fused_mlp._update_running_stats([(None, bn)], [F], rows)
    torch.testing.assert_close(bn.running_mean.cpu(), ref.running_mean, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(bn.running_var.cpu(), ref.running_var, atol=1e-4, rtol=1e-4)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == len(rows)
