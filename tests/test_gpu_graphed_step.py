"""Whole-step hipGraph replay (runtime.GraphedTrainStep) of the scene-graph model: the gradients a
replay leaves in the flat buffer equal an eager backward on the same scan, for two alternating scan
shapes (two graphs), and a captured AdamW trajectory trains."""
import copy

import pytest
import torch

from runtime import GraphedTrainStep
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

pytestmark = pytest.mark.gpu


def _model():
    torch.manual_seed(0)
    cfg = config_loader("no_gt.json")
    m = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                   # deterministic comparison
    for n, p in m.named_parameters():
        if ".backbone.fc_layer." in n:
            p.requires_grad_(False)
    return m.cuda().train()


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def test_replayed_gradients_equal_eager_backward_for_two_scan_shapes():
    model = _model()
    ref = copy.deepcopy(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0)                 # weights frozen: every call must reproduce the eager gradient
    stepper = GraphedTrainStep(model.pure_training_step, params, opt)
    scans = [to_device(synthetic_scan(n, 512, 1024, seed=s), "cuda") for n, s in ((9, 1), (10, 2), (9, 3), (10, 4), (9, 5), (10, 6))]   # >= 9 nodes: the GCN BatchNorms over nodes stay well conditioned
    for i, scan in enumerate(scans):
        loss, rel_pred = stepper(scan)
        wants = []
        for _ in range(2):                                # twice: the eager path's own run-to-run spread (fp32 atomics
            ref.zero_grad(set_to_none=True)               # order, amplified by the BatchNorms over a handful of nodes)
            rl, rp = ref.pure_training_step(scan)
            rl.backward()
            wants.append(torch.cat([p.grad.flatten() for p in ref.parameters() if p.requires_grad]))
        want, noise = wants[0], _rel(wants[1], wants[0])
        assert abs(float(loss) - float(rl.detach())) < 1e-3, i
        torch.testing.assert_close(rel_pred, rp, atol=5e-3, rtol=2e-3)   # log-probs behind BatchNorms over 5-6 nodes
        assert _rel(stepper.grads.flat, want) < max(3e-2, 10 * noise), (i, noise)   # a stale-buffer bug would be O(1); 1e-2 was hit once in ~10 suite runs
    assert stepper.num_graphs == 2                        # scans 0/1 ran eagerly, 2/3 captured, 4/5 replayed


def test_prefetched_geometry_is_a_graph_input():
    """batch["geometry"] (nested dict of index tensors computed ahead of the step) is copied into the graph's static
    buffers on every call: replays with the geometry of DIFFERENT scans must reproduce each scan's own eager gradient."""
    model = _model()
    ref = copy.deepcopy(model)
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = GraphedTrainStep(model.pure_training_step, params, torch.optim.SGD(params, lr=0.0))
    scans = [to_device(synthetic_scan(9, 512, 1024, seed=s), "cuda") for s in (11, 12, 13, 14)]
    for i, scan in enumerate(scans):
        batch = dict(scan, geometry=model.precompute_geometry(scan))
        loss, _ = stepper(batch)
        ref.zero_grad(set_to_none=True)
        rl, _ = ref.pure_training_step(scan)              # eager, geometry computed inside the forward
        rl.backward()
        want = torch.cat([p.grad.flatten() for p in ref.parameters() if p.requires_grad])
        assert abs(float(loss) - float(rl.detach())) < 1e-3, i
        assert _rel(stepper.grads.flat, want) < 3e-2, i   # the geometry of the previous scan would give O(1)
    assert stepper.num_graphs == 1                        # scan 0 eager, scan 1 captured, scans 2-3 replayed


def test_captured_adamw_trajectory_trains():
    model = _model()
    params = [p for p in model.parameters() if p.requires_grad]
    before = torch.cat([p.detach().flatten() for p in params]).clone()
    opt = torch.optim.AdamW(params, lr=1e-3, capturable=True)
    stepper = GraphedTrainStep(model.pure_training_step, params, opt)
    scan = to_device(synthetic_scan(3, 1024, 2048, seed=7), "cuda")
    losses = [float(stepper(scan)[0]) for _ in range(12)]
    assert stepper.num_graphs == 1
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    after = torch.cat([p.detach().flatten() for p in params])
    assert float((after - before).abs().max()) > 0


def test_non_capturable_adam_is_rejected():
    model = _model()
    params = [p for p in model.parameters() if p.requires_grad]
    with pytest.raises(ValueError, match="capturable=True"):
        GraphedTrainStep(model.pure_training_step, params, torch.optim.AdamW(params, lr=1e-3))


def test_geometry_prefetcher_yields_identical_steps():
    """runtime.GeometryPrefetcher: every yielded batch carries its OWN scan's geometry (computed on the side stream) and
    a forward with it equals the plain forward."""
    from runtime import GeometryPrefetcher
    model = _model().eval()
    scans = [to_device(synthetic_scan(n, 512, 1024, seed=s), "cuda") for n, s in ((9, 21), (7, 22), (9, 23))]
    seen = 0
    with torch.no_grad():
        for scan, batch in zip(scans, GeometryPrefetcher(model.precompute_geometry, iter(scans))):
            assert batch["obj_points"] is scan["obj_points"] and "geometry" in batch
            a_obj, a_rel = model(batch)
            b_obj, b_rel = model(scan)
            assert torch.equal(a_obj, b_obj) and torch.equal(a_rel, b_rel)
            seen += 1
    assert seen == len(scans)


def test_block_diagonal_batches_are_graph_inputs():
    """A collated batch of scans carries a SceneBatch object (row offsets on the device + per-scan counts on the host): its
    tensors are copied into the graph's static buffers like any other input, its counts are part of the signature.  Replays
    with DIFFERENT scans of the same shape reproduce each batch's own eager gradient (per-scan statistics in training)."""
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans
    model = _model()
    ref = copy.deepcopy(model)
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = GraphedTrainStep(model.pure_training_step, params, torch.optim.SGD(params, lr=0.0))
    batches = [to_device(collate_scans([synthetic_scan(9, 512, 1024, seed=3 * k + i) for i in range(2)]), "cuda") for k in range(4)]
    other = to_device(collate_scans([synthetic_scan(n, 512, 1024, seed=40 + n) for n in (8, 10)]), "cuda")   # same totals? no: 18 nodes
    for i, batch in enumerate(batches):
        loss, _ = stepper(batch)
        ref.zero_grad(set_to_none=True)
        rl, _ = ref.pure_training_step(batch)
        rl.backward()
        want = torch.cat([p.grad.flatten() for p in ref.parameters() if p.requires_grad])
        assert abs(float(loss) - float(rl.detach())) < 1e-3, i
        assert _rel(stepper.grads.flat, want) < 3e-2, i
    assert stepper.num_graphs == 1
    # 8 + 10 objects: the same 18 node rows but other per-scan counts (and 56 + 90 edges): a different signature
    stepper(other)
    stepper(other)
    assert stepper.num_graphs == 2
