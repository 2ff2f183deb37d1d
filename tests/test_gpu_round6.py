"""Round 6 GPU parity.

1. The reference-generated GCN / whole-model fixtures (tests/golden/triplet_gcn.npz, sgpn.npz: the reference's own
   network_TripletGCN.py:30-80 and scene_graph_prediction_model.py:31-141 run in the build container) through the HIP
   kernels, every TripletGCN route.
"""
import numpy as np
import pytest
import torch

import fixture_checks as fc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("route", ["fused", "lifted", "concat"])
def test_triplet_gcn_model_matches_reference_class_on_gpu(route, monkeypatch):
    from pointnet2_ops import _ext
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    monkeypatch.setattr(gcn, "FUSED_LAYER", route == "fused")
    monkeypatch.setattr(gcn, "LIFT_MIN_EDGES", 0 if route == "lifted" else 1 << 60)
    calls = {"n": 0}
    real = _ext.gcn_layer_forward
    monkeypatch.setattr(_ext, "gcn_layer_forward", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real(*a, **k))[1])
    z = fc.load("triplet_gcn.npz")
    fc.check_gcn_model(z, "l2", 2, (256, 256, 512), 71, "cuda", atol=1e-4, rtol=1e-3)
    if route == "fused":
        assert calls["n"] >= 4                      # train + eval forward of both layers went through the fused kernels
    fc.check_gcn_model(z, "l3", 3, (64, 48, 96), 72, "cuda", atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("route", ["default", "concat"])
def test_triplet_gcn_layer_irregular_and_hand_case_on_gpu(route, monkeypatch):
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    if route == "concat":
        monkeypatch.setattr(gcn, "FUSED_LAYER", False)
        monkeypatch.setattr(gcn, "LIFT_MIN_EDGES", 1 << 60)
    fc.check_gcn_layer_cases(fc.load("triplet_gcn.npz"), "cuda", atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("tag,seed", [("no_gt", 82), ("no_gt_image", 83)])
def test_sgpn_matches_reference_class_on_gpu(tag, seed):
    """Whole model, both configs: ordered manifest, eval forward (encoder features at 1e-4; behind the 3-row BatchNorms of
    the GCN at 1e-3, see fixture_checks.check_sgpn), train-mode loss and gradient norms."""
    fc.check_sgpn(fc.load("sgpn.npz"), tag, seed, "cuda", atol=1e-4, rtol=1e-3, loss_tol=1e-3)
