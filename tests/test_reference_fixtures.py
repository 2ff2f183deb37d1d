"""CPU (oracle backend): the product's TripletGCN / TripletGCNModel / SGPNModelWrapper against fixtures generated from
the REFERENCE's own classes (SGH/model/gcns/network_TripletGCN.py:30-80, SGH/model/scene_graph_prediction_model.py:31-141;
tests/golden/make_golden.py `triplet_gcn`, `sgpn`).  Pins the python layer: registration / initialisation order,
state_dict keys, MessagePassing(source_to_target) index roles, between-layer ReLU, batch-statistic BatchNorm in eval,
loss.  The same checks run through the HIP kernels in tests/test_gpu_round6.py."""
import pytest
import torch

import fixture_checks as fc


def test_triplet_gcn_model_matches_reference_class(oracle_backend):
    z = fc.load("triplet_gcn.npz")
    fc.check_gcn_model(z, "l2", 2, (256, 256, 512), 71, "cpu", atol=2e-5, rtol=1e-4)
    fc.check_gcn_model(z, "l3", 3, (64, 48, 96), 72, "cpu", atol=2e-5, rtol=1e-4)


def test_triplet_gcn_layer_irregular_and_hand_case(oracle_backend):
    fc.check_gcn_layer_cases(fc.load("triplet_gcn.npz"), "cpu")


@pytest.mark.parametrize("tag,seed", [("no_gt", 82), ("no_gt_image", 83)])
def test_sgpn_manifest_is_the_reference_order(tag, seed):
    """Ordered keys / shapes / seeded sums: `full_image_feature_reduction` sits between the encoders and the GCN like in
    the reference (:47-57), so construction draws from the RNG in the reference's order."""
    z = fc.load("sgpn.npz")
    model = fc.build_sgpn(z, tag, seed)
    fc.check_manifest(model, z, f"{tag}/")
    assert [k for k, _ in model.named_parameters()] == [str(k) for k in z[f"{tag}/param_names"]]


@pytest.mark.parametrize("tag,seed", [("no_gt", 82), ("no_gt_image", 83)])
def test_sgpn_forward_loss_gradients_match_reference_class(oracle_backend, tag, seed):
    fc.check_sgpn(fc.load("sgpn.npz"), tag, seed, "cpu", atol=2e-5, rtol=1e-4, loss_tol=1e-4)


@pytest.mark.parametrize("fast", [False, True])
def test_gf3d_backbone_eval_mode_matches_reference_class(oracle_backend, fast):
    """The inference path (eval-mode BatchNorm on running statistics, EXT/GF3D models/backbone_module.py:95-135 under
    main.py:84-117) of the headline backbone against the reference's own class (tests/golden/make_golden.py
    `gf3d_backbone_eval`), literal and rows python paths on the oracle backend."""
    from pointnet2_ops import pointnet2_modules as pm
    prev = pm.set_fast_path(fast)
    try:
        fc.check_gf3d_eval(fc.load("gf3d_backbone_eval.npz"), "cpu", atol=2e-5, rtol=1e-4)
    finally:
        pm.set_fast_path(prev)
