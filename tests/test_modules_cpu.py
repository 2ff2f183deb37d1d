"""Host logic of the product's python layer on CPU (oracle backend): the rows fast
path equals the literal (reference-staged) path, BN bookkeeping matches torch's
modules, state_dict layout is the reference's."""
import copy

import pytest
import torch

from pointnet2_ops import pointnet2_modules as pm
from pointnet2_ops import pointnet2_utils as pu


def _cloud(B, N, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    pc = torch.rand(B, N, 3 + C, generator=g) * 2 - 1
    return pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()


def _both_paths(module, fn):
    outs = []
    for fast in (False, True):
        m = copy.deepcopy(module)
        prev = pm.set_fast_path(fast)
        try:
            outs.append(fn(m))
        finally:
            pm.set_fast_path(prev)
    return outs


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("normalize", [False, True])
def test_sa_msg_rows_equals_literal(oracle_backend, train, normalize):
    xyz, feats = _cloud(2, 300, 3)
    torch.manual_seed(3)
    sa = pm.PointnetSAModuleMSG(npoint=40, radii=[0.3, 0.6], nsamples=[4, 9],
                                mlps=[[3, 8, 12], [3, 8, 16]], normalize_xyz=normalize)
    sa.train(train)

    def run(m):
        f = feats.clone().requires_grad_(True)
        nx, nf = m(xyz, f)
        (nf * torch.linspace(0.5, 1.5, nf.numel()).view_as(nf)).sum().backward()
        grads = [p.grad.clone() for p in m.parameters()]
        return nx, nf.detach().contiguous(), f.grad.clone(), grads, copy.deepcopy(m.state_dict())

    (nx0, nf0, gf0, gp0, sd0), (nx1, nf1, gf1, gp1, sd1) = _both_paths(sa, run)
    assert torch.equal(nx0, nx1)
    assert nf0.shape == nf1.shape == (2, 28, 40)
    torch.testing.assert_close(nf1, nf0, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(gf1, gf0, atol=1e-5, rtol=1e-4)
    for a, b in zip(gp1, gp0):
        torch.testing.assert_close(a, b, atol=1e-4, rtol=1e-3)
    assert sd0.keys() == sd1.keys()
    for k in sd0:                                        # running stats / num_batches_tracked bookkeeping
        torch.testing.assert_close(sd1[k].float(), sd0[k].float(), atol=1e-5, rtol=1e-4)


def test_group_all_rows_equals_literal(oracle_backend):
    xyz, feats = _cloud(3, 50, 5, seed=4)
    torch.manual_seed(5)
    sa = pm.PointnetSAModule(mlp=[5, 16, 8], use_xyz=True)

    def run(m):
        nx, nf = m(xyz, feats)
        return nx, nf.detach().contiguous()

    (nx0, nf0), (nx1, nf1) = _both_paths(sa, run)
    assert nx0 is None and nx1 is None
    assert nf0.shape == (3, 8, 1)
    torch.testing.assert_close(nf1, nf0, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("with_skip", [True, False])
def test_fp_rows_equals_literal(oracle_backend, with_skip):
    unknown, uf = _cloud(2, 60, 4, seed=6)
    known, kf = _cloud(2, 25, 7, seed=7)
    torch.manual_seed(8)
    fp = pm.PointnetFPModule(mlp=[7 + (4 if with_skip else 0), 16, 8])

    def run(m):
        k = kf.clone().requires_grad_(True)
        out = m(unknown, known, uf if with_skip else None, k)
        out.square().sum().backward()
        return out.detach().contiguous(), k.grad.clone()

    (o0, g0), (o1, g1) = _both_paths(fp, run)
    assert o0.shape == (2, 8, 60)
    torch.testing.assert_close(o1, o0, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(g1, g0, atol=1e-5, rtol=1e-4)


def test_rows_outputs_feed_next_module_without_copy(oracle_backend):
    xyz, feats = _cloud(2, 200, 3, seed=9)
    sa = pm.PointnetSAModule(mlp=[3, 8], npoint=32, radius=0.5, nsample=8)
    _, nf = sa(xyz, feats)
    assert nf.shape == (2, 8, 32)
    assert nf.transpose(1, 2).is_contiguous()            # rows layout underneath
    assert pu.as_rows(nf).data_ptr() == nf.data_ptr()    # zero-copy hand-over


def test_state_dict_layout_and_spec_mutation_match_reference():
    spec = [[3, 64, 64], [3, 64, 128]]
    sa = pm.PointnetSAModuleMSG(npoint=512, radii=[0.1, 0.2], nsamples=[16, 32], mlps=spec)
    assert spec[0][0] == 6 and spec[1][0] == 6          # OPS/pointnet2_modules.py:112-113
    keys = list(sa.state_dict().keys())
    assert keys[:6] == ["mlps.0.0.weight", "mlps.0.1.weight", "mlps.0.1.bias", "mlps.0.1.running_mean",
                        "mlps.0.1.running_var", "mlps.0.1.num_batches_tracked"]
    assert sa.state_dict()["mlps.1.3.weight"].shape == (128, 64, 1, 1)
    fp = pm.PointnetFPModule(mlp=[10, 4])
    assert list(fp.state_dict().keys())[0] == "mlp.0.weight"


def test_literal_path_serves_coordinate_gradients(oracle_backend):
    xyz, feats = _cloud(1, 40, 2, seed=11)
    xyz.requires_grad_(True)
    sa = pm.PointnetSAModule(mlp=[2, 4], npoint=8, radius=0.8, nsample=4)
    _, nf = sa(xyz, feats)
    nf.sum().backward()
    assert xyz.grad is not None and xyz.grad.abs().sum() > 0


def test_query_and_group_unique_cnt_needs_sample_uniformly():
    """ret_unique_cnt without sample_uniformly is the reference's assertion (GF3D/pointnet2/pointnet2_utils.py:309-310)."""
    with pytest.raises(AssertionError):
        pu.QueryAndGroup(0.2, 8, ret_unique_cnt=True)
    q = pu.QueryAndGroup(0.2, 8, sample_uniformly=True, ret_unique_cnt=True)
    assert q.sample_uniformly and q.ret_unique_cnt
