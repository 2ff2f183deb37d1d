"""Fused fp32-MFMA shared-MLP kernels (csrc/mlp_gemm.hip) against the torch ops they
replace (Linear/BatchNorm/ReLU/max on the same rows, same parameters): forward,
running statistics, and every gradient."""
import copy

import pytest
import torch

from pointnet2_ops import pointnet2_modules as pm

pytestmark = pytest.mark.gpu


def _mlp(spec, seed):
    torch.manual_seed(seed)
    m = pm.build_shared_mlp(list(spec)).cuda()
    with torch.no_grad():
        for layer in m:
            if isinstance(layer, torch.nn.BatchNorm2d):
                layer.weight.uniform_(0.5, 1.5)
                layer.bias.uniform_(-0.3, 0.3)
                layer.running_mean.uniform_(-0.2, 0.2)
                layer.running_var.uniform_(0.5, 1.5)
    return m


def _run(mlp, x, ns, fused, train, x_grad=True):
    m = copy.deepcopy(mlp)
    m.train(train)
    prev = pm.set_fused_mlp(fused)
    try:
        xx = x.clone().requires_grad_(x_grad)
        if ns:
            R = x.size(0) // ns
            out = pm.mlp_pool_rows(m, xx.view(1, R, ns, -1)).view(R, -1)
        else:
            out = pm.mlp_rows(m, xx)
        w = torch.linspace(0.5, 1.5, out.numel(), device=out.device).view_as(out)
        (out * w).sum().backward()
        return out.detach(), xx.grad, [p.grad for p in m.parameters()], m.state_dict()
    finally:
        pm.set_fused_mlp(prev)


CASES = [((6, 64, 64, 128), 64 * 40, 64), ((131, 128, 128, 256), 32 * 50, 32), ((259, 128, 128, 256), 16 * 70, 16),
         ((512, 256, 288), 1000, 0), ((512, 256, 256), 777, 0), ((5, 16), 300, 0), ((9, 32, 40), 24 * 12, 12),
         ((259, 256, 256), 3 * 128, 128),
         # hidden layers on the one-pass backward kernel (csrc/mlp_bwd_fused.hip) with ragged N / K and a partial
         # last row tile, pooled (ns = 16) and unpooled
         ((20, 72, 96, 100), 16 * 37, 16), ((30, 48, 120, 64), 777, 0), ((7, 128, 40, 128, 128), 32 * 33, 32),
         # first-layer wgrad with the three leading xyz columns reduced on the VALU side (needs M >= 2^18)
         ((131, 128, 64), 8192 * 32 + 64, 32), ((258, 64, 64), 1 << 18, 0)]


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("spec,M,ns", CASES)
def test_fused_matches_torch(spec, M, ns, train):
    mlp = _mlp(spec, seed=len(spec) + M)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, spec[0], generator=g).cuda()
    o0, gx0, gp0, sd0 = _run(mlp, x, ns, fused=False, train=train)
    o1, gx1, gp1, sd1 = _run(mlp, x, ns, fused=True, train=train)
    torch.testing.assert_close(o1, o0, atol=1e-4, rtol=1e-4)
    # A pre-activation within rounding of the ReLU threshold flips its mask between two implementations — and, in training
    # mode, between two RUNS of this one (the fp64 statistics atomics commit in another order: last-bit differences in
    # mean / rstd).  A flipped unit changes one whole row of the input gradient by O(1) and moves the weight-gradient sums
    # by one row's share.  Large M (tens of millions of pre-activations) always has a handful, small M has one in about
    # 1 of 100 runs (seen once in 25 full-suite runs: spec3-1000-0-True): up to two such rows are tolerated, with the
    # bounds in norm that the large cases use.
    bad_rows = int(((gx1 - gx0).abs() > 1e-4 + 1e-3 * gx0.abs()).any(dim=1).sum())
    flips = M >= (1 << 17) or bad_rows > 0
    if flips:
        assert M >= (1 << 17) or bad_rows <= 2, bad_rows
        assert float((gx1 - gx0).norm() / gx0.norm()) < 2e-3
    for a, b in zip(gp1, gp0):
        scale = float(b.abs().max()) + 1e-6
        tol = 5e-3 if flips else 2e-4                     # the mask flips above also move the sums
        assert float((a - b).abs().max()) <= tol * scale + 1e-5, (a - b).abs().max()
    for k in sd0:
        torch.testing.assert_close(sd1[k].float(), sd0[k].float(), atol=1e-5, rtol=1e-4)


# first-layer fold (csrc/mlp_bwd_fused.hip, FOLD): the input rows need no gradient and are <= 8 columns wide, the layer
# above is 32 < N, K <= 64 -> the first layer's weight gradient comes from gz^T X, X^T X and the column sums of X.
FOLD_CASES = [((6, 64, 64, 128), 16 * 37, 16), ((6, 64, 64), 32 * 21, 32), ((7, 64, 64, 128), 777, 0),
              ((3, 48, 40, 64), 1000, 0), ((8, 64, 64), 64 * 3, 0), ((1, 33, 64), 130, 0),
              ((6, 64, 64, 128), 64 * 4096 + 64 * 3, 64)]


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("spec,M,ns", FOLD_CASES)
def test_first_layer_fold_matches_torch(spec, M, ns, train):
    from pointnet2_ops import pointnet2_utils as pu
    assert pu._ext.mlp_bwd_fused_fold_supported(spec[2], spec[1], spec[0])
    mlp = _mlp(spec, seed=len(spec) + M)
    g = torch.Generator().manual_seed(M + 1)
    x = (torch.randn(M, spec[0], generator=g) + 0.3).cuda()        # non-zero column means: the c3 (1^T X) term matters
    o0, _, gp0, _ = _run(mlp, x, ns, fused=False, train=train, x_grad=False)
    o1, _, gp1, _ = _run(mlp, x, ns, fused=True, train=train, x_grad=False)
    torch.testing.assert_close(o1, o0, atol=1e-4, rtol=1e-4)
    for a, b in zip(gp1, gp0):
        scale = float(b.abs().max()) + 1e-6
        tol = 5e-3 if M >= (1 << 17) else 2e-4
        assert float((a - b).abs().max()) <= tol * scale + 1e-5, (a - b).abs().max()


def test_full_size_sa1_layer_statistics_and_throughput_shape():
    """BASELINE size: 32x2048x64 rows through the SA1 stack; checks the batch statistics path at
    4.2M rows (fp64 accumulation) against torch."""
    mlp = _mlp((6, 64, 64, 128), seed=3)
    x = torch.randn(32 * 2048 * 64, 6, device="cuda")
    o0, _, gp0, sd0 = _run(mlp, x, 64, fused=False, train=True)
    o1, _, gp1, sd1 = _run(mlp, x, 64, fused=True, train=True)
    torch.testing.assert_close(o1, o0, atol=2e-4, rtol=1e-3)
    for k in sd0:
        torch.testing.assert_close(sd1[k].float(), sd0[k].float(), atol=1e-5, rtol=1e-4)
    for a, b in zip(gp1, gp0):
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 5e-3
