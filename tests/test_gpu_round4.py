"""GPU parity cases added in round 4.

* the HEADLINE's route through an SA level — crowded balls, so first layer without its output
  (`pn2_mlp_gemm_first`), pooled last layer without its output (`pn2_mlp_gemm_pool`), Gram-form pooled backward
  (`pn2_pool_bwd`), first-layer fold (`pn2_mlp_bwd_fused_fold_first`) and, for a level whose features need a gradient, the
  inverse-index feature-gradient sum (`pn2_group_rows_grad_csr`) — against the ORACLE backend at module level (until now
  that route was compared with fp32 torch and with other HIP kernels only; the oracle backbone tests use 3 000-point
  clouds whose SA1 is sparse).
"""
import copy

import pytest
import torch

import oracle_ext
from pointnet2_ops import pointnet2_utils as pu

pytestmark = pytest.mark.gpu


def _with_backend(backend, fn):
    saved = pu._ext
    pu._ext = backend
    try:
        return fn()
    finally:
        pu._ext = saved


def _unit_ball(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    return p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)


class _Calls:
    """Counts the calls of the named `_ext` entry points while the HIP backend runs (which route did the level take?)."""

    def __init__(self, ext, names):
        self.ext, self.names, self.count, self.saved = ext, names, {n: 0 for n in names}, {}

    def __enter__(self):
        for n in self.names:
            fn = getattr(self.ext, n)
            self.saved[n] = fn

            def wrapped(*a, _fn=fn, _n=n, **k):
                self.count[_n] += 1
                return _fn(*a, **k)
            setattr(self.ext, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.ext, n, fn)
        return False


def test_sa1_at_crowded_density_takes_the_headline_route_and_matches_the_oracle():
    """The backbone's SA1 (2048 centres, r 0.2, 64 samples, [3+3, 64, 64, 128], normalize_xyz) on 2 x 40 000 points of the
    unit ball: N r^3 = 320 > 4 nsample, the density of the headline batch (50k: 400).  Train mode, forward + backward, HIP
    against the oracle backend: indices bit-exact, features and every parameter gradient within 1e-4."""
    from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from pointnet2_ops import _ext
    torch.manual_seed(5)
    sa = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True, normalize_xyz=True).train()
    xyz = _unit_ball(2, 40000, 11)
    rgb = torch.rand(2, 3, 40000, generator=torch.Generator().manual_seed(12))
    gout = torch.randn(2, 128, 2048, generator=torch.Generator().manual_seed(13))

    def run(dev, backend):
        m = copy.deepcopy(sa).to(dev)
        nx, nf, inds = _with_backend(backend, lambda: m(xyz.to(dev), rgb.to(dev)))
        _with_backend(backend, lambda: (nf * gout.to(dev)).sum().backward())
        return nx.cpu(), nf.detach().cpu(), inds.cpu(), {n: p.grad.cpu() for n, p in m.named_parameters()}, \
            {n: b.cpu() for n, b in m.named_buffers()}

    nx_r, nf_r, inds_r, g_r, b_r = run("cpu", oracle_ext.OracleRowsExt)
    names = ["mlp_gemm_first", "mlp_gemm_pool", "pool_bwd", "mlp_bwd_fused_fold_first", "first_layer_dw"]
    with _Calls(_ext, names) as calls:
        nx_g, nf_g, inds_g, g_g, b_g = run("cuda", _ext)
    assert all(calls.count[n] == 1 for n in names), calls.count          # the headline's kernels, once each
    assert torch.equal(inds_g, inds_r) and torch.equal(nx_g, nx_r)
    err = float((nf_g - nf_r).abs().max())
    print(f"\n[crowded SA1] features: max abs err {err:.3e} (max |ref| {float(nf_r.abs().max()):.2f})", end="")
    torch.testing.assert_close(nf_g, nf_r, atol=1e-4, rtol=1e-4)
    for k in g_r:
        e = float((g_g[k] - g_r[k]).abs().max())
        print(f"\n[crowded SA1] d{k}: max abs err {e:.3e} (max |ref| {float(g_r[k].abs().max()):.3f})", end="")
        torch.testing.assert_close(g_g[k], g_r[k], atol=1e-4, rtol=1e-3)
    for k in b_r:                                                          # running statistics of the three BatchNorms
        torch.testing.assert_close(b_g[k].float(), b_r[k].float(), atol=1e-5, rtol=1e-4)


def test_sa2_at_crowded_density_with_feature_gradient_matches_the_oracle():
    """The backbone's SA2 (1024 centres, r 0.4, 32 samples, [128+3, 128, 128, 256]) on 2 x 2 500 points: N r^3 = 160 > 4
    nsample; the 128 feature channels need a gradient, so the level takes the pooled layer, `pn2_pool_bwd` for K = 128 and,
    with a prefetched geometry, the inverse-index per-point sum.  HIP against the oracle backend: features, the input
    feature gradient and every parameter gradient within 1e-4."""
    from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from pointnet2_ops import _ext
    torch.manual_seed(6)
    sa = PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True).train()
    xyz = _unit_ball(2, 2500, 21)
    feats = torch.randn(2, 128, 2500, generator=torch.Generator().manual_seed(22))
    gout = torch.randn(2, 256, 1024, generator=torch.Generator().manual_seed(23))

    def run(dev, backend, prefetch):
        m = copy.deepcopy(sa).to(dev)
        f = feats.to(dev).requires_grad_(True)
        x = xyz.to(dev)

        def fwd():
            geo = m.sample_and_query(x, inverse_index=True) if prefetch else None
            return m(x, f, geometry=geo)
        nx, nf, inds = _with_backend(backend, fwd)
        _with_backend(backend, lambda: (nf * gout.to(dev)).sum().backward())
        return nf.detach().cpu(), inds.cpu(), f.grad.cpu(), {n: p.grad.cpu() for n, p in m.named_parameters()}

    nf_r, inds_r, gf_r, g_r = run("cpu", oracle_ext.OracleRowsExt, False)
    with _Calls(_ext, ["mlp_gemm_pool", "pool_bwd", "group_rows_grad_csr"]) as calls:
        nf_g, inds_g, gf_g, g_g = run("cuda", _ext, True)
    assert calls.count == {"mlp_gemm_pool": 1, "pool_bwd": 1, "group_rows_grad_csr": 1}, calls.count
    assert torch.equal(inds_g, inds_r)
    torch.testing.assert_close(nf_g, nf_r, atol=1e-4, rtol=1e-4)
    e = float((gf_g - gf_r).abs().max())
    print(f"\n[crowded SA2] d features: max abs err {e:.3e} (max |ref| {float(gf_r.abs().max()):.3f})", end="")
    torch.testing.assert_close(gf_g, gf_r, atol=1e-4, rtol=1e-3)
    for k in g_r:
        torch.testing.assert_close(g_g[k], g_r[k], atol=1e-4, rtol=1e-3)
