"""Round-3 kernels on the GPU, through the C ABI: the max-pooled last layer without its materialised output
(pn2_mlp_gemm_pool + pn2_pool_finalize) and the Gram-form backward of that layer (pn2_pool_bwd_*)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ext():
    from pointnet2_ops import _ext
    return _ext


@pytest.mark.parametrize("ns", [16, 32, 64, 128])
@pytest.mark.parametrize("K,N", [(64, 128), (128, 256), (64, 64), (48, 96), (16, 32), (128, 288)])
def test_pooled_layer_without_materialised_output_equals_gemm_plus_rows_max(ns, K, N):
    """max_s relu(bn(y_s)) from the fused epilogue == the materialised y followed by pn2_bn_relu_rows_max
    (OPS/pointnet2_modules.py:58-70): same pooled values, same raw value at the arg-max, same statistics; the arg-max
    itself may differ only where two rows tie after BatchNorm's rounding."""
    e = _ext()
    dev = torch.device("cuda:0")
    torch.manual_seed(ns * 1000 + K + N)
    R = 37 if ns <= 32 else 11                     # partial last 128-row tile for ns = 16 / 32
    M = R * ns
    x = torch.randn(M, K, device=dev)
    # ball-query padding: repeated rows inside a group give exact ties -> the FIRST one must win
    xg = x.view(R, ns, K)
    xg[:, ns // 2:] = xg[:, :1]
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    gamma = torch.randn(N, device=dev)             # about half of the columns have a negative scale
    gamma[0] = 0.0
    beta = torch.randn(N, device=dev) * 0.1

    stats_ref = torch.zeros(2, N, dtype=torch.float64, device=dev)
    y = e.mlp_gemm(x, W, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=p, stats=stats_ref)
    fin = e.bn_finalize(stats_ref, M, gamma, beta, 1e-5, 0.0, None, None, None)
    out_ref, arg_ref, yraw_ref = e.bn_relu_rows_max(y, fin, ns)

    Wf, sgn = e.pool_flip_rows(W, gamma)
    assert torch.equal(sgn, torch.where(gamma < 0, -1.0, 1.0))
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    pmax, parg = e.mlp_gemm_pool(x, Wf, sgn, ns, p=p, stats=stats)
    fin2 = e.bn_finalize(stats, M, gamma, beta, 1e-5, 0.0, None, None, None)
    out, arg, yraw = e.pool_finalize(pmax, parg, fin2, sgn, ns)

    assert torch.allclose(stats, stats_ref, rtol=1e-5, atol=1e-3)         # fp32 partial sums, other association
    assert torch.allclose(fin2, fin, rtol=1e-4, atol=1e-5)
    out_chk, _, _ = e.pool_finalize(pmax, parg, fin, sgn, ns)          # same constants: bit-identical pooled values
    assert torch.equal(out_chk, out_ref)
    # the picked row holds the extreme raw value of its column in the direction of the scale's sign
    assert torch.equal((y.view(R, ns, N) * sgn).max(1).values, yraw * sgn)
    live = (out_ref > 0) & (gamma != 0)
    same = (arg == arg_ref) | ~live
    assert same.float().mean() > 0.999
    yr = y.view(R, ns, N).gather(1, arg.long().unsqueeze(1)).squeeze(1)
    assert torch.equal(yr, yraw)
    # first maximum among exact ties (padding rows repeat row 0): never an index in the padded half unless row 0 lost
    assert int((arg[live] >= ns // 2).sum()) == 0
