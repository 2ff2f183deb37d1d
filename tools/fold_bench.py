"""First layer without its output tensor: the two kernels that recompute it against the ones that read it, SA1 shape.

  pn2_mlp_gemm (K0 -> 64, statistics) + pn2_mlp_gemm (BN+ReLU prologue)   vs   pn2_mlp_gemm_first
  pn2_mlp_bwd_fused_fold (reads y_0)                                      vs   pn2_mlp_bwd_fused_fold_first
ms per call from events (10 calls after 2 warm-ups)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/4d-or_amd")
from pointnet2_ops import _ext as e  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    M, K0, N0, N1 = 32 * 2048 * 64, 6, 64, 64
    g = torch.Generator().manual_seed(0)
    X0 = (torch.randn(M, K0, generator=g) + 0.3).cuda()
    W0 = (torch.randn(N0, K0, generator=g) * 0.5).cuda()
    W1 = (torch.randn(N1, N0, generator=g) * 0.2).cuda()
    gamma, beta = (torch.rand(N0, generator=g) + 0.5).cuda(), (torch.randn(N0, generator=g) * 0.2).cuda()
    st0 = torch.zeros(2, N0, dtype=torch.float64, device="cuda")
    y0 = e.mlp_gemm(X0, W0, pro=e.PRO_NONE, epi=e.EPI_STATS, stats=st0)
    fin0 = e.bn_finalize(st0, M, gamma, beta, 1e-5, 0.0, None, None)
    st1 = torch.zeros(2, N1, dtype=torch.float64, device="cuda")
    y1 = e.mlp_gemm(y0, W1, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=(fin0[2], fin0[3]), stats=st1)
    consts = (torch.randn(3, N1, generator=g) * 0.3).cuda().contiguous()
    G = torch.randn(M, N1, device="cuda")
    print("gemm K0->64 + stats      %.3f ms" % timed(lambda: e.mlp_gemm(X0, W0, pro=e.PRO_NONE, epi=e.EPI_STATS, stats=st0)))
    print("gemm 64->64 bnrelu       %.3f ms" % timed(lambda: e.mlp_gemm(y0, W1, pro=e.PRO_BNRELU, epi=e.EPI_STATS, p=(fin0[2], fin0[3]), stats=st1)))
    print("gemm_first               %.3f ms" % timed(lambda: e.mlp_gemm_first(X0, W0, fin0, W1, epi=e.EPI_STATS, stats=st1)))
    s, dW, P1 = (torch.zeros(2, N0, dtype=torch.float64, device="cuda"), torch.zeros(N1, N0, device="cuda"),
                 torch.zeros(N0, K0, device="cuda"))
    print("fold (reads y_0)         %.3f ms" % timed(lambda: e.mlp_bwd_fused_fold(y1, consts, W1, y0, fin0, X0, e.PRO_GY, G=G, sums=s, dW=dW, P1=P1)))
    print("fold_first (recomputes)  %.3f ms" % timed(lambda: e.mlp_bwd_fused_fold_first(y1, consts, W1, W0, fin0, X0, e.PRO_GY, G=G, sums=s, dW=dW, P1=P1)))


if __name__ == "__main__":
    main()
