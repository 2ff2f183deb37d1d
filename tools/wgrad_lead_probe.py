import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext
from microbench import timeit
dev = torch.device("cuda:0")
for M, N, K in [(1048576, 128, 131), (262144, 128, 259), (131072, 128, 259), (1048576, 128, 128)]:
    y = torch.randn(M, N, device=dev); g = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    consts = torch.rand(3, N, device=dev)
    res = {}
    for mode in ("lead", "nolead"):
        if mode == "nolead": os.environ["PN2_WGRAD_NOLEAD"] = "1"
        else: os.environ.pop("PN2_WGRAD_NOLEAD", None)
        res[mode] = round(timeit(lambda: _ext.mlp_wgrad(y, consts, x, _ext.PRO_GY, _ext.PRO_NONE, G=g), iters=10) * 1e6)
        res[mode + "_dW"] = _ext.mlp_wgrad(y, consts, x, _ext.PRO_GY, _ext.PRO_NONE, G=g)
    err = float((res["lead_dW"] - res["nolead_dW"]).abs().max() / res["nolead_dW"].abs().max())
    print((M, N, K), {k: v for k, v in res.items() if not k.endswith("_dW")}, "rel diff", err)
