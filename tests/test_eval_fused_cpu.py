"""Host logic of the one-kernel eval level (pointnet2_ops/eval_fused.py) without a GPU: the BatchNorm fold, the stack
classification (mode / widths) and the kmap of the weight fragments against the accumulator layout it has to match."""
import torch
import torch.nn as nn

from pointnet2_ops import eval_fused, fused_mlp
from pointnet2_ops import pointnet2_modules as pm


def test_fold_batchnorm_reproduces_conv_bn_in_eval_mode():
    torch.manual_seed(0)
    mlp = pm.build_shared_mlp([7, 16, 24]).eval()
    for mod in mlp.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.4, 1.6)
            mod.weight.data.normal_()
            mod.bias.data.normal_(0, 0.2)
    x = torch.randn(5, 7, 11, 3)
    want = mlp(x)
    layers = fused_mlp.parse_stack(mlp)
    h = x.permute(0, 2, 3, 1).reshape(-1, 7)
    for conv, bn in layers:
        W, b = eval_fused.fold_batchnorm(conv, bn)
        h = torch.relu(h @ W.t() + b)
    got = h.view(5, 11, 3, 24).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, want, atol=1e-5, rtol=1e-5)


def test_stack_classification():
    two = fused_mlp.parse_stack(pm.build_shared_mlp([6, 64, 128]))
    three = fused_mlp.parse_stack(pm.build_shared_mlp([131, 128, 128, 256]))
    assert eval_fused._shape_of(two, 3, True) == (0, 64, 0, 128)          # <= 12 feature columns: gathered first layer
    assert eval_fused._shape_of(three, 128, True) == (1, 128, 128, 256)   # lifted first layer
    assert eval_fused._shape_of(three, 128, False) is None                # use_xyz = False is not covered
    assert eval_fused._shape_of(two, 4, True) is None                     # the stack does not belong to these inputs


def test_permuted_contraction_order_matches_the_accumulator_layout():
    """x3_common.h: a transposed 32 x 32 accumulator tile holds, in lane half h, register r, channel (r & 3) + 8 (r >> 2) + 4 h;
    registers 8 q .. 8 q + 7 are the operand fragment of chunk q: kmap_perm(c, h, i) must name exactly those channels."""
    def kmap(c, h, i, perm):
        return 16 * c + ((4 * h + i) if i < 4 else (8 + 4 * h + i - 4)) if perm else 16 * c + 8 * h + i
    for T in range(2):
        for h in range(2):
            for q in range(2):
                held = [32 * T + (r & 3) + 8 * (r >> 2) + 4 * h for r in range(8 * q, 8 * q + 8)]
                assert held == [kmap(2 * T + q, h, i, True) for i in range(8)]
    # both orders are permutations of a chunk's 16 indices
    for perm in (False, True):
        assert sorted(kmap(3, h, i, perm) for h in range(2) for i in range(8)) == list(range(48, 64))
