"""Shared-MLP building blocks with the Group-Free-3D parameter naming.

Mirrors the part of GF3D/pointnet2/pytorch_utils.py (GF3D =
external_src/group_free_3D) that the SA/FP backbone instantiates: ``SharedMLP``
(:12-37) made of ``Conv2d`` units (:158-189) = 1x1 conv [+ BatchNorm2d wrapper]
[+ ReLU], so ``state_dict`` keys read ``layer{i}.conv.weight`` /
``layer{i}.bn.bn.{weight,bias,running_mean,...}`` exactly like the reference and
its checkpoints load strictly.  Initialisation follows the reference
(kaiming-normal conv weights, zero conv bias, BN weight 1 / bias 0).
Every container is an ``nn.Sequential`` of plain torch layers, which is what
``pointnet2_ops.pointnet2_modules.shared_mlp_rows`` walks on the rows fast path.
"""
from typing import List

import torch.nn as nn


class BatchNorm2d(nn.Sequential):
    """`bn` child holding the real nn.BatchNorm2d (keys: ``bn.weight`` ...)."""

    def __init__(self, channels: int, name: str = ""):
        super().__init__()
        self.add_module(name + "bn", nn.BatchNorm2d(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class Conv2d(nn.Sequential):
    """1x1 convolution unit: conv -> [bn] -> [activation] (pre-activation order when `preact`)."""

    def __init__(self, in_size: int, out_size: int, *, kernel_size=(1, 1), stride=(1, 1),
                 padding=(0, 0), activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = ""):
        super().__init__()
        conv = nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        norm = BatchNorm2d(in_size if preact else out_size) if bn else None
        tail = [("bn", norm), ("activation", activation)]
        if preact:
            for key, mod in tail:
                if mod is not None:
                    self.add_module(name + key, mod)
        self.add_module(name + "conv", conv)
        if not preact:
            for key, mod in tail:
                if mod is not None:
                    self.add_module(name + key, mod)


class SharedMLP(nn.Sequential):
    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True),
                 preact: bool = False, first: bool = False, name: str = ""):
        super().__init__()
        for i, (c_in, c_out) in enumerate(zip(args[:-1], args[1:])):
            plain = first and preact and i == 0      # first pre-activated unit: bare conv
            self.add_module(name + f"layer{i}",
                            Conv2d(c_in, c_out, bn=bn and not plain,
                                   activation=None if plain else activation, preact=preact))
