"""Eval-mode set abstraction: one scale of an SA level as ONE kernel (csrc/x3_chain.hip, `pn2_sa_eval_x3`).

In eval mode every BatchNorm2d of a shared MLP (OPS/pointnet2_modules.py:9-19) is an affine map with constant running
statistics: it folds into its Conv2d as ``W' = diag(gamma / sqrt(var + eps)) W``, ``b' = beta - gamma mean / sqrt(var + eps)``
and the level (:58-70: group -> MLP -> max over nsample) no longer depends on the batch.  The kernel gathers the
neighbourhood rows, runs the folded layers on the matrix cores with the activations held in registers and writes only the
pooled ``(B, npoint, C_out)`` rows — no grouped tensor, no per-layer ``(B npoint nsample, C)`` tensor.

Arithmetic: the split-bf16 product ("f32x3", csrc/x3_common.h) — fp32-grade error; the per-level parity tests hold the same
1e-4 against the oracle as the exact fp32 kernels (tests/test_gpu_round6.py).  Inference only: the route is taken when no
gradient is being recorded; ``PN2_EVAL_FUSED=0`` (or ``set_eval_fused(False)``) restores the layer-by-layer kernels.
"""
import os
import weakref
from typing import Optional

import torch
import torch.nn as nn

from pointnet2_ops import pointnet2_utils as _pu

_ENABLED = os.environ.get("PN2_EVAL_FUSED", "1") != "0"


def set_eval_fused(enabled: bool) -> bool:
    global _ENABLED
    prev, _ENABLED = _ENABLED, bool(enabled)
    return prev


def eval_fused_enabled() -> bool:
    return _ENABLED


def _ext():
    return _pu._ext


def fold_batchnorm(conv: nn.Conv2d, bn: Optional[nn.modules.batchnorm._BatchNorm]):
    """(W' (out, in), b' (out)) fp32 of Conv1x1 followed by an eval-mode BatchNorm (folded in float64, rounded once)."""
    W = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).double()
    b = conv.bias.detach().double() if conv.bias is not None else torch.zeros(conv.out_channels, dtype=torch.float64, device=W.device)
    if bn is not None:
        s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        W = W * s[:, None]
        b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return W.float().contiguous(), b.float().contiguous()


class _Plan:
    """Packed weights of one (shared MLP, grouper) pair; rebuilt when a parameter / buffer changes (version counters)."""
    __slots__ = ("key", "mode", "c1", "c_mid", "c_out", "w0", "wstream", "bias_mid", "bias_fin", "Wx", "Wf", "b0")


_PLANS = {}      # id(mlp) -> (weakref, plan)


def _state_key(layers, grouper, C, device):
    parts = [str(device), int(C), float(grouper.radius), bool(grouper.normalize_xyz)]
    for conv, bn in layers:
        for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var):
            parts += [t.data_ptr(), t._version]
    return tuple(parts)


def _shape_of(layers, C, use_xyz):
    """(mode, c1, c_mid, widths ok) of a parsed stack fed by [xyz | C feature columns]."""
    if not use_xyz or len(layers) not in (2, 3):
        return None
    widths = [conv.out_channels for conv, _ in layers]
    if layers[0][0].in_channels != 3 + C:
        return None
    mode = 0 if C <= 12 else 1
    c1 = widths[0]
    c_mid = widths[1] if len(layers) == 3 else 0
    return mode, c1, c_mid, widths[-1]


def plan_for(grouper, mlp, C, nsample, device) -> Optional[_Plan]:
    from pointnet2_ops import fused_mlp
    e = _ext()
    if getattr(e, "sa_eval_x3", None) is None:
        return None
    layers = fused_mlp.parse_stack(mlp)
    if layers is None or any(bn.running_mean is None or bn.training for _, bn in layers):
        return None
    shape = _shape_of(layers, C, grouper.use_xyz)
    if shape is None:
        return None
    mode, c1, c_mid, c_out = shape
    if not e.sa_eval_x3_supported(mode, nsample, C if mode == 0 else 0, c1, c_mid, c_out):
        return None
    if mode == 1 and not (getattr(e, "lift_points", None) and e.group_lift_supported(c1)):
        return None
    key = _state_key(layers, grouper, C, device)
    hit = _PLANS.get(id(mlp))
    if hit is not None and hit[0]() is mlp and hit[1].key == key:
        return hit[1]
    with torch.no_grad():
        plan = _Plan()
        plan.key, plan.mode, plan.c1, plan.c_mid, plan.c_out = key, mode, c1, c_mid, c_out
        folded = [fold_batchnorm(conv, bn) for conv, bn in layers]
        W0, b0 = folded[0]
        plan.w0 = plan.Wx = plan.Wf = plan.b0 = None
        if mode == 0:
            # (c1, 16): [W'_x (/ radius) | W'_f | b' | 0]: the bias rides in the padding column 3 + C
            M0 = torch.zeros(c1, 16, dtype=torch.float32, device=device)
            M0[:, :3 + C] = W0
            if grouper.normalize_xyz:
                M0[:, :3] /= float(grouper.radius)
            M0[:, 3 + C] = b0
            plan.w0 = e.x3_pack_weight(M0, perm=False)
        else:
            plan.Wx, plan.Wf, plan.b0 = W0[:, :3].contiguous(), W0[:, 3:].contiguous(), b0
        rest = folded[1:]
        sizes = [e.x3_weight_bytes(*W.shape) for W, _ in rest]
        stream = torch.zeros(sum(sizes), dtype=torch.uint8, device=device)
        off = 0
        for i, ((W, _), n) in enumerate(zip(rest, sizes)):
            # the layer reads the previous layer's accumulators (permuted contraction order) unless it is the first matrix
            # layer behind a mode-1 gather
            perm = not (mode == 1 and i == 0)
            e.x3_pack_weight(W, perm=perm, out=stream[off:off + n])
            off += n
        plan.wstream = stream
        plan.bias_mid = rest[0][1] if c_mid else None
        plan.bias_fin = rest[-1][1]
    _PLANS[id(mlp)] = (weakref.ref(mlp, lambda _r, k=id(mlp): _PLANS.pop(k, None)), plan)
    return plan


def applicable(grouper, mlp, xyz, new_xyz, feats_rows) -> Optional[_Plan]:
    """The plan if this scale can take the one-kernel eval route right now, else None."""
    if not _ENABLED or torch.is_grad_enabled() and (
            (feats_rows is not None and feats_rows.requires_grad) or any(p.requires_grad for p in mlp.parameters())):
        return None
    # (ret_grouped_xyz only matters to the literal forward(): the rows path returns pooled features either way)
    if not isinstance(grouper, _pu.QueryAndGroup) or new_xyz is None:
        return None
    if not xyz.is_cuda or xyz.dtype != torch.float32 or (feats_rows is not None and feats_rows.dtype != torch.float32):
        return None
    C = 0 if feats_rows is None else feats_rows.size(2)
    return plan_for(grouper, mlp, C, grouper.nsample, xyz.device)


def sa_scale_eval(plan: _Plan, grouper, xyz, new_xyz, feats_rows, idx, out=None, col0=0) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,m,3), feats_rows (B,N,C)|None, idx (B,m,ns) -> (B, m, c_out) [or columns of `out`]."""
    e = _ext()
    B, m, _ns = idx.shape
    with torch.no_grad():
        if out is None:
            out = torch.empty(B, m, plan.c_out, dtype=torch.float32, device=xyz.device)
        if plan.mode == 0:
            feats = None if feats_rows is None else feats_rows.contiguous()
            e.sa_eval_x3(0, xyz, new_xyz, idx, feats, None, plan.c1, plan.w0, plan.c_mid, plan.wstream, plan.bias_mid,
                         plan.bias_fin, out, col0)
        else:
            N, C = feats_rows.size(1), feats_rows.size(2)
            P = e.mlp_gemm(feats_rows.contiguous().view(B * N, C), plan.Wf, pro=e.PRO_NONE, epi=e.EPI_NONE).view(B, N, -1)
            Pq, Q = e.lift_points(P, xyz, new_xyz, plan.Wx, grouper.normalize_xyz, grouper.radius)
            Q.sub_(plan.b0)                      # first activation = relu(Pq[idx] - Q[centre])
            e.sa_eval_x3(1, xyz, new_xyz, idx, Pq, Q, plan.c1, None, plan.c_mid, plan.wstream, plan.bias_mid, plan.bias_fin,
                         out, col0)
    return out
