"""Fused shared-MLP (+ neighbourhood max) on the fp32 MFMA kernels of libpn2_hip.so.

Runs a ``build_shared_mlp`` stack — [Conv2d 1x1 (bias=False), BatchNorm2d, ReLU] x L
(OPS/pointnet2_modules.py:9-19) — on point-major rows, optionally followed by the max
over each group of ``ns`` consecutive rows (``F.max_pool2d``, :67-70), with the
BatchNorm statistics, normalisation, ReLU and their backward folded into the GEMM
prologues / epilogues (csrc/mlp_gemm.hip).  Per layer the forward is one kernel
(+ a 1-block finalisation), the backward two (wgrad, dgrad); only the raw pre-BN
outputs ``y_l`` are ever written to HBM.

Semantics are those of the torch modules the parameters live in: training mode uses
batch statistics (biased variance) and updates ``running_mean`` / ``running_var``
(unbiased) / ``num_batches_tracked`` with the module's momentum; eval mode uses the
running statistics.  Gradients are produced for the input rows, the conv weights and
the BatchNorm affine parameters.
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from pointnet2_ops import pointnet2_utils as _pu


def _ext():
    return _pu._ext


#: hidden layers with 32 < N, K <= 128 run dgrad and wgrad as one kernel (csrc/mlp_bwd_fused.hip)
FUSED_BACKWARD = True

#: arithmetic of the shared-MLP stacks.  float32 = exact fp32 MFMA (the parity path, default).  bfloat16 = the MI355X
#: counterpart of the reference's 16-bit AMP training (scene_graph_prediction/main.py:64 `precision=16`): activations
#: between the layers stored as bf16, bf16 MFMA with fp32 accumulation, fp32 weights / BatchNorm statistics / weight
#: gradients; the geometry and grouping kernels stay fp32 like the reference's `custom_fwd(cast_inputs=float32)`
#: (OPS/pointnet2_utils.py:198).
_MLP_DTYPE = torch.float32


def set_mlp_dtype(dtype) -> torch.dtype:
    """torch.float32 | torch.bfloat16 (or "f32" / "bf16"); returns the previous setting."""
    global _MLP_DTYPE
    dtype = {"f32": torch.float32, "fp32": torch.float32, "bf16": torch.bfloat16}.get(dtype, dtype)
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("shared-MLP dtype must be torch.float32 or torch.bfloat16")
    prev, _MLP_DTYPE = _MLP_DTYPE, dtype
    return prev


def mlp_dtype() -> torch.dtype:
    return _MLP_DTYPE


def _bf16_ok(layers, ns) -> bool:
    """Shapes the bf16 kernels cover (csrc/mlp_bf16.hip): output widths that are multiples of 8 (16-byte bf16 row
    groups) up to 320, any input width."""
    return (all(conv.out_channels % 8 == 0 and conv.out_channels <= 320 and conv.in_channels <= 4096 for conv, _ in layers)
            and (not ns or layers[-1][0].out_channels % 2 == 0))


def parse_stack(mlp: nn.Module) -> Optional[List[Tuple[nn.Conv2d, nn.modules.batchnorm._BatchNorm]]]:
    """Flatten `mlp` into [(conv1x1, bn), ...] if it is exactly (conv, bn, relu)*; else None."""
    flat = []

    def walk(m):
        if isinstance(m, nn.Sequential):
            for c in m:
                walk(c)
        else:
            flat.append(m)

    walk(mlp)
    if not flat or len(flat) % 3:
        return None
    layers = []
    for i in range(0, len(flat), 3):
        conv, bn, act = flat[i:i + 3]
        ok = (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is None
              and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
              and isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.affine
              and bn.num_features == conv.out_channels and isinstance(act, nn.ReLU))
        if not ok:
            return None
        layers.append((conv, bn))
    return layers


def supported(mlp: nn.Module, x: torch.Tensor, ns: int = 0) -> bool:
    e = _ext()
    if not getattr(e, "HAS_FUSED_MLP", False) or not x.is_cuda or x.dtype != torch.float32:
        return False
    if ns and ns < 16:          # the pooled-gradient patch of the backward kernels is sized for ns >= 16
        return False
    layers = parse_stack(mlp)
    if layers is None:
        return False
    # limits of pn2_mlp_gemm / pn2_mlp_wgrad (csrc/mlp_gemm.hip): N <= 320 output and K <= 2048 input channels
    return all(conv.out_channels <= 320 and conv.in_channels <= 2048 for conv, _ in layers)


class _FusedMLP(Function):
    @staticmethod
    def forward(ctx, x, ns, layers, group, *params):
        """x (M,K0) rows; ns > 0 pools groups of ns rows; layers [(conv, bn)];
        params = (W_1, gamma_1, beta_1, W_2, ...) only so autograd tracks them.

        With `group` = (xyz, new_xyz, idx, use_xyz, normalize, radius) the rows are produced HERE
        from the point-major features `x` (B,N,C) by the fused gather kernel (QueryAndGroup tail),
        and the backward returns the feature gradient directly: the first layer's input gradient
        is only computed for the C feature columns (the relative-xyz columns never need one) and
        scattered back through the neighbourhood indices in the same autograd node."""
        e = _ext()
        ctx.group = group
        if group is not None:
            xyz, new_xyz, idx, use_xyz, normalize, radius = group
            ctx.feat_shape = None if x is None else tuple(x.shape)
            x = e.group_concat_rows(xyz, new_xyz, None if x is None else x.contiguous(), idx, use_xyz, normalize,
                                    radius).view(-1, (3 if use_xyz else 0) + (0 if x is None else x.size(2)))
        x = x.contiguous()
        M = x.size(0)
        L = len(layers)
        ys, fins, batch_flags = [], [], []
        stat_bufs = e.zero_arena(x.device, [((2, conv.out_channels), torch.float64) for conv, _ in layers])
        cur = x
        for l, (conv, bn) in enumerate(layers):
            W = params[3 * l].view(conv.out_channels, conv.in_channels)
            gamma, beta = params[3 * l + 1], params[3 * l + 2]
            use_batch = bn.training or bn.running_mean is None
            pro = e.PRO_NONE if l == 0 else e.PRO_BNRELU
            p = None if l == 0 else (fins[-1][2], fins[-1][3])
            if use_batch:
                stats = stat_bufs[l]
                y = e.mlp_gemm(cur, W, pro=pro, epi=e.EPI_STATS, p=p, stats=stats)
                momentum = 0.0
                rm = rv = nbt = None
                if bn.training and bn.track_running_stats and bn.running_mean is not None:
                    rm, rv = bn.running_mean, bn.running_var
                    if bn.momentum is not None:
                        momentum, nbt = bn.momentum, bn.num_batches_tracked       # counter bumped by the finalize kernel
                    else:                                                       # cumulative average: needs the count on the host
                        if bn.num_batches_tracked is not None:
                            bn.num_batches_tracked.add_(1)
                        momentum = 1.0 / float(bn.num_batches_tracked)
                fin = e.bn_finalize(stats, M, gamma, beta, bn.eps, momentum, rm, rv, nbt)
            else:
                y = e.mlp_gemm(cur, W, pro=pro, epi=e.EPI_NONE, p=p)
                rstd = torch.rsqrt(bn.running_var + bn.eps)
                scale = gamma * rstd
                fin = torch.stack([bn.running_mean, rstd, scale, beta - bn.running_mean * scale]).contiguous()
            ys.append(y)
            fins.append(fin)
            batch_flags.append(use_batch)
            cur = y
        yraw = None
        if ns:
            out, arg, yraw = e.bn_relu_rows_max(ys[-1], fins[-1], ns)
        else:
            out, arg = e.bn_relu_apply(ys[-1], fins[-1]), None
        ctx.ns, ctx.L, ctx.batch_flags = ns, L, batch_flags
        ctx.shapes = [params[3 * l].shape for l in range(L)]
        saved = [x] + ys + fins + [params[3 * l] for l in range(L)] + [params[3 * l + 1] for l in range(L)]
        if ns:
            saved += [out, arg, yraw]
            ctx.mark_non_differentiable(arg)
        ctx.save_for_backward(*saved)
        return (out, arg) if ns else out

    @staticmethod
    def backward(ctx, g_out, *unused):
        e = _ext()
        L, ns = ctx.L, ctx.ns
        saved = ctx.saved_tensors
        x = saved[0]
        ys = saved[1:1 + L]
        fins = saved[1 + L:1 + 2 * L]
        Ws = [w.view(w.size(0), w.size(1)) for w in saved[1 + 2 * L:1 + 3 * L]]
        gammas = saved[1 + 3 * L:1 + 4 * L]
        M = x.size(0)
        g_out = g_out.contiguous()

        # every zero-initialised accumulator of this backward from one allocation / one fill
        f64, f32 = torch.float64, torch.float32
        # first-layer fold (csrc/mlp_bwd_fused.hip): the layer above the first one reduces gz^T X instead of storing gz
        # when the first layer's input needs no gradient (raw coordinates / colours) and is at most 8 columns wide
        need_dgrad0 = ctx.needs_input_grad[0] and (ctx.group is None or ctx.feat_shape is not None)
        K0 = x.size(1)
        fold = (L >= 2 and FUSED_BACKWARD and not need_dgrad0 and ctx.batch_flags[0]
                and e.mlp_bwd_fused_fold_supported(Ws[1].size(0), Ws[1].size(1), K0))
        arena = e.zero_arena(x.device, [((2, Ws[-1].size(0)), f64)] + [((2, Ws[l].size(1)), f64) for l in range(L)] +
                             [(tuple(Ws[l].shape), f32) for l in range(L)] +
                             ([((Ws[0].size(0), K0), f32), ((K0 * K0 + K0,), f64)] if fold else []))
        sums0, sums_in, dWs = arena[0], arena[1:1 + L], arena[1 + L:1 + 2 * L]
        if ns:
            pooled, arg, yraw = saved[1 + 4 * L], saved[2 + 4 * L], saved[3 + 4 * L]
            gPm, sums = e.pool_bwd_prep(yraw, pooled, g_out, fins[-1], sums=sums0)
            gmode, G = e.PRO_POOLG, None
        else:
            G, sums = e.bn_relu_bwd_prep(ys[-1], g_out, fins[-1], sums=sums0)
            gmode, arg, gPm = e.PRO_GY, None, None

        grads = [None] * (3 * L)
        gx = None
        for l in range(L - 1, -1, -1):
            one_pass = (l <= 1 and fold) or (l > 0 and FUSED_BACKWARD and
                                             e.mlp_bwd_fused_supported(Ws[l].size(0), Ws[l].size(1)))
            Wt = None
            if not one_pass and (l > 0 or need_dgrad0):
                # the dgrad GEMM takes the weight as (K, N) rows (feature columns only at a grouped first layer):
                # transposed by the constants kernel
                k0 = 3 if (l == 0 and ctx.group is not None and ctx.group[3]) else 0
                consts, dgamma, dbeta, Wt = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l],
                                                            W=Ws[l].contiguous(), k0=k0)
            else:
                consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l])
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if l == 1 and fold:
                sums, dW, P1 = e.mlp_bwd_fused_fold(ys[1], consts, Ws[1].contiguous(), ys[0], fins[0], x, gmode, G=G, arg=arg,
                                                    gP=gPm, ns=ns, sums=sums_in[1], dW=dWs[1], P1=arena[1 + 2 * L])
                grads[3] = dW.view(ctx.shapes[1])
                continue
            if l == 0 and fold:
                gram = e.rows_gram(x, arena[2 + 2 * L])
                grads[0] = e.first_layer_dw(consts, P1, Ws[0].contiguous(), gram).view(ctx.shapes[0])
                continue
            if l > 0 and FUSED_BACKWARD and e.mlp_bwd_fused_supported(Ws[l].size(0), Ws[l].size(1)):
                # hidden layer: dgrad + wgrad from one read of (g, y_l, y_{l-1})
                G, sums, dW = e.mlp_bwd_fused(ys[l], consts, Ws[l].contiguous(), ys[l - 1], fins[l - 1], gmode,
                                              G=G, arg=arg, gP=gPm, ns=ns, sums=sums_in[l], dW=dWs[l])
                grads[3 * l] = dW.view(ctx.shapes[l])
                gmode, arg, gPm = e.PRO_GY, None, None
                continue
            act = x if l == 0 else ys[l - 1]
            dW = e.mlp_wgrad(ys[l], consts, act, gmode, e.PRO_NONE if l == 0 else e.PRO_BNRELU,
                             G=G, arg=arg, gP=gPm, ns=ns, a_fin=None if l == 0 else fins[l - 1], dW=dWs[l])
            grads[3 * l] = dW.view(ctx.shapes[l])
            need_dgrad = l > 0 or need_dgrad0
            if need_dgrad:
                # Wt (K_l [- 3], N_l): dgrad is out[M,K_l] = gy[M,N_l] @ Wt^T (feature columns only at a grouped first layer)
                p = (consts[0], consts[1], consts[2])
                if l > 0:
                    sums = sums_in[l]
                    Gn = e.mlp_gemm(G, Wt, pro=gmode, epi=e.EPI_MASK, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns,
                                    stats=sums, Yprev=ys[l - 1], e_fin=fins[l - 1], M=M)
                    G, gmode, arg, gPm = Gn, e.PRO_GY, None, None
                else:
                    gx = e.mlp_gemm(G, Wt, pro=gmode, epi=e.EPI_NONE, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns, M=M)
        if gx is not None and ctx.group is not None:
            idx = ctx.group[2]
            Bq, npoint, nsample = idx.shape
            Bf, Nf, Cf = ctx.feat_shape
            inv = e.inverse_index_of(idx, Nf)
            if inv is not None:          # prefetched inverse index: per-point sum, no atomics (csrc/group_csr.hip)
                gx = e.group_rows_grad_csr(gx.view(Bq, npoint, nsample, Cf), inv, Nf, Cf, 0)
            else:
                gx = e.group_rows_grad(gx.view(Bq, npoint, nsample, Cf), idx, Nf, Cf, 0)
        return (gx, None, None, None, *grads)


class _FusedMLPBf16(Function):
    """The same stack on the bf16 MFMA kernels: y_l and the activation gradients are bf16 tensors, everything that is
    reduced (BatchNorm sums, weight gradients) or small (pooled outputs, per-channel constants) stays fp32 / fp64."""

    @staticmethod
    def forward(ctx, x, ns, layers, group, *params):
        e = _ext()
        ctx.group = group
        if group is not None:
            xyz, new_xyz, idx, use_xyz, normalize, radius = group
            ctx.feat_shape = None if x is None else tuple(x.shape)
            k_in = (3 if use_xyz else 0) + (0 if x is None else x.size(2))
            x = e.group_concat_rows_bf16(xyz, new_xyz, None if x is None else x.contiguous(), idx, use_xyz, normalize, radius)
            x = x.view(-1, x.size(-1))                                      # (M, pad8(k_in)) bf16, zero pad columns
        else:
            x = x.contiguous()                                              # fp32 rows of any width
            k_in = x.size(1)
        M = x.size(0)
        L = len(layers)
        ys, fins, batch_flags = [], [], []
        stat_bufs = e.zero_arena(x.device, [((2, conv.out_channels), torch.float64) for conv, _ in layers])
        cur = x
        for l, (conv, bn) in enumerate(layers):
            W = params[3 * l].view(conv.out_channels, conv.in_channels)
            gamma, beta = params[3 * l + 1], params[3 * l + 2]
            use_batch = bn.training or bn.running_mean is None
            pro = e.PRO_NONE if l == 0 else e.PRO_BNRELU
            p = None if l == 0 else (fins[-1][2], fins[-1][3])
            if use_batch:
                stats = stat_bufs[l]
                y = e.mlp_gemm_bf16(cur, W, pro=pro, epi=e.EPI_STATS, p=p, stats=stats)
                momentum = 0.0
                rm = rv = nbt = None
                if bn.training and bn.track_running_stats and bn.running_mean is not None:
                    rm, rv = bn.running_mean, bn.running_var
                    if bn.momentum is not None:
                        momentum, nbt = bn.momentum, bn.num_batches_tracked
                    else:
                        if bn.num_batches_tracked is not None:
                            bn.num_batches_tracked.add_(1)
                        momentum = 1.0 / float(bn.num_batches_tracked)
                fin = e.bn_finalize(stats, M, gamma, beta, bn.eps, momentum, rm, rv, nbt)
            else:
                y = e.mlp_gemm_bf16(cur, W, pro=pro, epi=e.EPI_NONE, p=p)
                rstd = torch.rsqrt(bn.running_var + bn.eps)
                scale = gamma * rstd
                fin = torch.stack([bn.running_mean, rstd, scale, beta - bn.running_mean * scale]).contiguous()
            ys.append(y)
            fins.append(fin)
            batch_flags.append(use_batch)
            cur = y
        yraw = None
        if ns:
            out, arg, yraw = e.bn_relu_rows_max_bf16(ys[-1], fins[-1], ns)
        else:
            out, arg = e.bn_relu_apply_bf16(ys[-1], fins[-1]), None
        ctx.ns, ctx.L, ctx.batch_flags, ctx.k_in = ns, L, batch_flags, k_in
        ctx.shapes = [params[3 * l].shape for l in range(L)]
        saved = [x] + ys + fins + [params[3 * l] for l in range(L)] + [params[3 * l + 1] for l in range(L)]
        if ns:
            saved += [out, arg, yraw]
            ctx.mark_non_differentiable(arg)
        ctx.save_for_backward(*saved)
        return (out, arg) if ns else out

    @staticmethod
    def backward(ctx, g_out, *unused):
        e = _ext()
        L, ns = ctx.L, ctx.ns
        saved = ctx.saved_tensors
        x = saved[0]
        ys = saved[1:1 + L]
        fins = saved[1 + L:1 + 2 * L]
        Ws = [w.view(w.size(0), w.size(1)) for w in saved[1 + 2 * L:1 + 3 * L]]
        gammas = saved[1 + 3 * L:1 + 4 * L]
        M = x.size(0)
        g_out = g_out.contiguous()
        f64, f32 = torch.float64, torch.float32
        need_dgrad0 = ctx.needs_input_grad[0] and (ctx.group is None or ctx.feat_shape is not None)
        arena = e.zero_arena(x.device, [((2, Ws[-1].size(0)), f64)] + [((2, Ws[l].size(1)), f64) for l in range(L)] +
                             [(tuple(Ws[l].shape), f32) for l in range(L)])
        sums0, sums_in, dWs = arena[0], arena[1:1 + L], arena[1 + L:1 + 2 * L]
        if ns:
            pooled, arg, yraw = saved[1 + 4 * L], saved[2 + 4 * L], saved[3 + 4 * L]
            gPm, sums = e.pool_bwd_prep(yraw, pooled, g_out, fins[-1], sums=sums0)
            gmode, G = e.PRO_POOLG, None
        else:
            G, sums = e.bn_relu_bwd_prep_bf16(ys[-1], g_out, fins[-1], sums=sums0)
            gmode, arg, gPm = e.PRO_GY, None, None

        grads = [None] * (3 * L)
        gx = None
        for l in range(L - 1, -1, -1):
            need_dgrad = l > 0 or need_dgrad0
            if need_dgrad:
                k0 = 3 if (l == 0 and ctx.group is not None and ctx.group[3]) else 0
                consts, dgamma, dbeta, Wt = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l],
                                                            W=Ws[l].contiguous(), k0=k0)
            else:
                consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l])
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if l > 0 and FUSED_BACKWARD and e.mlp_bwd_bf16_supported(Ws[l].size(0), Ws[l].size(1)):
                # hidden layer: dgrad + wgrad from one read of (g, y_l, y_{l-1})
                G, sums, dW = e.mlp_bwd_bf16(ys[l], consts, Wt, ys[l - 1], fins[l - 1], gmode, G=G, arg=arg, gP=gPm, ns=ns,
                                             sums=sums_in[l], dW=dWs[l])
                grads[3 * l] = dW.view(ctx.shapes[l])
                gmode, arg, gPm = e.PRO_GY, None, None
                continue
            act = x if l == 0 else ys[l - 1]
            dW = e.mlp_wgrad_bf16(ys[l], consts, act, gmode, e.PRO_NONE if l == 0 else e.PRO_BNRELU, Ws[l].size(1),
                                  G=G, arg=arg, gP=gPm, ns=ns, a_fin=None if l == 0 else fins[l - 1], dW=dWs[l])
            grads[3 * l] = dW.view(ctx.shapes[l])
            if need_dgrad:
                p = (consts[0], consts[1], consts[2])
                if l > 0:
                    sums = sums_in[l]
                    G = e.mlp_gemm_bf16(G, Wt, pro=gmode, epi=e.EPI_MASK, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns,
                                        stats=sums, Yprev=ys[l - 1], e_fin=fins[l - 1], M=M)
                    gmode, arg, gPm = e.PRO_GY, None, None
                else:
                    # gradient rows of a grouped first layer stay bf16 (the scatter / per-point sum accumulates in fp32);
                    # a plain row input gets its fp32 gradient directly
                    rows_bf16 = ctx.group is not None and Wt.size(0) % 4 == 0
                    gx = e.mlp_gemm_bf16(G, Wt, pro=gmode, epi=e.EPI_NONE, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns, M=M,
                                         out_f32=not rows_bf16)
        if gx is not None and ctx.group is not None:
            idx = ctx.group[2]
            Bq, npoint, nsample = idx.shape
            Bf, Nf, Cf = ctx.feat_shape
            inv = e.inverse_index_of(idx, Nf)
            if inv is not None:          # prefetched inverse index: per-point sum, no atomics (csrc/group_csr.hip)
                gx = e.group_rows_grad_csr(gx.view(Bq, npoint, nsample, Cf), inv, Nf, Cf, 0)
            else:
                gx = e.group_rows_grad(gx.view(Bq, npoint, nsample, Cf), idx, Nf, Cf, 0)
        return (gx, None, None, None, *grads)


def _node(layers, ns):
    """The autograd node for the current arithmetic (set_mlp_dtype) that covers this stack."""
    if _MLP_DTYPE == torch.bfloat16 and getattr(_ext(), "HAS_BF16_MLP", False) and _bf16_ok(layers, ns):
        return _FusedMLPBf16
    return _FusedMLP


def _params(layers):
    params = []
    for conv, bn in layers:
        params += [conv.weight, bn.weight, bn.bias]
    return params


def fused_shared_mlp(mlp: nn.Module, x: torch.Tensor, ns: int = 0) -> torch.Tensor:
    """x (M, C_in) rows -> (M, C_out) [ns == 0] or (M // ns, C_out) max-pooled over groups of ns rows."""
    layers = parse_stack(mlp)
    assert layers is not None, "fused_shared_mlp: unsupported stack (call supported() first)"
    res = _node(layers, ns).apply(x, int(ns), layers, None, *_params(layers))
    return res[0] if ns else res


def fused_group_mlp_pool(mlp: nn.Module, xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius) -> torch.Tensor:
    """Ball-query neighbourhoods -> shared MLP -> max, one autograd node:
    xyz (B,N,3), new_xyz (B,m,3), feats_rows (B,N,C)|None, idx (B,m,ns) -> (B, m, C_out)."""
    layers = parse_stack(mlp)
    assert layers is not None
    B, m, ns = idx.shape
    group = (xyz, new_xyz, idx, bool(use_xyz), bool(normalize), radius)
    res = _node(layers, ns).apply(feats_rows, int(ns), layers, group, *_params(layers))
    return res[0].view(B, m, -1)
