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
import collections
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from pointnet2_ops import pointnet2_utils as _pu


def _ext():
    return _pu._ext


#: hidden layers with 32 < N, K <= 128 run dgrad and wgrad as one kernel (csrc/mlp_bwd_fused.hip)
FUSED_BACKWARD = True

#: the max-pooled last layer of a stack never materialises its (M, C_out) output: the forward GEMM reduces the group
#: maxima in its epilogue (pn2_mlp_gemm_pool), the backward runs in Gram form from y_{L-1} alone (pn2_pool_bwd)
POOL_FUSED = True
#: (measurement switch, tools only: PN2_POOL_FUSED=0 python bench.py ... runs the materialised path for an A/B)
import os as _os
if _os.environ.get("PN2_POOL_FUSED") == "0":
    POOL_FUSED = False

#: a first layer with <= 8 input columns (grouped xyz / colour rows) never materialises its output either: its batch
#: statistics come from the 8 x 8 Gram matrix of the rows, the second layer recomputes it while staging its A tiles
#: (pn2_mlp_gemm_first) and so does the backward (pn2_mlp_bwd_fused_fold_first)
FIRST_FREE = True
#: bf16 path: first-layer fold of the backward (pn2_mlp_bwd_bf16_fold); PN2_BF16_FOLD=0 switches it off for an A/B
BF16_FOLD = _os.environ.get("PN2_BF16_FOLD") != "0"
if _os.environ.get("PN2_FIRST_FREE") == "0":
    FIRST_FREE = False

#: the first layer of a grouped stack whose input is [relative xyz | C >= 16 feature channels] is applied BEFORE the grouping
#: (csrc/group_lift.hip): W [rel | f[idx]] = Wx rel + (Wf f)[idx] — the feature product once per point instead of once per
#: (centre, sample) row, no grouped tensor, no M-row dgrad / wgrad / scatter in the backward.  fp32 node, ungrouped input,
#: and (when gradients are needed) an inverse neighbourhood index next to idx.  PN2_LIFT_FIRST=0 switches it off (A/B).
LIFT_FIRST = _os.environ.get("PN2_LIFT_FIRST") != "0"
#: ... also for SPARSE balls (the scene-graph encoders' second level: most slots of a neighbourhood repeat its first hit, a
#: few points own hundreds of rows — the backward walks those with sixteen waves each), the inverse index built in the step:
#: 8 scans per step, whole-batch statistics, fp32: 169 -> 242 scans/s.  PN2_LIFT_SPARSE=0 restores the grouped route (A/B).
LIFT_SPARSE = _os.environ.get("PN2_LIFT_SPARSE") != "0"
#: ... and on the bf16 node (same kernels with bf16 rows: y0 and the gradient rows in bf16, per-point products and every sum
#: in fp32).  PN2_BF16_LIFT=0 restores the grouped bf16 route (A/B).
BF16_LIFT = _os.environ.get("PN2_BF16_LIFT") != "0"
#: the lifted layer's OUTPUT is not stored either (fp32 node, training mode, stacks of three layers or more): y0[row] =
#: Pq[point] - Q[centre] is re-formed by the layer above — its GEMM (pn2_mlp_gemm_lift), its weight gradient
#: (pn2_mlp_wgrad_lift) and the mask / BatchNorm sums of its input gradient (pn2_mlp_dgrad_lift); the rows of Pq (a cloud's
#: share: 1 MB at the headline's SA2) come out of L2.  OFF by default: it removes 2.1 GB (SA2) of HBM traffic per headline step and
#: is 0.11 ms faster kernel by kernel (tools/lift_free_bench.py), but the step does not get shorter (three same-box A/B pairs,
#: tools/ab_lift_free.sh: 10.78-10.79 ms stored, 10.81-10.82 re-formed) — the kernels involved sit at their instruction-issue
#: limit, not at HBM's, the stored tensor's tail is still in the 256 MB MALL when the next kernel reads it, and the gathering
#: variants hold more registers next to the resident sampling workgroups.  PN2_LIFT_FREE=1 switches it on.
LIFT_FREE = _os.environ.get("PN2_LIFT_FREE") == "1"
#: ... from this many grouped rows on (tools/lift_free_bench.py, profiles/r05_lift_free.jsonl: 1M rows 1.34 -> 1.24 ms for the
#: four kernels involved, 262k rows 0.400 -> 0.398, 131k rows 0.228 -> 0.241: below, the two extra launches cost more than
#: the tensor)
LIFT_FREE_MIN_ROWS = int(_os.environ.get("PN2_LIFT_FREE_MIN_ROWS", str(1 << 17 | 1 << 16)))
#: bf16 node: the max-pooled last layer never stores its (M, C_out) output either (pn2_mlp_gemm_pool_bf16: maxima of the fp32
#: accumulators in the GEMM's epilogue; the backward re-forms y_L from y_{L-1} on the matrix pipe, pn2_mlp_bwd_bf16_pool) — stacks
#: of three layers or more whose last layer is at most 128 wide (the backbone's SA1).  PN2_BF16_POOL=0: stored route (A/B).
BF16_POOL = _os.environ.get("PN2_BF16_POOL") != "0"
#: ... nor does a first layer with <= 8 input columns (grouped xyz / colour rows): statistics from the Gram matrix of the rows,
#: the second layer re-forms it while staging its A tiles (pn2_mlp_gemm_first_bf16), the fold of the backward likewise
#: (pn2_mlp_bwd_bf16_fold_first).  PN2_BF16_FIRST=0: stored route (A/B).
BF16_FIRST = _os.environ.get("PN2_BF16_FIRST") != "0"
#: ... also inside segment-table stacks (per-scan statistics at the whole-batch launch count): the lifted layer is issued once
#: per scan (the launches of single-scan steps), the rest of the stack through the table.  PN2_BF16_LIFT_SEG=0: grouped route.
BF16_LIFT_SEG = _os.environ.get("PN2_BF16_LIFT_SEG") != "0"

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


def set_x3(enabled: bool) -> bool:
    """Opt-in arithmetic "f32x3" of the fp32 node: the shared-MLP GEMMs the split-bf16 product covers (K in {64, 128}: hidden
    layers, input gradients, pooled last layers of the SA stacks) run on the bf16 matrix cores as hi/mid/lo pieces with six
    partial products and fp32 accumulation (csrc/x3_common.h) — fp32-grade error (tests/test_gpu_round6.py), not the exact
    fp32 MFMA arithmetic of the default path.  Returns the previous setting.  (PN2_X3=1 switches it on at import.)"""
    e = _ext()
    prev, e.X3_GEMM = bool(getattr(e, "X3_GEMM", False)), bool(enabled)
    return prev


def _bf16_ok(layers, ns) -> bool:
    """Shapes the bf16 kernels cover (csrc/mlp_bf16.hip): output widths that are multiples of 8 (16-byte bf16 row
    groups) up to 320, any input width."""
    return (all(conv.out_channels % 8 == 0 and conv.out_channels <= 320 and conv.in_channels <= 4096 for conv, _ in layers)
            and (not ns or layers[-1][0].out_channels % 2 == 0))


_PARSED = {}        # id(module) -> (weak reference, structural fingerprint, parsed stack)


def _fingerprint(mlp: nn.Module):
    """Identity of every direct child slot, two levels deep (a stack is Sequential(conv, bn, relu, ...) or a Sequential of
    such blocks): replacing a layer — convert_sync_batchnorm, conv/bn folding, a swapped module — changes it."""
    fp = []
    for c in mlp._modules.values():
        fp.append(id(c))
        sub = getattr(c, "_modules", None)
        if sub:
            fp.extend(id(g) for g in sub.values())
    return tuple(fp)


def parse_stack(mlp: nn.Module) -> Optional[List[Tuple[nn.Conv2d, nn.modules.batchnorm._BatchNorm]]]:
    """Flatten `mlp` into [(conv1x1, bn), ...] if it is exactly (conv, bn, relu)*; else None.  Cached per module object (the
    walk is ~20 us of Python, paid 20 times per one-scan step of the scene-graph model) under a fingerprint of the
    container's children, so a stack whose layers were replaced after its first call is parsed again; entries of dead
    modules are purged by the weak reference's callback (an id can be reused)."""
    key = id(mlp)
    fp = _fingerprint(mlp)
    hit = _PARSED.get(key)
    if hit is not None and hit[0]() is mlp and hit[1] == fp:
        return hit[2]
    layers = _parse_stack(mlp)
    import weakref
    _PARSED[key] = (weakref.ref(mlp, lambda _r, key=key: _PARSED.pop(key, None)), fp, layers)
    return layers


def _parse_stack(mlp: nn.Module):
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
        L = len(layers)
        lift = False
        feats = None
        if group is not None:
            xyz, new_xyz, idx, use_xyz, normalize, radius = group[:6]
            ctx.feat_shape = None if x is None else tuple(x.shape)
            pre = getattr(ctx, "x_rows", None)          # a segmented call groups the whole batch once and hands in the rows
            if pre is None and len(group) > 8 and group[8] is not None:
                pre = group[8].view(-1, group[8].size(-1))     # rows the fused query + grouping kernel emitted next to idx
            # first layer before the grouping (LIFT_FIRST): the rows are never formed
            # ... whenever the backward has (or, for crowded balls, may build) the inverse neighbourhood index it sums
            # through: the route must not depend on whether the geometry was prefetched (identical results either way)
            has_inv = len(group) > 6 and group[6] is not None
            crowded = len(group) > 7 and bool(group[7])
            lift = bool(pre is None and _lift_eligible(e, layers, use_xyz, x)
                        and (not any(ctx.needs_input_grad) or has_inv or crowded or LIFT_SPARSE))
            ctx.lift_inv = None
            if lift and any(ctx.needs_input_grad) and not has_inv:
                ctx.lift_inv = tuple(e.group_inverse_index(idx, xyz.size(1)))      # (not prefetched: built here)
            if lift:
                feats = x.contiguous()
                M, K0 = idx.numel(), 3 + feats.size(2)
                x = None
            else:
                x = pre if pre is not None else e.group_concat_rows(
                    xyz, new_xyz, None if x is None else x.contiguous(), idx, use_xyz, normalize,
                    radius).view(-1, (3 if use_xyz else 0) + (0 if x is None else x.size(2)))
        if not lift:
            x = x.contiguous()
            M, K0 = x.size(0), x.size(1)
        ys, fins, batch_flags = [], [], []
        stat_bufs = e.zero_arena((feats if lift else x).device, [((2, conv.out_channels), torch.float64) for conv, _ in layers])
        cur = x
        # pooled last layer without its output tensor: needs the Gram-form backward whenever anything needs a gradient
        Kl, Nl = layers[-1][0].in_channels, layers[-1][0].out_channels
        needs_grad = any(ctx.needs_input_grad)
        # grouped input: only for CROWDED balls (group[7], see fused_group_mlp_pool) — with sparse balls most slots of a
        # neighbourhood repeat its first hit, the arg-max ties go to row 0 and nearly all of a group's columns land on one
        # or two rows: the row lists of pn2_pool_bwd overflow into their one-row-at-a-time path and the materialised
        # kernels are 2-4x faster (measured on the scene-graph encoders, profiles/r03_sgp8_pool_ab.md)
        want = True if group is None or len(group) < 8 or group[7] is None else bool(group[7])
        pool_fused = bool(POOL_FUSED and want and ns and L >= 2 and M % ns == 0 and getattr(e, "pool_layer_supported", None)
                          and e.pool_layer_supported(Kl, Nl, ns) and (not needs_grad or e.pool_bwd_supported(Nl, Kl, ns)))
        pooled_parts = None
        # first layer without its output tensor: when gradients are needed, only if the backward will take the fold path
        bn0 = layers[0][1]
        need_dgrad0 = ctx.needs_input_grad[0] and (group is None or ctx.feat_shape is not None)
        first_free = bool(
            FIRST_FREE and L >= 2 and K0 <= 8 and not (pool_fused and L == 2) and getattr(e, "mlp_gemm_first", None)
            and e.mlp_gemm_first_supported(K0, layers[0][0].out_channels, layers[1][0].out_channels)
            and (not needs_grad or ((bn0.training or bn0.running_mean is None) and FUSED_BACKWARD and not need_dgrad0
                                    and e.mlp_bwd_fused_fold_supported(layers[1][0].out_channels,
                                                                       layers[0][0].out_channels, K0))))
        # lifted first layer without its output: the layer above re-forms it (module comment at LIFT_FREE)
        lift_free = bool(
            lift and LIFT_FREE and L >= 3 and M >= LIFT_FREE_MIN_ROWS and not isinstance(ctx, _SegCtx)
            and getattr(e, "mlp_gemm_lift", None) is not None and feats.is_cuda
            and all((bn.training or bn.running_mean is None) for _c, bn in layers[:2])
            and M % idx.size(2) == 0 and e.mlp_lift_supported(layers[0][0].out_channels, layers[1][0].out_channels, idx.size(2)))
        ctx.lift_free = None
        gram = W0c = None
        if first_free:
            gram = e.rows_gram(x, e.zero_arena(x.device, [((K0 * K0 + K0,), torch.float64)])[0])
            W0c = params[0].view(layers[0][0].out_channels, K0).contiguous()
        for l, (conv, bn) in enumerate(layers):
            W = params[3 * l].view(conv.out_channels, conv.in_channels)
            gamma, beta = params[3 * l + 1], params[3 * l + 2]
            use_batch = bn.training or bn.running_mean is None
            pro = e.PRO_NONE if l == 0 else e.PRO_BNRELU
            p = None if l == 0 else (fins[-1][2], fins[-1][3])
            if pool_fused and l == L - 1:
                Wf, sgn = e.pool_flip_rows(W.contiguous(), gamma)
                pooled_parts = e.mlp_gemm_pool(cur, Wf, sgn, ns, p=p, stats=stat_bufs[l]) + (sgn,)
            if use_batch:
                stats = stat_bufs[l]
                if lift_free and l == 0:
                    y, ctx.lift_P, ctx.lift_free = _lift_forward_free(e, feats, W, group, stats)
                elif lift_free and l == 1:
                    y = e.mlp_gemm_lift(*ctx.lift_free, idx.size(2), fins[0], W.contiguous(), stats)
                elif lift and l == 0:
                    y, ctx.lift_P = _lift_forward(e, feats, W, group, stats)
                elif first_free and l == 0:
                    y = None
                    e.first_layer_stats(W.contiguous(), gram, stats)
                elif first_free and l == 1:
                    y = e.mlp_gemm_first(x, W0c, fins[0], W.contiguous(), epi=e.EPI_STATS, stats=stats)
                else:
                    y = None if pooled_parts is not None else e.mlp_gemm(cur, W, pro=pro, epi=e.EPI_STATS, p=p, stats=stats)
                momentum = 0.0
                rm = rv = nbt = None
                if (bn.training and bn.track_running_stats and bn.running_mean is not None
                        and not getattr(ctx, "defer_running", False)):     # (a segmented call updates them itself, in scan order)
                    rm, rv = bn.running_mean, bn.running_var
                    if bn.momentum is not None:
                        momentum, nbt = bn.momentum, bn.num_batches_tracked       # counter bumped by the finalize kernel
                    else:                                                       # cumulative average: needs the count on the host
                        if bn.num_batches_tracked is not None:
                            bn.num_batches_tracked.add_(1)
                        momentum = 1.0 / float(bn.num_batches_tracked)
                fo = getattr(ctx, "fin_out", None)          # segmented call: this scan's block of the layer's (S,4,C) buffer
                fin = e.bn_finalize(stats, M, gamma, beta, bn.eps, momentum, rm, rv, nbt, out=None if fo is None else fo[l])
            else:
                if lift and l == 0:
                    y, ctx.lift_P = _lift_forward(e, feats, W, group, None)
                elif first_free and l == 0:
                    y = None
                elif first_free and l == 1:
                    y = e.mlp_gemm_first(x, W0c, fins[0], W.contiguous(), epi=e.EPI_NONE)
                else:
                    y = None if pooled_parts is not None else e.mlp_gemm(cur, W, pro=pro, epi=e.EPI_NONE, p=p)
                rstd = torch.rsqrt(bn.running_var + bn.eps)
                scale = gamma * rstd
                fin = torch.stack([bn.running_mean, rstd, scale, beta - bn.running_mean * scale]).contiguous()
            ys.append(y)
            fins.append(fin)
            batch_flags.append(use_batch)
            cur = y
        yraw = None
        if pooled_parts is not None:
            out, arg, yraw = e.pool_finalize(pooled_parts[0], pooled_parts[1], fins[-1], pooled_parts[2], ns)
        elif ns:
            out, arg, yraw = e.bn_relu_rows_max(ys[-1], fins[-1], ns)
        else:
            out, arg = e.bn_relu_apply(ys[-1], fins[-1]), None
        ctx.ns, ctx.L, ctx.batch_flags = ns, L, batch_flags
        ctx.pool_fused = pooled_parts is not None
        ctx.first_free = first_free
        ctx.lift, ctx.M, ctx.K0 = lift, M, K0
        ctx.shapes = [params[3 * l].shape for l in range(L)]
        saved = [feats if lift else x] + ys + fins + [params[3 * l] for l in range(L)] + [params[3 * l + 1] for l in range(L)]
        if ns:
            saved += [out, arg, yraw]
            ctx.mark_non_differentiable(arg)
            ctx.set_materialize_grads(False)       # (no zeros_like(arg) — an int32 (R, C) fill per stack — for the index output)
        # (a piece of the shared zero slab: its version counter moves with every in-place op on a sibling piece, so it
        # travels as an attribute like the geometry tensors, not through save_for_backward)
        ctx.gram = gram
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
        lift = getattr(ctx, "lift", False)            # saved[0] is then the point-major feature tensor, not grouped rows
        M = ctx.M if lift else x.size(0)
        g_out = g_out.contiguous()

        # every zero-initialised accumulator of this backward from one allocation / one fill
        f64, f32 = torch.float64, torch.float32
        # first-layer fold (csrc/mlp_bwd_fused.hip): the layer above the first one reduces gz^T X instead of storing gz
        # when the first layer's input needs no gradient (raw coordinates / colours) and is at most 8 columns wide
        need_dgrad0 = ctx.needs_input_grad[0] and (ctx.group is None or ctx.feat_shape is not None)
        K0 = ctx.K0 if lift else x.size(1)
        fold = (L >= 2 and FUSED_BACKWARD and not need_dgrad0 and ctx.batch_flags[0] and not lift
                and not (ctx.pool_fused and L == 2)          # (layer 1 is then the pooled layer: Gram-form kernel)
                and e.mlp_bwd_fused_fold_supported(Ws[1].size(0), Ws[1].size(1), K0))
        first_free = getattr(ctx, "first_free", False)
        if first_free and not fold:
            raise RuntimeError("fused_mlp: the first layer's output was not stored but the backward cannot fold it")
        arena = e.zero_arena(x.device, [((2, Ws[-1].size(0)), f64)] + [((2, Ws[l].size(1)), f64) for l in range(L)] +
                             [(tuple(Ws[l].shape), f32) for l in range(L)] +
                             ([((Ws[0].size(0), K0), f32), ((K0 * K0 + K0,), f64)] if fold else []) +
                             ([((3 * Ws[0].size(0) + 9,), f32)] if lift else []))
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
            if l == 0 and lift:
                # first layer applied before the grouping: per-point sums of dL/dy0 through the inverse index, then GEMMs
                # over the B N points (csrc/group_lift.hip)
                consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[0], fins[0], ctx.batch_flags[0])
                grads[1], grads[2] = dgamma, dbeta
                if gmode != e.PRO_GY:
                    raise RuntimeError("fused_mlp: the lifted first layer expects the dense gradient of the layer above")
                xyz, new_xyz, idx, _u, normalize, radius = ctx.group[:6]
                N0 = Ws[0].size(0)
                inv = ctx.group[6] if (len(ctx.group) > 6 and ctx.group[6] is not None) else ctx.lift_inv
                lift_P, Wx, WfT = ctx.lift_P
                consts = consts.contiguous()
                S = e.group_lift_rows_grad(G, lift_P, Wx, consts, xyz, new_xyz, inv, idx.size(2), normalize,
                                           radius, arena[-1]).view(-1, N0)
                # dWf = S^T f and dL/df = S Wf over the B N points, on the library's own kernels (a vendor GEMM picks a
                # 32 x 32 tile for the 128 x 128 x 65 536 reduction: 0.18 ms; pn2_mlp_wgrad is built for long reductions)
                dWf = e.mlp_wgrad(S, _unit_consts(S.device, N0), x.view(-1, K0 - 3), e.PRO_GY, e.PRO_NONE, G=S)
                # dW = [dWx + c2 Wx RR | dWf]: the kernel left out c2 Wx RR (RR = sum_r rel rel^T is only complete after the launch)
                grads[0] = e.lift_dw_assemble(arena[-1], Wx, consts[1], dWf).view(ctx.shapes[0])
                if need_dgrad0:
                    gx = e.mlp_gemm(S, WfT, pro=e.PRO_NONE, epi=e.EPI_NONE).view(ctx.feat_shape)
                    dst = getattr(ctx, "gx_out", None)       # (a segmented call: this scan's slice of the batch's gradient)
                    if dst is not None:
                        gx = dst.copy_(gx)
                continue
            if l == L - 1 and ctx.pool_fused:
                # pooled layer in Gram form: y_l was never stored (csrc/pool_bwd.hip)
                consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l])
                grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
                G, dW = e.pool_bwd(ys[l - 1], fins[l - 1], Ws[l].contiguous(), consts, arg, gPm, ns, sums_in[l])
                grads[3 * l] = dW.view(ctx.shapes[l])
                sums = sums_in[l]
                gmode, arg, gPm = e.PRO_GY, None, None
                continue
            lift_free = getattr(ctx, "lift_free", None)
            if l == 1 and lift_free is not None:
                # the layer above a lifted first layer whose output was not stored: weight gradient and input gradient with
                # y0 = Pq[point] - Q[centre] re-formed in the kernels (csrc/mlp_gemm.hip PRO_LIFT / EPI_MASKL)
                if gmode != e.PRO_GY:
                    raise RuntimeError("fused_mlp: the layer above a lifted first layer expects a dense gradient")
                consts, dgamma, dbeta, Wt = e.bn_bwd_consts(sums, M, gammas[1], fins[1], ctx.batch_flags[1],
                                                            W=Ws[1].contiguous(), k0=0)
                grads[4], grads[5] = dgamma, dbeta
                Pq, gidx, Q = lift_free
                nsl = ctx.group[2].size(2)
                grads[3] = e.mlp_wgrad_lift(ys[1], consts, G, Pq, gidx, Q, nsl, fins[0], dW=dWs[1]).view(ctx.shapes[1])
                sums = sums_in[1]
                G = e.mlp_dgrad_lift(G, ys[1], consts, Wt, sums, Pq, gidx, Q, nsl, fins[0])
                continue
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
                if first_free:
                    sums, dW, P1 = e.mlp_bwd_fused_fold_first(ys[1], consts, Ws[1].contiguous(), Ws[0].contiguous(), fins[0], x,
                                                              gmode, G=G, arg=arg, gP=gPm, ns=ns, sums=sums_in[1], dW=dWs[1],
                                                              P1=arena[1 + 2 * L])
                else:
                    sums, dW, P1 = e.mlp_bwd_fused_fold(ys[1], consts, Ws[1].contiguous(), ys[0], fins[0], x, gmode, G=G,
                                                        arg=arg, gP=gPm, ns=ns, sums=sums_in[1], dW=dWs[1],
                                                        P1=arena[1 + 2 * L])
                grads[3] = dW.view(ctx.shapes[1])
                continue
            if l == 0 and fold:
                gram = ctx.gram if first_free else e.rows_gram(x, arena[2 + 2 * L])
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
        if gx is not None and ctx.group is not None and not lift:
            idx = ctx.group[2]
            Bq, npoint, nsample = idx.shape
            Bf, Nf, Cf = ctx.feat_shape
            inv = ctx.group[6] if len(ctx.group) > 6 else None          # (ptr, refs) carried next to idx
            if inv is not None:          # prefetched inverse index: per-point sum, no atomics (csrc/group_csr.hip)
                gx = e.group_rows_grad_csr(gx.view(Bq, npoint, nsample, Cf), inv, Nf, Cf, 0, out=getattr(ctx, "gx_out", None))
            else:                        # (a segmented call hands in its zero-filled slice of the batch's gradient)
                gx = e.group_rows_grad(gx.view(Bq, npoint, nsample, Cf), idx, Nf, Cf, 0, out=getattr(ctx, "gx_out", None))
        return (gx, None, None, None, *grads)


def _lift_eligible(e, layers, use_xyz, feats) -> bool:
    """Shapes the lifted first layer covers: [relative xyz | >= 16 feature channels] into a stack of >= 2 layers whose first
    width is a multiple of 4 up to 256 (fp32 node)."""
    return bool(LIFT_FIRST and feats is not None and use_xyz and len(layers) >= 2 and feats.size(2) >= 16
                and getattr(e, "group_lift_rows", None) and e.group_lift_supported(layers[0][0].out_channels))


_UNIT_CONSTS = {}


def _unit_consts(device, n):
    """(3, n) constants (1, 0, 0): pn2_mlp_wgrad's gradient transform c1 g + c2 y + c3 as the identity.  Cached per
    (device, width): created once, eagerly (a stream capture must not meet the allocation + fill)."""
    key = (device, int(n))
    t = _UNIT_CONSTS.get(key)
    if t is None:
        t = _UNIT_CONSTS[key] = torch.cat([torch.ones(1, n, device=device), torch.zeros(2, n, device=device)]).contiguous()
    return t


def _lift_forward_scans(e, feats, W, group, stat_blocks, seg):
    """_lift_forward for a segment-table stack (bf16 node): ONE per-point GEMM for the batch, then ONE gather launch whose
    grid.y walks the scans — every scan with the grid and the (2, N0) statistics block of its own single-scan call."""
    xyz, new_xyz, idx, _use_xyz, normalize, radius = group[:6]
    B, N, C = feats.shape
    N0 = W.size(0)
    Wx, Wf, WfT = _lift_weights(e, W)
    P = e.mlp_gemm(feats.view(B * N, C), Wf, pro=e.PRO_NONE, epi=e.EPI_NONE).view(B, N, -1)
    per = idx.size(1) * idx.size(2)
    Y = e.group_lift_rows_seg(P, xyz, new_xyz, idx, Wx, normalize, radius, stat_blocks, seg, out_bf16=True)
    return Y, (P, Wx, WfT)


def _lift_weights(e, W):
    """(Wx, Wf, Wf^T) of the lifted layer's weight: one launch where the library has it, three strided copies otherwise."""
    W = W.contiguous()
    if getattr(e, "lift_split_weight", None) is not None and W.is_cuda:
        return e.lift_split_weight(W)
    return W[:, :3].contiguous(), W[:, 3:].contiguous(), W[:, 3:].t().contiguous()


def _lift_forward_free(e, feats, W, group, stats):
    """The lifted first layer without its output: (None, (P, Wx, WfT), (Pq, gidx, Q)) — P as _lift_forward, Pq = P + Wx x / r and
    Q = Wx c / r (y0[row] = Pq[gidx[row]] - Q[row // ns]), the batch statistics of y0 accumulated into `stats`."""
    xyz, new_xyz, idx, _use_xyz, normalize, radius = group[:6]
    B, N, C = feats.shape
    Wx, Wf, WfT = _lift_weights(e, W)
    P = e.mlp_gemm(feats.view(B * N, C), Wf, pro=e.PRO_NONE, epi=e.EPI_NONE).view(B, N, -1)
    Pq, Q = e.lift_points(P, xyz, new_xyz, Wx, normalize, radius)
    gidx = e.group_lift_stats(Pq, Q, idx, N, stats)
    return None, (P, Wx, WfT), (Pq, gidx, Q)


def _lift_forward(e, feats, W, group, stats, out_bf16=False):
    """y0 (B m ns, N0) of a grouped stack's first layer without the grouped rows: P = f Wf^T over the B N points (a plain
    fp32 GEMM), then the gather + coordinate terms + column sums in one kernel (pn2_group_lift_rows)."""
    xyz, new_xyz, idx, _use_xyz, normalize, radius = group[:6]
    B, N, C = feats.shape
    # (the library's own fp32-MFMA GEMM: bit-reproducible from call to call, which a vendor GEMM's kernel choice is not)
    Wx, Wf, WfT = _lift_weights(e, W)
    P = e.mlp_gemm(feats.view(B * N, C), Wf, pro=e.PRO_NONE, epi=e.EPI_NONE).view(B, N, -1)
    # (P travels to the backward as an attribute: 1/16 of y0's size, and the backward recomputes y0 from it; the weight's
    # pieces ride along: the parameters do not change between a step's forward and its backward)
    return e.group_lift_rows(P, xyz, new_xyz, idx, Wx, normalize, radius, stats=stats, out_bf16=out_bf16), (P, Wx, WfT)


class _FusedMLPBf16(Function):
    """The same stack on the bf16 MFMA kernels: y_l and the activation gradients are bf16 tensors, everything that is
    reduced (BatchNorm sums, weight gradients) or small (pooled outputs, per-channel constants) stays fp32 / fp64."""

    @staticmethod
    def forward(ctx, x, ns, layers, group, *params):
        e = _ext()
        ctx.group = group
        seg = getattr(ctx, "seg", None)
        lift = False
        feats = None
        ctx.lift_inv = None
        if group is not None:
            xyz, new_xyz, idx, use_xyz, normalize, radius = group[:6]
            ctx.feat_shape = None if x is None else tuple(x.shape)
            k_in = (3 if use_xyz else 0) + (0 if x is None else x.size(2))
            pre = getattr(ctx, "x_rows", None)          # a segmented call groups the whole batch once and hands in the rows
            if pre is None and len(group) > 8 and group[8] is not None and group[8].dtype == torch.bfloat16:
                pre = group[8]                           # bf16 rows grouped next to the ball query (prefetched geometry)
            # first layer before the grouping, as on the fp32 node (csrc/group_lift.hip): per-point products in fp32, y0 and
            # the gradient rows in bf16 — no grouped tensor, no M-row first-layer GEMMs, no feature-gradient scatter
            has_inv = len(group) > 6 and group[6] is not None
            crowded = len(group) > 7 and bool(group[7])
            # (with a segment table: one launch per scan for this layer — the launches of single-scan steps, each scan's
            # column sums in its own block — and the table for the rest of the stack)
            lift = bool(BF16_LIFT and pre is None and _lift_eligible(e, layers, use_xyz, x)
                        and (seg is None or (BF16_LIFT_SEG and all(r % (idx.size(1) * idx.size(2)) == 0 for r in seg.rows)))
                        and (not any(ctx.needs_input_grad) or has_inv or crowded or LIFT_SPARSE))
            if lift:
                if any(ctx.needs_input_grad) and not has_inv:
                    ctx.lift_inv = tuple(e.group_inverse_index(idx, xyz.size(1)))      # (not prefetched: built here)
                feats = x.contiguous()
                x = None
            else:
                if pre is None:
                    pre = e.group_concat_rows_bf16(xyz, new_xyz, None if x is None else x.contiguous(), idx, use_xyz,
                                                   normalize, radius)
                x = pre.view(-1, pre.size(-1))                              # (M, pad8(k_in)) bf16, zero pad columns
        else:
            x = x.contiguous()                                              # fp32 rows of any width
            k_in = x.size(1)
        M = idx.numel() if lift else x.size(0)
        L = len(layers)
        ys, fins, batch_flags = [], [], []
        # segment table (a _SegTableMLP call): the rows are S scans with their own batch statistics — every per-channel
        # buffer gets a leading scan dimension and each kernel walks the scans in its grid (csrc: pn2_*_seg)
        lead = () if seg is None else (seg.nseg,)
        stat_bufs = e.zero_arena((feats if lift else x).device,
                                 [(lead + (2, conv.out_channels), torch.float64) for conv, _ in layers])
        cur = x
        bnL = layers[-1][1]
        per_scan = group is not None and len(group) > 9 and bool(group[9])
        pool_bf = bool(BF16_POOL and ns and L >= 3 and seg is None and not per_scan and not isinstance(ctx, _SegCtx) and M % ns == 0
                       and (bnL.training or bnL.running_mean is None) and getattr(e, "mlp_gemm_pool_bf16", None) is not None
                       and e.pool_layer_bf16_supported(layers[-1][0].in_channels, layers[-1][0].out_channels, ns))
        pooled_parts = None
        # first layer without its output: when gradients are needed, only if the backward will take the fold path
        need_dgrad0 = ctx.needs_input_grad[0] and (group is None or ctx.feat_shape is not None)
        first_bf = bool(
            BF16_FIRST and not lift and seg is None and not per_scan and not isinstance(ctx, _SegCtx) and L >= 2
            and not (pool_bf and L == 2)
            and x.dtype == torch.bfloat16 and x.size(1) == 8 and k_in <= 8
            and all((bn_.training or bn_.running_mean is None) for _c, bn_ in layers[:2])
            and getattr(e, "mlp_gemm_first_bf16", None) is not None
            and e.mlp_gemm_first_bf16_supported(k_in, layers[0][0].out_channels, layers[1][0].out_channels)
            and (not any(ctx.needs_input_grad) or (BF16_FOLD and FUSED_BACKWARD and not need_dgrad0 and e.mlp_bwd_bf16_fold_supported(
                layers[1][0].out_channels, layers[0][0].out_channels, k_in))))
        ctx.first_bf, ctx.gram_bf = first_bf, None
        W0c = params[0].view(layers[0][0].out_channels, -1)[:, :k_in].contiguous() if first_bf else None
        for l, (conv, bn) in enumerate(layers):
            W = params[3 * l].view(conv.out_channels, conv.in_channels)
            gamma, beta = params[3 * l + 1], params[3 * l + 2]
            use_batch = bn.training or bn.running_mean is None
            pro = e.PRO_NONE if l == 0 else e.PRO_BNRELU
            p = None if l == 0 else ((fins[-1][2], fins[-1][3]) if seg is None else (fins[-1][:, 2], fins[-1][:, 3]))
            if pool_bf and l == L - 1:
                Wfl, sgn = e.pool_flip_rows(W.contiguous(), gamma)
                pooled_parts = e.mlp_gemm_pool_bf16(cur, Wfl, sgn, ns, p, stat_bufs[l]) + (sgn,)
            if seg is not None:
                if not use_batch:
                    raise RuntimeError("fused_mlp: a segment table needs training-mode BatchNorm in every layer")
                if lift and l == 0:
                    y, ctx.lift_P = _lift_forward_scans(e, feats, W, group, stat_bufs[0], seg)
                else:
                    y = e.mlp_gemm_bf16(cur, W, pro=pro, epi=e.EPI_STATS, p=p, stats=stat_bufs[l], seg=seg)
                rm = rv = nbt = None
                if bn.training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None:
                    rm, rv, nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
                # running statistics: the S momentum updates in scan order inside the kernel (a cumulative average,
                # momentum None, needs the batch count on the host: _SegTableMLP applies it afterwards)
                fin = e.bn_finalize_seg(stat_bufs[l], seg, gamma, beta, bn.eps, 0.0 if rm is None else bn.momentum, rm, rv, nbt)
            elif use_batch:
                stats = stat_bufs[l]
                if lift and l == 0:
                    y, ctx.lift_P = _lift_forward(e, feats, W, group, stats, out_bf16=True)
                elif first_bf and l == 0:
                    y = None
                    ctx.gram_bf = e.rows_gram_bf16(x, k_in, e.zero_arena(x.device, [((k_in * k_in + k_in,), torch.float64)])[0])
                    e.first_layer_stats(W0c, ctx.gram_bf, stats)
                elif first_bf and l == 1:
                    y = e.mlp_gemm_first_bf16(x, k_in, W0c, fins[0], W.contiguous(), stats)
                elif pooled_parts is not None:
                    y = None
                else:
                    y = e.mlp_gemm_bf16(cur, W, pro=pro, epi=e.EPI_STATS, p=p, stats=stats)
                momentum = 0.0
                rm = rv = nbt = None
                if (bn.training and bn.track_running_stats and bn.running_mean is not None
                        and not getattr(ctx, "defer_running", False)):     # (a segmented call updates them itself, in scan order)
                    rm, rv = bn.running_mean, bn.running_var
                    if bn.momentum is not None:
                        momentum, nbt = bn.momentum, bn.num_batches_tracked
                    else:
                        if bn.num_batches_tracked is not None:
                            bn.num_batches_tracked.add_(1)
                        momentum = 1.0 / float(bn.num_batches_tracked)
                fo = getattr(ctx, "fin_out", None)          # segmented call: this scan's block of the layer's (S,4,C) buffer
                fin = e.bn_finalize(stats, M, gamma, beta, bn.eps, momentum, rm, rv, nbt, out=None if fo is None else fo[l])
            else:
                if lift and l == 0:
                    y, ctx.lift_P = _lift_forward(e, feats, W, group, None, out_bf16=True)
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
        ctx.pool_bf = pooled_parts is not None
        if pooled_parts is not None:
            out, arg, yraw = e.pool_finalize(pooled_parts[0], pooled_parts[1], fins[-1], pooled_parts[2], ns)
        elif ns:
            out, arg, yraw = e.bn_relu_rows_max_bf16(ys[-1], fins[-1], ns, seg=seg)
        elif seg is not None:
            raise RuntimeError("fused_mlp: a segment table needs a pooled stack")
        else:
            out, arg = e.bn_relu_apply_bf16(ys[-1], fins[-1]), None
        ctx.ns, ctx.L, ctx.batch_flags, ctx.k_in = ns, L, batch_flags, k_in
        ctx.lift, ctx.M = lift, M
        ctx.shapes = [params[3 * l].shape for l in range(L)]
        saved = [feats if lift else x] + ys + fins + [params[3 * l] for l in range(L)] + [params[3 * l + 1] for l in range(L)]
        if ns:
            saved += [out, arg, yraw]
            ctx.mark_non_differentiable(arg)
            ctx.set_materialize_grads(False)       # (no zeros_like(arg) — an int32 (R, C) fill per stack — for the index output)
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
        lift = getattr(ctx, "lift", False)            # saved[0] is then the point-major fp32 feature tensor, not grouped rows
        M = ctx.M if lift else x.size(0)
        g_out = g_out.contiguous()
        f64, f32 = torch.float64, torch.float32
        need_dgrad0 = ctx.needs_input_grad[0] and (ctx.group is None or ctx.feat_shape is not None)
        # first-layer fold (csrc/mlp_bf16.hip, FOLD): as on the fp32 path — the layer above the first one reduces gz^T X
        # instead of storing gz when the (<= 8-column, bf16, pitch 8) input rows need no gradient
        k_in = ctx.k_in
        seg = getattr(ctx, "seg", None)
        lead = () if seg is None else (seg.nseg,)
        fold = bool(BF16_FOLD and not lift and seg is None and L >= 2 and FUSED_BACKWARD and not need_dgrad0 and ctx.batch_flags[0]
                    and x.dtype == torch.bfloat16 and x.size(1) == 8 and k_in <= 8
                    and getattr(e, "mlp_bwd_bf16_fold", None)
                    and e.mlp_bwd_bf16_fold_supported(Ws[1].size(0), Ws[1].size(1), k_in))
        arena = e.zero_arena(x.device, [(lead + (2, Ws[-1].size(0)), f64)] +
                             [(lead + (2, Ws[l].size(1)), f64) for l in range(L)] +
                             [(tuple(Ws[l].shape), f32) for l in range(L)] +
                             ([((Ws[0].size(0), k_in), f32), ((k_in * k_in + k_in,), f64)] if fold else []) +
                             ([((3 * Ws[0].size(0) + 9,), f32)] if lift else []))
        sums0, sums_in, dWs = arena[0], arena[1:1 + L], arena[1 + L:1 + 2 * L]
        if getattr(ctx, "first_bf", False) and not fold:
            raise RuntimeError("fused_mlp: the bf16 first layer's output was not stored but the backward cannot fold it")
        if ns:
            pooled, arg, yraw = saved[1 + 4 * L], saved[2 + 4 * L], saved[3 + 4 * L]
            gPm, sums = e.pool_bwd_prep(yraw, pooled, g_out, fins[-1], sums=sums0, seg=seg, ns=ns)
            gmode, G = e.PRO_POOLG, None
        else:
            G, sums = e.bn_relu_bwd_prep_bf16(ys[-1], g_out, fins[-1], sums=sums0)
            gmode, arg, gPm = e.PRO_GY, None, None
        P1 = None

        grads = [None] * (3 * L)
        gx = None
        for l in range(L - 1, -1, -1):
            if l == 0 and lift:
                # first layer applied before the grouping: per-point sums of dL/dy0 (bf16 gradient rows, fp32 sums) through
                # the inverse index, then fp32 GEMMs over the B N points — the fp32 node's code (csrc/group_lift.hip)
                if gmode != e.PRO_GY or G is None:
                    raise RuntimeError("fused_mlp: the lifted first layer expects the dense gradient of the layer above")
                xyz, new_xyz, idx, _u, normalize, radius = ctx.group[:6]
                N0, Kf = Ws[0].size(0), Ws[0].size(1) - 3
                inv = ctx.group[6] if (len(ctx.group) > 6 and ctx.group[6] is not None) else ctx.lift_inv
                lift_P, Wx, WfT = ctx.lift_P
                if seg is not None:
                    # per-scan constants, one launch sequence per scan on that scan's points (the whole batch's gradient
                    # rows, centres and row ids are indexed in place)
                    consts, dgamma, dbeta = e.bn_bwd_consts_seg(sums, seg, gammas[0], fins[0], ctx.batch_flags[0])
                    grads[1], grads[2] = dgamma, dbeta
                    Bq, Nq = xyz.size(0), xyz.size(1)
                    per = idx.size(1) * idx.size(2)
                    accs = e.zero_arena(x.device, [((seg.nseg, 3 * N0 + 9), f32)])[0]
                    S = e.group_lift_rows_grad_seg(G, lift_P, Wx, consts.contiguous(), xyz, new_xyz, inv, idx.size(2),
                                                   normalize, radius, accs, seg)
                    S = S.view(-1, N0)
                    RR = accs[:, 3 * N0:].view(seg.nseg, 3, 3)
                    dWx = accs[:, :3 * N0].view(seg.nseg, N0, 3).sum(0) + torch.einsum("sn,nk,skj->nj", consts[:, 1], Wx, RR)
                    dWf = e.mlp_wgrad(S, _unit_consts(S.device, N0), x.view(-1, Kf), e.PRO_GY, e.PRO_NONE, G=S)
                    grads[0] = torch.cat([dWx, dWf], dim=1).view(ctx.shapes[0])
                else:
                    consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[0], fins[0], ctx.batch_flags[0])
                    grads[1], grads[2] = dgamma, dbeta
                    consts = consts.contiguous()
                    S = e.group_lift_rows_grad(G, lift_P, Wx, consts, xyz, new_xyz, inv, idx.size(2), normalize,
                                               radius, arena[-1]).view(-1, N0)
                    dWf = e.mlp_wgrad(S, _unit_consts(S.device, N0), x.view(-1, Kf), e.PRO_GY, e.PRO_NONE, G=S)
                    grads[0] = e.lift_dw_assemble(arena[-1], Wx, consts[1], dWf).view(ctx.shapes[0])
                if need_dgrad0:
                    gx = e.mlp_gemm(S, WfT, pro=e.PRO_NONE, epi=e.EPI_NONE).view(ctx.feat_shape)
                    dst = getattr(ctx, "gx_out", None)
                    if dst is not None:
                        gx = dst.copy_(gx)
                continue
            if l == L - 1 and getattr(ctx, "pool_bf", False):
                # pooled last layer whose output was not stored: y_L is re-formed inside the kernel (csrc/mlp_bf16.hip RECOMP)
                consts, dgamma, dbeta, Wt = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l],
                                                            W=Ws[l].contiguous(), k0=0)
                grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
                G, sums, dW = e.mlp_bwd_bf16_pool(consts, Wt, ys[l - 1], fins[l - 1], arg, gPm, ns, sums=sums_in[l], dW=dWs[l])
                grads[3 * l] = dW.view(ctx.shapes[l])
                gmode, arg, gPm = e.PRO_GY, None, None
                continue
            need_dgrad = l > 0 or need_dgrad0
            k0 = 3 if (l == 0 and ctx.group is not None and ctx.group[3]) else 0
            if seg is not None:
                res = e.bn_bwd_consts_seg(sums, seg, gammas[l], fins[l], ctx.batch_flags[l],
                                          W=Ws[l].contiguous() if need_dgrad else None, k0=k0)
                consts, dgamma, dbeta = res[:3]
                Wt = res[3] if need_dgrad else None
            elif need_dgrad:
                consts, dgamma, dbeta, Wt = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l],
                                                            W=Ws[l].contiguous(), k0=k0)
            else:
                consts, dgamma, dbeta = e.bn_bwd_consts(sums, M, gammas[l], fins[l], ctx.batch_flags[l])
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if l == 1 and fold and getattr(ctx, "first_bf", False):
                W0c = Ws[0][:, :k_in].contiguous()
                sums, dW, P1 = e.mlp_bwd_bf16_fold_first(ys[1], consts, Wt, W0c, fins[0], x, k_in, gmode, G=G, arg=arg, gP=gPm,
                                                         ns=ns, sums=sums_in[1], dW=dWs[1], P1=arena[1 + 2 * L])
                grads[3] = dW.view(ctx.shapes[1])
                gmode, arg, gPm, G = e.PRO_GY, None, None, None
                continue
            if l == 1 and fold:
                sums, dW, P1 = e.mlp_bwd_bf16_fold(ys[1], consts, Wt, ys[0], fins[0], x, k_in, gmode, G=G, arg=arg, gP=gPm, ns=ns,
                                                   sums=sums_in[1], dW=dWs[1], P1=arena[1 + 2 * L])
                grads[3] = dW.view(ctx.shapes[1])
                gmode, arg, gPm, G = e.PRO_GY, None, None, None
                continue
            if l == 0 and fold:
                gram = ctx.gram_bf if getattr(ctx, "gram_bf", None) is not None else e.rows_gram_bf16(x, k_in, arena[2 + 2 * L])
                grads[0] = e.first_layer_dw(consts, P1, Ws[0].contiguous(), gram).view(ctx.shapes[0])
                continue
            if l > 0 and FUSED_BACKWARD and e.mlp_bwd_bf16_supported(Ws[l].size(0), Ws[l].size(1)):
                # hidden layer: dgrad + wgrad from one read of (g, y_l, y_{l-1})
                G, sums, dW = e.mlp_bwd_bf16(ys[l], consts, Wt, ys[l - 1], fins[l - 1], gmode, G=G, arg=arg, gP=gPm, ns=ns,
                                             sums=sums_in[l], dW=dWs[l], seg=seg)
                grads[3 * l] = dW.view(ctx.shapes[l])
                gmode, arg, gPm = e.PRO_GY, None, None
                continue
            act = x if l == 0 else ys[l - 1]
            dW = e.mlp_wgrad_bf16(ys[l], consts, act, gmode, e.PRO_NONE if l == 0 else e.PRO_BNRELU, Ws[l].size(1),
                                  G=G, arg=arg, gP=gPm, ns=ns, a_fin=None if l == 0 else fins[l - 1], dW=dWs[l], seg=seg)
            grads[3 * l] = dW.view(ctx.shapes[l])
            if need_dgrad:
                p = (consts[0], consts[1], consts[2]) if seg is None else (consts[:, 0], consts[:, 1], consts[:, 2])
                if l > 0:
                    sums = sums_in[l]
                    G = e.mlp_gemm_bf16(G, Wt, pro=gmode, epi=e.EPI_MASK, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns,
                                        stats=sums, Yprev=ys[l - 1], e_fin=fins[l - 1], M=M, seg=seg)
                    gmode, arg, gPm = e.PRO_GY, None, None
                else:
                    # gradient rows of a grouped first layer stay bf16 (the scatter / per-point sum accumulates in fp32);
                    # a plain row input gets its fp32 gradient directly
                    rows_bf16 = ctx.group is not None and Wt.size(0) % 4 == 0
                    gx = e.mlp_gemm_bf16(G, Wt, pro=gmode, epi=e.EPI_NONE, X2=ys[l], p=p, arg=arg, gP=gPm, ns=ns, M=M,
                                         out_f32=not rows_bf16, seg=seg)
        if gx is not None and ctx.group is not None and not lift:
            idx = ctx.group[2]
            Bq, npoint, nsample = idx.shape
            Bf, Nf, Cf = ctx.feat_shape
            inv = ctx.group[6] if len(ctx.group) > 6 else None          # (ptr, refs) carried next to idx
            if inv is not None:          # prefetched inverse index: per-point sum, no atomics (csrc/group_csr.hip)
                gx = e.group_rows_grad_csr(gx.view(Bq, npoint, nsample, Cf), inv, Nf, Cf, 0, out=getattr(ctx, "gx_out", None))
            else:                        # (a segmented call hands in its zero-filled slice of the batch's gradient)
                gx = e.group_rows_grad(gx.view(Bq, npoint, nsample, Cf), idx, Nf, Cf, 0, out=getattr(ctx, "gx_out", None))
        return (gx, None, None, None, *grads)


class _SegCtx:
    """What _FusedMLP / _FusedMLPBf16 need from an autograd ctx, for one scan of a segmented call."""
    defer_running = True        # the scans may run concurrently: running statistics are updated after all of them

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


_EMA_WEIGHTS = collections.OrderedDict()       # bounded (LRU): scans of varying size give ever new row-count tuples
_EMA_WEIGHTS_MAX = 4096


def _ema_weights(device, dtype, momentum, rows_per_scan):
    """(w (S): weight of scan s's statistic after S momentum updates, wu (S): w times the biased -> unbiased variance
    factor n / (n - 1)).  Cached per configuration: built from host numbers (a copy), which a stream capture does not
    allow — the first, eager call of a step signature creates them, replays find them."""
    key = (device, dtype, momentum, tuple(rows_per_scan))
    hit = _EMA_WEIGHTS.get(key)
    if hit is not None:
        _EMA_WEIGHTS.move_to_end(key)
    else:
        if len(_EMA_WEIGHTS) >= _EMA_WEIGHTS_MAX:
            _EMA_WEIGHTS.popitem(last=False)
        S = len(rows_per_scan)
        w = [momentum * (1.0 - momentum) ** (S - 1 - s) for s in range(S)]
        wu = [ws * n / max(n - 1, 1) for ws, n in zip(w, rows_per_scan)]
        hit = _EMA_WEIGHTS[key] = (torch.tensor(w, dtype=dtype, device=device), torch.tensor(wu, dtype=dtype, device=device))
    return hit


def _update_running_stats(layers, fins, rows_per_scan):
    """The running statistics after S single-scan training steps, in scan order, from the scans' batch statistics.
    `fins[l]` (S,4,C_l): rows 0 / 1 of every scan = mean / rstd as the finalize kernel leaves them.  running <- (1 - m)
    running + m stat, S times, in closed form — ONE launch per layer (pn2_bn_running_update; as torch ops the same update
    was ~16 tiny kernels per layer, 670 per 8-scan step of the scene-graph model); unbiased variance like
    torch.nn.functional.batch_norm.  momentum None (cumulative average) and CPU tensors (unit test) take the torch form."""
    S = len(rows_per_scan)
    with torch.no_grad():
        for (_, bn), F in zip(layers, fins):
            if F is None or not (bn.training and bn.track_running_stats and bn.running_mean is not None):
                continue
            if bn.momentum is not None and F.is_cuda:
                mom = float(bn.momentum)
                w, wu = _ema_weights(F.device, F.dtype, mom, rows_per_scan)
                _ext().bn_running_update(F, bn.eps, (1.0 - mom) ** S, w, wu, bn.running_mean, bn.running_var,
                                         bn.num_batches_tracked)
                continue
            mean, rstd = F[:, 0], F[:, 1]
            n = torch.tensor(rows_per_scan, dtype=F.dtype, device=F.device).unsqueeze(1)
            var = (1.0 / (rstd * rstd) - bn.eps).clamp_min(0) * n / (n - 1).clamp_min(1)
            if bn.momentum is None:
                for s in range(S):
                    bn.num_batches_tracked += 1
                    f = 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.lerp_(mean[s], f)
                    bn.running_var.lerp_(var[s], f)
            else:
                mom = float(bn.momentum)
                w = (mom * (1.0 - mom) ** torch.arange(S - 1, -1, -1, device=F.device, dtype=F.dtype)).unsqueeze(1)
                bn.running_mean.mul_((1.0 - mom) ** S).add_((w * mean).sum(0))
                bn.running_var.mul_((1.0 - mom) ** S).add_((w * var).sum(0))
                bn.num_batches_tracked += S


#: streams the scans of a segmented call are spread over (1 = all on the calling stream).  One scan's layer chain is
#: ~20 dependent kernels of 10-90 us each — most of them too small to fill 256 CUs and all of them paying their launch
#: latency in series; the scans are independent, so chains on different streams overlap on the chip.
#: Streams are mapped onto FOUR hardware queues: the training stream, the geometry prefetch and the second encoder stream of
#: the scene-graph model (SGPNModelWrapper.encoder_streams) take three, and from the fifth stream on two of them share a queue
#: and run strictly one after the other (8 scans fp32: 1 stream here 199, 2 streams 194, 3-4 streams 122 scans/s; without the
#: encoder fork 167 / 184).
SEGMENT_STREAMS = 1 if _os.environ.get("PN2_ENCODER_STREAMS", "1") == "1" else 2
_WORKERS = {}


def _worker_streams(device, n):
    """n - 1 side streams per device (created once, high priority like the geometry prefetch stream: default-priority
    streams can share a hardware queue with the calling stream, profiles/r02_stream_queue_aliasing.md)."""
    pool = _WORKERS.setdefault(device, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=device, priority=-1))
    return pool[:n - 1]


class _Fork:
    """Round-robin scans over the calling stream and the worker streams; `join()` makes the calling stream wait for all
    of them.  Tensors a worker allocated and the calling stream reads afterwards are registered with the allocator
    (record_stream), or their blocks could be handed out again on the worker while the calling stream still reads them."""

    def __init__(self, device, n_scans):
        self.main = torch.cuda.current_stream(device)
        n = max(1, min(int(SEGMENT_STREAMS), n_scans)) if device.type == "cuda" else 1
        self.streams = [self.main] + (_worker_streams(device, n) if n > 1 else [])
        if len(self.streams) > 1:
            start = torch.cuda.Event()
            start.record(self.main)
            for st in self.streams[1:]:
                st.wait_event(start)

    def stream(self, s):
        return self.streams[s % len(self.streams)]

    def join(self, *tensor_lists):
        for st in self.streams[1:]:
            self.main.wait_stream(st)
        if len(self.streams) > 1:
            for tensors in tensor_lists:
                for t in tensors:
                    if t is not None:
                        t.record_stream(self.main)


class _SegmentedGroupMLP(Function):
    """fused_group_mlp_pool over a batch of S scans with PER-SCAN BatchNorm statistics, as ONE autograd node: the clouds
    [c_s, c_{s+1}) of every scan run through the inner node (own batch statistics) and the S results are concatenated; the
    running statistics then receive the S momentum updates in scan order (_update_running_stats).  Equivalent to S
    separate nodes + split/cat (pointnet2_modules.sa_scale_rows without the fused kernels does exactly that) — but the
    scans' kernel chains are issued on SEGMENT_STREAMS streams so that they overlap on the GPU."""

    @staticmethod
    def forward(ctx, x, ns, layers, group, sizes, inner, *params):
        xyz, new_xyz, idx, use_xyz, normalize, radius = group[:6]
        m = idx.size(1)
        # the grouping has no statistics: ONE gather for the whole batch (before the fork), the scans take row slices
        # (same-box A/B at 8 scans, bf16: 195.4 -> 202.4 scans/s)
        e = _ext()
        feats = None if x is None else x.contiguous()
        # fp32 stacks whose first layer can be applied before the grouping (_lift_eligible) form no rows at all: every scan's
        # inner call lifts, with its slice of ONE inverse index of the whole batch (rows and points of a scan are contiguous)
        lift_all = bool((inner is _FusedMLP or (inner is _FusedMLPBf16 and BF16_LIFT)) and _lift_eligible(e, layers, use_xyz, feats)
                        and not (len(group) > 8 and group[8] is not None))
        inv_all = None
        if lift_all and any(ctx.needs_input_grad):
            inv_all = group[6] if (len(group) > 6 and group[6] is not None) else tuple(e.group_inverse_index(idx, xyz.size(1)))
        if lift_all:
            rows_all = None
        elif inner is _FusedMLPBf16:
            pre16 = group[8] if (len(group) > 8 and group[8] is not None and group[8].dtype == torch.bfloat16) else None
            rows_all = pre16 if pre16 is not None else e.group_concat_rows_bf16(xyz, new_xyz, feats, idx, use_xyz, normalize, radius)
        elif len(group) > 8 and group[8] is not None:
            rows_all = group[8]                         # emitted by the fused query + grouping kernel
        else:
            rows_all = e.group_concat_rows(xyz, new_xyz, feats, idx, use_xyz, normalize, radius)
        # per layer ONE (S,4,C) buffer for the scans' (mean | rstd | scale | shift) blocks: the running-statistics update
        # reads it as it lies
        fin_bufs = [torch.empty(len(sizes), 4, conv.out_channels, dtype=torch.float32, device=idx.device)
                    if (bn.training or bn.running_mean is None) else None for conv, bn in layers]
        fork = _Fork(idx.device, len(sizes))
        subs, outs, args, c0 = [], [], [], 0
        for s, n_clouds in enumerate(sizes):
            c1 = c0 + n_clouds
            # (x, ns, layers, group, *params) of the inner node <- (x, ns, layers, group, sizes, inner, *params) here
            sub = _SegCtx(tuple(ctx.needs_input_grad[:4]) + tuple(ctx.needs_input_grad[6:]))
            sub.x_rows = None if lift_all else rows_all[c0:c1].view(-1, rows_all.size(-1))
            sub.fin_out = [None if F is None else F[s] for F in fin_bufs]
            inv_s = None
            if inv_all is not None:
                r0, n_pts = c0 * m * idx.size(2), xyz.size(1)
                inv_s = (inv_all[0][c0 * n_pts:c1 * n_pts + 1] - r0, inv_all[1][r0:c1 * m * idx.size(2)] - r0)
            g = (xyz[c0:c1], new_xyz[c0:c1], idx[c0:c1], use_xyz, normalize, radius, inv_s, group[7] if len(group) > 7 else None,
                 None)
            with torch.cuda.stream(fork.stream(s)):
                out, arg = inner.forward(sub, None if x is None else x[c0:c1], ns, layers, g, *params)
            subs.append(sub)
            outs.append(out)
            args.append(arg)
            c0 = c1
        fork.join(outs, args)
        ctx.subs, ctx.inner, ctx.rows, ctx.sizes = subs, inner, [n * m for n in sizes], sizes
        ctx.feat_shape = None if x is None else tuple(x.shape)
        _update_running_stats(layers, fin_bufs, [n * m * ns for n in sizes])
        out, arg = torch.cat(outs, 0), torch.cat(args, 0)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)       # (no zeros_like(arg) — an int32 (R, C) fill per stack — for the index output)
        return out, arg

    @staticmethod
    def backward(ctx, g_out, *unused):
        g_out = g_out.contiguous()
        # the scans scatter their feature gradient straight into their slice of ONE zero-filled (B, N, C) tensor (allocated
        # and cleared on the calling stream before the fork) instead of S tensors + a concatenating copy
        gx_all = None
        if ctx.needs_input_grad[0] and ctx.feat_shape is not None:
            gx_all = torch.zeros(ctx.feat_shape, dtype=torch.float32, device=g_out.device)
        fork = _Fork(g_out.device, len(ctx.subs))
        gxs, pgs, r0, c0 = [], [], 0, 0
        for s, (sub, rows, n_clouds) in enumerate(zip(ctx.subs, ctx.rows, ctx.sizes)):
            sub.gx_out = None if gx_all is None else gx_all[c0:c0 + n_clouds]
            with torch.cuda.stream(fork.stream(s)):      # the stream this scan's forward ran on (same round-robin)
                res = ctx.inner.backward(sub, g_out[r0:r0 + rows])
            r0 += rows
            c0 += n_clouds
            gxs.append(res[0])
            pgs.append(list(res[4:]))
        fork.join(*pgs)
        acc = pgs[0]
        for pg in pgs[1:]:                                          # one multi-tensor add per scan
            torch._foreach_add_(acc, pg)
        ctx.subs = None
        gx = gx_all if (gx_all is not None and gxs[0] is not None) else None
        return (gx, None, None, None, None, None, *acc)


#: per-scan statistics through the kernels' segment tables (ONE launch per kernel for all scans of the batch) where the stack
#: allows it — bf16 node, pooled, every BatchNorm in training mode; False: the per-scan loop of _SegmentedGroupMLP
SEG_TABLE = _os.environ.get("PN2_SEG_TABLE", "1") != "0"


def seg_table_ok(layers, ns, inner) -> bool:
    return bool(SEG_TABLE and inner is _FusedMLPBf16 and ns and getattr(_ext(), "bn_finalize_seg", None)
                and all(bn.training or bn.running_mean is None for _, bn in layers))


def seg_table_supported(mlp: nn.Module, x: torch.Tensor, ns: int) -> bool:
    """`mlp` over rows like `x` can take fused_shared_mlp(..., rows_per_scan=...)."""
    if not supported(mlp, x, ns):
        return False
    layers = parse_stack(mlp)
    return seg_table_ok(layers, ns, _node(layers, ns))


class _SegTableMLP(Function):
    """A pooled stack over a batch of S scans with PER-SCAN BatchNorm statistics at the launch count of one call: the inner
    node runs ONCE over all rows with the scans' row ranges as a segment table (csrc: pn2_*_seg — per-scan sums, finalize
    blocks and backward constants inside the kernels; weight gradients summed over the scans), then the running statistics
    receive the S momentum updates in scan order.  Same arithmetic per scan as _SegmentedGroupMLP's loop (a scan's row
    tiles, sums and constants are its own), i.e. the reference's one-scan steps (SGP/main.py:54-56)."""

    @staticmethod
    def forward(ctx, x, ns, layers, group, rows_per_scan, inner, *params):
        # (x, ns, layers, group, *params) of the inner node <- (x, ns, layers, group, rows_per_scan, inner, *params) here
        sub = _SegCtx(tuple(ctx.needs_input_grad[:4]) + tuple(ctx.needs_input_grad[6:]))
        dev = (x if x is not None else group[0]).device
        sub.seg = _ext().SegTable.get(dev, rows_per_scan)
        out, arg = inner.forward(sub, x, ns, layers, group, *params)
        L = len(layers)
        if sub.seg.total != (sub.M if getattr(sub, "lift", False) else sub.saved_tensors[0].size(0)):
            raise RuntimeError("fused_mlp: the scans' row counts must sum to the rows of the stack")
        _update_running_stats(layers, [F if bn.momentum is None else None                 # (the others: in the finalize kernel)
                                       for (_, bn), F in zip(layers, sub.saved_tensors[1 + L:1 + 2 * L])], list(rows_per_scan))
        ctx.sub, ctx.inner = sub, inner
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)       # (no zeros_like(arg) — an int32 (R, C) fill per stack — for the index output)
        return out, arg

    @staticmethod
    def backward(ctx, g_out, *unused):
        res = ctx.inner.backward(ctx.sub, g_out)
        ctx.sub = None
        return (res[0], None, None, None, None, None, *res[4:])


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


def fused_shared_mlp(mlp: nn.Module, x: torch.Tensor, ns: int = 0, rows_per_scan: Optional[Sequence[int]] = None
                     ) -> torch.Tensor:
    """x (M, C_in) rows -> (M, C_out) [ns == 0] or (M // ns, C_out) max-pooled over groups of ns rows.
    `rows_per_scan` (sums to M, multiples of ns): BatchNorm batch statistics per scan — only for stacks with
    seg_table_ok(); other callers split the rows themselves."""
    layers = parse_stack(mlp)
    assert layers is not None, "fused_shared_mlp: unsupported stack (call supported() first)"
    node = _node(layers, ns)
    if rows_per_scan is not None and len(rows_per_scan) > 1:
        if not seg_table_ok(layers, ns, node):
            raise RuntimeError("fused_shared_mlp: rows_per_scan needs a stack with segment-table support")
        if sum(rows_per_scan) != x.size(0) or any(r % int(ns) for r in rows_per_scan):
            raise RuntimeError("fused_shared_mlp: rows_per_scan must sum to the rows and be multiples of ns")
        return _SegTableMLP.apply(x, int(ns), layers, None, tuple(int(r) for r in rows_per_scan), node, *_params(layers))[0]
    res = node.apply(x, int(ns), layers, None, *_params(layers))
    return res[0] if ns else res


def fused_group_mlp_pool(mlp: nn.Module, xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius,
                         clouds_per_scan: Optional[Sequence[int]] = None, inv=None, crowded: Optional[bool] = None,
                         rows=None, per_scan_caller: bool = False) -> torch.Tensor:
    """Ball-query neighbourhoods -> shared MLP -> max, one autograd node:
    xyz (B,N,3), new_xyz (B,m,3), feats_rows (B,N,C)|None, idx (B,m,ns) -> (B, m, C_out).
    `clouds_per_scan` (sums to B): BatchNorm batch statistics per scan (see _SegmentedGroupMLP)."""
    layers = parse_stack(mlp)
    assert layers is not None
    B, m, ns = idx.shape
    # inv: (ptr, refs) of the whole batch's idx; crowded: N r^3 > 4 nsample (pointnet2_modules.crowded_balls) -> the pooled
    # last layer runs without its output tensor (csrc/pool_bwd.hip)
    if rows is not None:
        width = (3 if use_xyz else 0) + (0 if feats_rows is None else feats_rows.size(2))
        if rows.dtype == torch.bfloat16:
            # bf16 rows (pitch rounded up to 8 columns) grouped next to the ball query for the bf16 node; another node
            # (the arithmetic was switched in between) groups for itself
            if tuple(rows.shape) != (B, m, ns, (width + 7) // 8 * 8) or rows.device != idx.device:
                raise RuntimeError(f"fused_group_mlp_pool: pre-grouped bf16 rows must be ({B}, {m}, {ns}, pad8({width})) on {idx.device}")
            if _node(layers, ns) is not _FusedMLPBf16:
                rows = None
        elif tuple(rows.shape) != (B, m, ns, width) or rows.dtype != torch.float32 or rows.device != idx.device:
            raise RuntimeError(f"fused_group_mlp_pool: pre-grouped rows must be fp32 ({B}, {m}, {ns}, {width}) on {idx.device}")
        elif _node(layers, ns) is _FusedMLPBf16:
            rows = None                                   # (fp32 rows are of no use to the bf16 node)
        rows = None if rows is None else rows.contiguous()
    # rows: the grouped rows (B, m, ns, [3+]C) fp32 if the query kernel already emitted them (pn2_ball_query_group)
    # [9]: the caller batches scans with per-scan statistics (inside pointnet2_modules.per_scan_statistics, which single-scan
    # steps of the scene-graph model enter too): the bf16 node then keeps the routes its segment-table form has, so that a
    # batched step stays the arithmetic of its single-scan steps
    group = (xyz, new_xyz, idx, bool(use_xyz), bool(normalize), radius, inv, crowded,
             None if rows is None else rows.detach(), bool(per_scan_caller) or clouds_per_scan is not None)
    if clouds_per_scan is not None and len(clouds_per_scan) > 1:
        if sum(clouds_per_scan) != B:
            raise RuntimeError("fused_group_mlp_pool: clouds_per_scan must sum to the number of clouds")
        node = _node(layers, ns)
        if seg_table_ok(layers, ns, node):
            res = _SegTableMLP.apply(None if feats_rows is None else feats_rows.contiguous(), int(ns), layers, group,
                                     tuple(int(v) * m * ns for v in clouds_per_scan), node, *_params(layers))
            return res[0].view(B, m, -1)
        # (the loop slices the clouds: a whole-batch inverse index is only used, in slices, by lifted first layers)
        res = _SegmentedGroupMLP.apply(None if feats_rows is None else feats_rows.contiguous(), int(ns), layers, group,
                                       tuple(int(v) for v in clouds_per_scan), _node(layers, ns), *_params(layers))
    else:
        res = _node(layers, ns).apply(feats_rows, int(ns), layers, group, *_params(layers))
    return res[0].view(B, m, -1)
