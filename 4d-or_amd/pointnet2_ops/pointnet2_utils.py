"""Operator layer of ``pointnet2_ops`` on MI355X.

API mirror of the reference's OPS/pointnet2_utils.py (OPS =
scene_graph_prediction/pointnet2_dir/pointnet2_ops_lib/pointnet2_ops): the six
autograd functions and their functional aliases (``furthest_point_sample``,
``gather_operation``, ``three_nn``, ``three_interpolate``,
``grouping_operation``, ``ball_query``) and the ``QueryAndGroup`` / ``GroupAll``
modules keep their names, argument order, tensor layouts and differentiability
(:36-280, :283-383).  ``QueryAndGroup`` additionally accepts the keyword
arguments of the Group-Free-3D copy (GF3D/pointnet2/pointnet2_utils.py:301-371:
``ret_grouped_xyz``, ``normalize_xyz``, ``sample_uniformly``, ``ret_unique_cnt``).

Everything numeric happens in ``_ext`` (libpn2_hip.so).  On top of the literal
API this file adds the *point-major* ("rows") operators the SA/FP modules use
on their fast path: the grouped tensor is produced directly as
``(B, npoint, nsample, 3+C)`` rows by one kernel instead of two channel-major
gathers, an in-place subtract and a ``torch.cat`` (OPS/pointnet2_utils.py:317-328
stages four passes over the largest tensor of the network).
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from pointnet2_ops import _ext

__all__ = [
    "FurthestPointSampling", "furthest_point_sample", "sample_centres", "GatherOperation", "gather_operation",
    "ThreeNN", "three_nn", "ThreeInterpolate", "three_interpolate", "GroupingOperation",
    "grouping_operation", "BallQuery", "ball_query", "QueryAndGroup", "GroupAll",
    "group_concat_rows", "rows_max", "three_interpolate_rows", "interpolate_concat_rows", "interp_inverse", "as_rows",
    "rows_to_channels",
]


def _dense(features):
    """Feature inputs of the literal ops may be the (B,C,N) *views* of point-major rows that the fast-path modules
    return (pointnet2_modules._forward_rows); the native ops need dense channel-major memory like the reference's
    CHECK_CONTIGUOUS (EXT/include/utils.h:5-10).  No copy when the tensor is already contiguous."""
    return features if features.is_contiguous() else features.contiguous()


def _fp32(t):
    """GroupingOperation runs in fp32 even under autocast
    (reference: @custom_fwd(cast_inputs=torch.float32), OPS/pointnet2_utils.py:198)."""
    return t if t.dtype == torch.float32 else t.float()


# ----------------------------------------------------------------- sampling
class FurthestPointSampling(Function):
    """xyz (B,N,3) -> (B,npoint) int32 indices; not differentiable."""

    @staticmethod
    def forward(ctx, xyz, npoint, ordered=False):
        """`ordered`: the clouds are believed to be in farthest-point order (centres of the SA level above): the HIP
        backend verifies that on the device and skips the sampling rounds where it holds — identical results."""
        if ordered and getattr(_ext, "FPS_ORDERED", False):
            sel = _ext.furthest_point_sampling(xyz, npoint, ordered=True)
        else:
            sel = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(sel)
        return sel

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


def sample_centres(xyz, npoint, inds=None):
    """The centre selection of an SA level (pointnet2_modules.py:38-48): -> (inds (B,npoint) i32, new_xyz (B,npoint,3)).
    Centres sampled HERE come out in the order the sampling picked them and are tagged as such; a level that samples from
    tagged centres tells the kernel (`ordered`: sampling a sampling order returns 0 .. npoint-1 unless a tie or a
    degenerate round intervenes — verified on the device, identical results; include/pn2_hip.h).  Caller-provided `inds`
    carry no such promise."""
    sampled = inds is None
    if sampled:
        inds = furthest_point_sample(xyz, npoint, bool(getattr(xyz, "_pn2_fps_order", False)))
    new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    if sampled:
        new_xyz._pn2_fps_order = True
    return inds, new_xyz


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_src = features.size(2)
        return _ext.gather_points(_dense(features), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


gather_operation = GatherOperation.apply


# ------------------------------------------------------------ interpolation
class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) euclidean, idx (B,n,3) int32)."""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m_src = features.size(2)
        return _ext.three_interpolate(_dense(features), idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_src)
        return g, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


# ----------------------------------------------------------------- grouping
class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""

    @staticmethod
    def forward(ctx, features, idx):
        features = _dense(_fp32(features))
        ctx.save_for_backward(idx)
        ctx.n_src = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        g = _ext.group_points_grad(_fp32(grad_out).contiguous(), idx, ctx.n_src)
        return g, torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> (B,npoint,nsample) int32.
    Note the python argument order differs from the native one
    (OPS/pointnet2_utils.py:249,269 vs EXT/include/ball_query.h:4)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        idx = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


# ---------------------------------------------------- point-major ("rows") ops
def as_rows(features: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """(B,C,N) -> contiguous (B,N,C).  Zero-copy when `features` is already a
    transposed view of a rows tensor (which is what the fast-path modules emit)."""
    if features is None:
        return None
    return features.transpose(1, 2).contiguous()


def rows_to_channels(rows: torch.Tensor) -> torch.Tensor:
    """(B,N,C) rows -> (B,C,N) view with the reference's logical layout."""
    return rows.transpose(1, 2)


class _GroupConcatRows(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius, inv=None):
        ctx.save_for_backward(idx)
        ctx.n_src = xyz.size(1)
        ctx.c = 0 if feats_rows is None else feats_rows.size(2)
        ctx.col0 = 3 if use_xyz else 0
        ctx.inv = inv                     # (ptr, refs) of _ext.group_inverse_index, or None
        return _ext.group_concat_rows(xyz, new_xyz, feats_rows, idx, use_xyz, normalize, radius)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        g = None
        if ctx.c and ctx.needs_input_grad[2]:
            if ctx.inv is not None:      # prefetched inverse index: per-point sum, no atomics (csrc/group_csr.hip)
                g = _ext.group_rows_grad_csr(grad_out.contiguous(), ctx.inv, ctx.n_src, ctx.c, ctx.col0)
            else:
                g = _ext.group_rows_grad(grad_out.contiguous(), idx, ctx.n_src, ctx.c, ctx.col0)
        return None, None, g, None, None, None, None, None


def group_concat_rows(xyz, new_xyz, feats_rows, idx, use_xyz=True, normalize=False, radius=None, inv=None):
    """Fused QueryAndGroup tail: -> (B,npoint,nsample,[3+]C) rows.  Gradients flow
    to `feats_rows` only (callers route coordinates that need grad to the literal path).
    `inv`: inverse of `idx` (_ext.group_inverse_index) -> atomic-free feature gradient."""
    return _GroupConcatRows.apply(xyz, new_xyz, feats_rows, idx, bool(use_xyz), bool(normalize), radius, inv)


class _RowsMax(Function):
    @staticmethod
    def forward(ctx, x):
        out, arg = _ext.rows_max(x)
        ctx.save_for_backward(arg)
        ctx.ns = x.size(1)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        return _ext.rows_max_grad(grad_out.contiguous(), arg, ctx.ns)


def rows_max(x: torch.Tensor) -> torch.Tensor:
    """x (R,ns,C) -> (R,C): the max over each neighbourhood
    (F.max_pool2d(kernel=[1,ns]) of OPS/pointnet2_modules.py:67-70)."""
    return _RowsMax.apply(x.contiguous())[0]


class _ThreeInterpolateRows(Function):
    @staticmethod
    def forward(ctx, feats_rows, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m_src, ctx.c = feats_rows.size(1), feats_rows.size(2)
        return _ext.three_interpolate_rows(feats_rows, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = _ext.three_interpolate_rows_grad(grad_out.contiguous(), idx, weight, ctx.m_src, ctx.c)
        return g, None, None


def three_interpolate_rows(feats_rows, idx, weight):
    """feats_rows (B,m,C), idx/weight (B,n,3) -> (B,n,C)."""
    return _ThreeInterpolateRows.apply(feats_rows.contiguous(), idx, weight)


class _InterpolateConcatRows(Function):
    """PointnetFPModule's `cat([three_interpolate(known_feats), unknow_feats])` (OPS/pointnet2_modules.py:197-204) on rows
    as ONE node: the interpolation writes its columns of the (B, n, C2 + C1) result directly (no intermediate + concat
    copy), the backward reads its columns of the incoming gradient in place and — given the inverse of `idx`
    (`interp_inverse`) — sums them per known point without atomics."""

    @staticmethod
    def forward(ctx, known_rows, idx, weight, unknown_rows, inv):
        B, m, C2 = known_rows.shape
        n = idx.size(1)
        C1 = 0 if unknown_rows is None else unknown_rows.size(2)
        out = torch.empty(B, n, C2 + C1, dtype=known_rows.dtype, device=known_rows.device)
        _ext.three_interpolate_rows(known_rows, idx, weight, out=out, col0=0)
        if C1:
            out[:, :, C2:].copy_(unknown_rows)
        ctx.save_for_backward(idx, weight)
        ctx.inv, ctx.m, ctx.c2, ctx.c1 = inv, m, C2, C1
        return out

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        g = g.contiguous()
        gk = None
        if ctx.needs_input_grad[0]:
            gk = _ext.three_interpolate_rows_grad(g, idx, weight, ctx.m, ctx.c2, col0=0, inv=ctx.inv)
        gu = g[:, :, ctx.c2:] if (ctx.c1 and ctx.needs_input_grad[3]) else None
        return gk, None, None, gu, None


def interp_inverse(idx, m):
    """Inverse of a 3-NN index (B,n,3) into m known points — (ptr, refs) of group_inverse_index — where the backend
    builds one (data only: goes next to idx / weight in a prefetched geometry)."""
    build = getattr(_ext, "group_inverse_index", None)
    if build is None or not idx.is_cuda:
        return None
    return tuple(build(idx, int(m)))


def interpolate_concat_rows(known_rows, idx, weight, unknown_rows=None, inv=None):
    """known_rows (B,m,C2), idx / weight (B,n,3), unknown_rows (B,n,C1) | None -> (B,n,C2+C1)."""
    if getattr(_ext, "HAS_ROWS", False) and known_rows.is_cuda and known_rows.dtype == torch.float32 \
            and (unknown_rows is None or unknown_rows.dtype == torch.float32):
        return _InterpolateConcatRows.apply(known_rows.contiguous(), idx, weight, unknown_rows, inv)
    spread = three_interpolate_rows(known_rows, idx, weight)
    return spread if unknown_rows is None else torch.cat([spread, unknown_rows], dim=2)


# ------------------------------------------------------------------- modules
class QueryAndGroup(nn.Module):
    """Ball query + grouping around each centre.

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) | None)
        -> (B, 3+C, npoint, nsample)           [+ grouped_xyz when ret_grouped_xyz]
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False,
                 normalize_xyz=False, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly           # GF3D/pointnet2/pointnet2_utils.py:309-310
        self.last_unique_cnt = None

    def query(self, xyz, new_xyz):
        """Neighbourhood indices.  With `sample_uniformly` (GF3D :327-336, a host loop of torch.unique + torch.randint
        per region in the reference) the padded tail of every row is redrawn uniformly from the row's unique hits by
        one device kernel; the draws come from a counter-based generator seeded from torch's host generator (so
        torch.manual_seed makes them reproducible) — same distribution as the reference, not the same stream."""
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if self.sample_uniformly:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)))
            self.last_unique_cnt = _ext.ball_query_unique_resample(idx, seed, want_cnt=True)
        return idx

    def fused_query_ok(self, xyz, new_xyz, feats_rows) -> bool:
        """The ball query of this grouper and the grouping of `feats_rows` can run as ONE kernel (pn2_ball_query_group:
        fp32 rows of at most 16 floats, nsample <= 256, no sample_uniformly redraw between query and grouping) and the
        shape is one where the slab cell lists pay (crowded balls in large clouds / clouds of >= 2048 points)."""
        fn = getattr(_ext, "ball_query_group", None)
        if fn is None or self.sample_uniformly or new_xyz is None or not (self.use_xyz or feats_rows is not None):
            return False
        if not xyz.is_cuda or xyz.dtype != torch.float32 or (feats_rows is not None and feats_rows.dtype != torch.float32):
            return False
        B, N, m = xyz.size(0), xyz.size(1), new_xyz.size(1)
        C = 0 if feats_rows is None else feats_rows.size(2)
        return bool(_ext.ball_query_group_supported(B, N, m, self.radius, self.nsample, C, self.use_xyz)
                    and _ext.ball_query_group_pays(B, N, m, self.radius, self.nsample))

    def query_rows(self, xyz, new_xyz, feats_rows=None):
        """(idx (B,npoint,nsample) i32, rows (B,npoint,nsample,[3+]C) f32) WITHOUT autograd — the fused query + grouping
        kernel where `fused_query_ok`, else the two kernels.  Identical results either way.  For callers that handle the
        feature gradient themselves (fused_mlp: scatter through idx in its own backward) or need none (input colours)."""
        with torch.no_grad():
            feats = None if feats_rows is None else feats_rows.detach().contiguous()
            if self.fused_query_ok(xyz, new_xyz, feats):
                return _ext.ball_query_group(new_xyz, xyz, feats, self.radius, self.nsample, self.use_xyz, self.normalize_xyz)
            idx = self.query(xyz, new_xyz)
            return idx, _ext.group_concat_rows(xyz, new_xyz, feats, idx, self.use_xyz, self.normalize_xyz, self.radius)

    def forward(self, xyz, new_xyz, features=None):
        idx = self.query(xyz, new_xyz)
        centres = new_xyz.transpose(1, 2).unsqueeze(-1)
        rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        rel -= centres
        if self.normalize_xyz:
            rel /= self.radius
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            grouped = rel
        else:
            picked = grouping_operation(features, idx)
            grouped = torch.cat([rel, picked], dim=1) if self.use_xyz else picked
        ret = [grouped]
        if self.ret_grouped_xyz:
            ret.append(rel)
        if self.ret_unique_cnt:
            ret.append(self.last_unique_cnt)          # (B, npoint) f32 on the device (the reference builds it on the host)
        return ret[0] if len(ret) == 1 else tuple(ret)

    def forward_rows(self, xyz, new_xyz, feats_rows=None):
        """Fast path: (B,npoint,nsample,[3+]C) rows in one fused kernel."""
        if feats_rows is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        idx = self.query(xyz, new_xyz)
        return group_concat_rows(xyz, new_xyz, feats_rows, idx, self.use_xyz,
                                 self.normalize_xyz, self.radius)


class GroupAll(nn.Module):
    """The whole cloud as a single group: -> (B, 3+C, 1, N)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        whole = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            grouped = whole
        else:
            f = features.unsqueeze(2)
            grouped = torch.cat([whole, f], dim=1) if self.use_xyz else f
        return (grouped, whole) if self.ret_grouped_xyz else grouped

    def forward_rows(self, xyz, new_xyz, feats_rows=None):
        """-> (B,1,N,[3+]C) rows."""
        if feats_rows is None:
            rows = xyz
        else:
            rows = torch.cat([xyz, feats_rows], dim=2) if self.use_xyz else feats_rows
        return rows.unsqueeze(1)
