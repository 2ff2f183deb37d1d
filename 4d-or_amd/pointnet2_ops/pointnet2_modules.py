"""Set-abstraction / feature-propagation modules of ``pointnet2_ops`` on MI355X.

API mirror of OPS/pointnet2_modules.py (``build_shared_mlp`` :9-19,
``_PointnetSAModuleBase`` :22-74, ``PointnetSAModuleMSG`` :77-115,
``PointnetSAModule`` :118-146, ``PointnetFPModule`` :149-209): same constructor
arguments, same parameter containers (``mlps.{i}.{0,1,3,4,...}`` Conv2d 1x1 +
BatchNorm2d, so reference ``state_dict``s load strictly), same input/output
shapes.  PointnetSAModuleMSG also reproduces the reference's in-place
``mlp_spec[0] += 3`` side effect on the caller's list (:112-113).

Two execution paths produce the same numbers (tests compare them):

* literal  — the reference's staging: channel-major group -> Conv2d/BN2d/ReLU
  -> max_pool2d.  Used when the fast path is switched off or coordinates
  require gradients.
* rows (default) — MI355X-first layout.  Features stay point-major
  ``(B, N, C)`` between layers; the grouped tensor is written once as
  ``(B*npoint*nsample, 3+C)`` rows by one fused HIP kernel; each 1x1 conv is a
  plain ``rows @ W^T`` GEMM (K contiguous, MFMA friendly), BatchNorm runs on the
  2-D rows tensor with the module's own running statistics, and the
  neighbourhood max is one HIP kernel.  Module outputs are returned as
  ``(B, C, npoint)`` *views* of the rows tensor, so the next module picks the
  rows layout up again without a transpose copy.
"""
from typing import List, Optional, Tuple

import contextlib
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from pointnet2_ops import pointnet2_utils

_FAST_PATH = True


def set_fast_path(enabled: bool) -> bool:
    """Switch the rows path on/off globally (returns the previous setting)."""
    global _FAST_PATH
    prev, _FAST_PATH = _FAST_PATH, bool(enabled)
    return prev


def fast_path_enabled() -> bool:
    return _FAST_PATH


def build_shared_mlp(mlp_spec: List[int], bn: bool = True):
    """[c0, c1, ..., ck] -> Sequential(Conv2d 1x1 (bias iff no bn), [BatchNorm2d], ReLU) * k."""
    stages = []
    for c_in, c_out in zip(mlp_spec[:-1], mlp_spec[1:]):
        stages.append(nn.Conv2d(c_in, c_out, kernel_size=1, bias=not bn))
        if bn:
            stages.append(nn.BatchNorm2d(c_out))
        stages.append(nn.ReLU(True))
    return nn.Sequential(*stages)


def _batch_norm_rows(bn: nn.modules.batchnorm._BatchNorm, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm over the rows of a (P, C) tensor with `bn`'s parameters and the
    bookkeeping of torch's _BatchNorm.forward (momentum / cumulative average,
    num_batches_tracked, batch statistics whenever running stats are absent)."""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    use_batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    keep_running = bn.training or bn.track_running_stats
    return F.batch_norm(x, bn.running_mean if keep_running else None,
                        bn.running_var if keep_running else None,
                        bn.weight, bn.bias, use_batch_stats, momentum, bn.eps)


def shared_mlp_rows(mlp: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Apply a `build_shared_mlp` stack (or any Conv2d-1x1/BN/ReLU sequence) to
    point-major rows x (P, C_in) -> (P, C_out)."""
    for layer in mlp:
        if isinstance(layer, nn.Conv2d):
            w = layer.weight.view(layer.out_channels, layer.in_channels)
            x = F.linear(x, w, layer.bias)
        elif isinstance(layer, nn.modules.batchnorm._BatchNorm):
            x = _batch_norm_rows(layer, x)
        elif isinstance(layer, nn.ReLU):
            x = F.relu(x, inplace=layer.inplace)
        elif isinstance(layer, nn.Sequential):
            x = shared_mlp_rows(layer, x)
        else:
            raise TypeError(f"shared_mlp_rows: unsupported layer {type(layer).__name__}")
    return x


_FUSED_MLP = True


def set_fused_mlp(enabled: bool) -> bool:
    """Switch the fused MFMA shared-MLP kernels on/off (off = torch GEMM/BN/ReLU on rows)."""
    global _FUSED_MLP
    prev, _FUSED_MLP = _FUSED_MLP, bool(enabled)
    return prev


def mlp_rows(mlp: nn.Module, rows: torch.Tensor) -> torch.Tensor:
    """(M, C_in) -> (M, C_out) through a shared MLP, fused kernels when the stack allows."""
    from pointnet2_ops import fused_mlp
    if _FUSED_MLP and fused_mlp.supported(mlp, rows):
        return fused_mlp.fused_shared_mlp(mlp, rows, 0)
    return shared_mlp_rows(mlp, rows)


# ---------------------------------------------------------------------------------- batched scans, per-scan statistics
# The reference trains on ONE scan per step (SGP/main.py:54-56), so every training-mode BatchNorm of the encoders sees the
# clouds of one scan (9 objects resp. 72 pairs).  A block-diagonal batch of S scans (dataset/synthetic.py::collate_scans)
# keeps that arithmetic when the shared MLPs are run per scan: inside `per_scan_statistics`, a stack whose BatchNorms are in
# training mode processes each scan's clouds [c_s, c_{s+1}) as its own call (own batch statistics, running statistics
# updated scan after scan, in order) — everything without statistics (sampling, ball query, grouping geometry) stays batched.
# The row counts per call stay large (>= 10^5), so the kernels keep their efficiency; the price is S times the launches.
class _ScanSegments(threading.local):
    """clouds in the batch -> clouds per scan, PER THREAD (a DataParallel replica or a threaded loader must not see
    another thread's segmentation); only SA stacks whose input has exactly that many clouds are split."""

    def __init__(self):
        self.table = {}
        self.active = 0          # nesting depth of per_scan_statistics (entered for batches of ONE scan too)


_SEG = _ScanSegments()


@contextlib.contextmanager
def per_scan_statistics(*clouds_per_scan):
    """`clouds_per_scan`: one sequence per encoder input of the step, e.g. ([9] * S, [72] * S) for the object and the
    relation encoder of the scene-graph model.  Batches of one scan need no entry."""
    table = _SEG.table
    saved = dict(table)
    _SEG.active += 1
    try:
        for sizes in clouds_per_scan:
            sizes = tuple(int(v) for v in sizes)
            if len(sizes) < 2:
                continue
            total = sum(sizes)
            if table.get(total, sizes) != sizes:
                raise RuntimeError("per_scan_statistics: two inputs with the same number of clouds but different scans")
            table[total] = sizes
        yield
    finally:
        _SEG.active -= 1
        table.clear()
        table.update(saved)


def _trains_batchnorm(mlp: nn.Module) -> bool:
    return any(isinstance(m, nn.modules.batchnorm._BatchNorm) and (m.training or m.running_mean is None)
               for m in mlp.modules())


def sa_scale_rows(grouper, mlp: nn.Module, xyz, new_xyz, feats_rows, idx=None, _whole_batch=False, inv=None,
                  rows=None) -> torch.Tensor:
    """One SA scale on the rows path: group -> shared MLP -> max -> (B, npoint, C_out).
    Ball-query groupers with a fusable MLP run as a single autograd node (gather, MLP, pool and
    the scatter of the feature gradient); anything else goes through forward_rows + mlp_pool_rows."""
    from pointnet2_ops import eval_fused, fused_mlp
    # inference (eval-mode BatchNorm, no gradient recorded): the whole scale as ONE kernel, activations in registers
    plan = eval_fused.applicable(grouper, mlp, xyz, new_xyz, feats_rows)
    if plan is not None:
        if idx is None:
            idx = grouper.query(xyz, new_xyz)
        return eval_fused.sa_scale_eval(plan, grouper, xyz, new_xyz, feats_rows, idx)
    sizes = None if _whole_batch else _SEG.table.get(xyz.size(0))
    if sizes is not None and _trains_batchnorm(mlp):
        if (_FUSED_MLP and isinstance(grouper, pointnet2_utils.QueryAndGroup) and new_xyz is not None
                and (grouper.use_xyz or feats_rows is not None)
                and fused_mlp.supported(mlp, xyz if feats_rows is None else feats_rows, grouper.nsample)):
            if idx is None:
                idx, rows = _query_maybe_fused(grouper, xyz, new_xyz, feats_rows)
            return fused_mlp.fused_group_mlp_pool(mlp, xyz, new_xyz, feats_rows, idx, grouper.use_xyz,
                                                  grouper.normalize_xyz, grouper.radius, clouds_per_scan=sizes, inv=inv,
                                                  crowded=crowded_balls(grouper, xyz.size(1)), rows=rows)
        if _FUSED_MLP and idx is None and not isinstance(grouper, pointnet2_utils.QueryAndGroup):
            # group-all (or any grouper without statistics of its own): group the whole batch once, then ONE call of the
            # stack with the scans' row ranges as a segment table — where the stack's kernels take one
            g = grouper.forward_rows(xyz, new_xyz, feats_rows)
            Bg, npoint, nsample, width = g.shape
            rows = g.reshape(-1, width)
            if fused_mlp.seg_table_supported(mlp, rows, nsample):
                per = npoint * nsample
                return fused_mlp.fused_shared_mlp(mlp, rows, nsample, rows_per_scan=[n * per for n in sizes]
                                                  ).view(Bg, npoint, -1)
            parts = [mlp_pool_rows(mlp, gs) for gs in g.split_with_sizes(sizes)]
            return torch.cat(parts, dim=0)
        split = lambda t: [None] * len(sizes) if t is None else t.split_with_sizes(sizes)
        parts = [sa_scale_rows(grouper, mlp, x, nx, f, i, _whole_batch=True)
                 for x, nx, f, i in zip(split(xyz), split(new_xyz), split(feats_rows), split(idx))]
        return torch.cat(parts, dim=0)
    if (_FUSED_MLP and isinstance(grouper, pointnet2_utils.QueryAndGroup) and new_xyz is not None
            and (grouper.use_xyz or feats_rows is not None)
            and fused_mlp.supported(mlp, xyz if feats_rows is None else feats_rows, grouper.nsample)):
        if idx is None:
            idx, rows = _query_maybe_fused(grouper, xyz, new_xyz, feats_rows)
        return fused_mlp.fused_group_mlp_pool(mlp, xyz, new_xyz, feats_rows, idx, grouper.use_xyz,
                                              grouper.normalize_xyz, grouper.radius, inv=inv,
                                              crowded=crowded_balls(grouper, xyz.size(1)), rows=rows,
                                              per_scan_caller=_SEG.active > 0)
    if idx is not None:
        g = pointnet2_utils.group_concat_rows(xyz, new_xyz, feats_rows, idx, grouper.use_xyz,
                                              grouper.normalize_xyz, grouper.radius, inv=inv)
        return mlp_pool_rows(mlp, g)
    return mlp_pool_rows(mlp, grouper.forward_rows(xyz, new_xyz, feats_rows))


def _query_maybe_fused(grouper, xyz, new_xyz, feats_rows, data_features=False):
    """(idx, rows | None) for a fused-MLP SA scale: the ball query and — where one kernel can do both (first levels: rows of
    at most 16 floats, fp32 arithmetic) — the grouped rows from the same pass (pn2_ball_query_group).
    `data_features`: `feats_rows` is input data without a gradient (a prefetched geometry of a first level): on the bf16
    node the grouped bf16 rows are then produced right here too (pn2_group_concat_rows_bf16) — next to the query, i.e. on
    the prefetch stream, instead of in front of the stack's first GEMM on the critical path."""
    from pointnet2_ops import fused_mlp
    _ext = pointnet2_utils._ext
    if fused_mlp.mlp_dtype() == torch.float32 and grouper.fused_query_ok(xyz, new_xyz, feats_rows):
        return grouper.query_rows(xyz, new_xyz, feats_rows)
    idx = grouper.query(xyz, new_xyz)
    if (data_features and fused_mlp.mlp_dtype() == torch.bfloat16 and feats_rows is not None and xyz.is_cuda
            and not grouper.sample_uniformly and feats_rows.size(2) < 16 and getattr(_ext, "group_concat_rows_bf16", None)):
        with torch.no_grad():
            rows = _ext.group_concat_rows_bf16(xyz, new_xyz, feats_rows.detach().contiguous(), idx, grouper.use_xyz,
                                               grouper.normalize_xyz, grouper.radius)
        return idx, rows
    return idx, None


class RowsSource:
    """Identity of the tensor the pre-grouped `rows` of a geometry were gathered from: storage address, shape, device and
    version counter (bumped by every in-place operation).  `forward` compares it with what it is actually given and
    drops the rows on a mismatch — augmentation, dropout or another tensor between the prefetch and the forward would
    otherwise be ignored silently.  An opaque object on purpose: nested python scalars of a batch are part
    of a captured graph's signature (runtime/graphed_step.py), addresses must not be.

    Two levels: a module-level `sample_and_query(feats_rows=...)` records the (B, N, C) feature rows themselves; a model's
    `precompute_geometry(pointcloud)` slices its own copy of the feature columns, so it records the POINT CLOUD
    (`rows_source(pointcloud)`) and its forward confirms the match (`confirm_rows`) before the levels run."""
    __slots__ = ("ptr", "shape", "version", "device", "confirmed", "storage")

    def __init__(self, t: torch.Tensor):
        self.ptr, self.shape, self.version, self.device = t.data_ptr(), tuple(t.shape), t._version, t.device
        self.confirmed = False
        # the token keeps the source's STORAGE alive: a cloud replaced out of place after the prefetch frees its
        # memory, the caching allocator hands the same address to an equal-shaped new tensor with version 0, and address +
        # shape + version would match it.  With the storage held, the address cannot be recycled while the token lives.
        self.storage = t.untyped_storage()

    def matches(self, t: Optional[torch.Tensor]) -> bool:
        return (t is not None and t.data_ptr() == self.ptr and tuple(t.shape) == self.shape
                and t._version == self.version and t.device == self.device)


def rows_source(t: Optional[torch.Tensor]) -> Optional[RowsSource]:
    return None if t is None else RowsSource(t)


def _capturing() -> bool:
    # inside a stream capture the batch is GraphedTrainStep's static clone of ONE consistent batch (geometry included):
    # addresses differ from the prefetch by construction, the pairing is the capture's own
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def rows_still_valid(geometry, feats_rows: Optional[torch.Tensor]) -> bool:
    """True when the geometry's pre-grouped rows belong to the features this forward was given."""
    tok = geometry.get("rows_src")
    if tok is None:
        return False
    return tok.confirmed or tok.matches(feats_rows) or _capturing()


def confirm_rows(geometry_levels, pointcloud: torch.Tensor):
    """Model level: `geometry_levels` (list of per-level geometry dicts) computed by `precompute_geometry(pointcloud)`.
    Levels whose rows were gathered from exactly this point cloud are marked confirmed; otherwise the rows are dropped
    (the level groups the features it is given)."""
    out = []
    for g in geometry_levels:
        tok = None if g is None else g.get("rows_src")
        if tok is not None and not tok.confirmed:
            if tok.matches(pointcloud) or _capturing():
                ok = RowsSource(pointcloud)
                ok.confirmed = True
                g = dict(g, rows_src=ok)
            else:
                g = dict(g, rows=None, rows_src=None)
        out.append(g)
    return out


def crowded_balls(grouper, n_src: int) -> bool:
    """Density rule shared with the ball query's cell-list switch (csrc/ball_query.hip), inverted: with N r^3 > 4 nsample
    nearly every slot of a neighbourhood is a genuine hit and each point is gathered many times — the regime where the
    feature-gradient scatter pays for an inverse index (csrc/group_csr.hip: 0.47 -> 0.15 ms at the headline SA2 level);
    with sparse balls most slots repeat the first hit, which the atomic kernel pre-reduces in registers."""
    return n_src * float(grouper.radius) ** 3 > 4.0 * grouper.nsample


def build_inverse_indices(groupers, idx_list, n_src: int):
    """Inverse neighbourhood index (ptr, refs) for every crowded ball-query scale, None for the others.  The result
    travels NEXT to the indices (geometry["inv"]): explicit tensors are seen by record_stream, by a graph capture's static
    copies and by _check_geometry, which a Python attribute on the idx tensor was not."""
    build = getattr(pointnet2_utils._ext, "group_inverse_index", None)
    out = []
    for g, idx in zip(groupers, idx_list):
        ok = (build is not None and idx is not None and isinstance(g, pointnet2_utils.QueryAndGroup)
              and crowded_balls(g, n_src))
        out.append(tuple(build(idx, n_src)) if ok else None)
    return out


def mlp_pool_rows(mlp: nn.Module, grouped: torch.Tensor) -> torch.Tensor:
    """grouped (B, npoint, nsample, C_in) -> (B, npoint, C_out): shared MLP then max over nsample."""
    from pointnet2_ops import fused_mlp
    B, npoint, nsample, width = grouped.shape
    rows = grouped.reshape(-1, width)
    if _FUSED_MLP and fused_mlp.supported(mlp, rows, nsample):
        return fused_mlp.fused_shared_mlp(mlp, rows, nsample).view(B, npoint, -1)
    h = shared_mlp_rows(mlp, rows)
    return pointnet2_utils.rows_max(h.view(B * npoint, nsample, -1)).view(B, npoint, -1)


def _rows_path_ok(xyz: torch.Tensor, features: Optional[torch.Tensor]) -> bool:
    if not _FAST_PATH or not getattr(pointnet2_utils._ext, "HAS_ROWS", False):
        return False
    if xyz.requires_grad:
        return False  # coordinate gradients only exist on the literal path
    return xyz.dtype == torch.float32 and (features is None or features.dtype == torch.float32)


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    # -- centre selection (shared by both paths) ------------------------------
    def _sample(self, xyz: torch.Tensor) -> Optional[torch.Tensor]:
        if self.npoint is None:
            return None
        return pointnet2_utils.sample_centres(xyz, self.npoint)[1]

    def sample_and_query(self, xyz: torch.Tensor, inverse_index: bool = False, feats_rows: Optional[torch.Tensor] = None):
        """The data-only part of the module (no parameters, no features): sampled centres and the ball-query
        neighbourhoods of every scale.  A training loop that already holds the next clouds can run this on a side
        stream while the current ones train and pass the result as `geometry=` (identical results).
        `inverse_index`: the features of this level will need a gradient — also build the inverse of crowded
        neighbourhood indices, which turns the backward's atomic scatter into a per-point sum."""
        new_xyz = self._sample(xyz)
        idx, rows = [], []
        for g in self.groupers:
            if new_xyz is None or not isinstance(g, pointnet2_utils.QueryAndGroup):
                idx.append(None), rows.append(None)
            elif feats_rows is not None:
                # `feats_rows` (B,N,C): the level's input features are DATA (colours / masks of the input cloud, no
                # gradient) -> the grouped rows can be produced right here, by the query kernel itself where it covers them
                i, r = _query_maybe_fused(g, xyz, new_xyz, feats_rows, data_features=True)
                idx.append(i), rows.append(r)
            else:
                idx.append(g.query(xyz, new_xyz)), rows.append(None)
        inv = build_inverse_indices(self.groupers, idx, xyz.size(1)) if inverse_index else [None] * len(idx)
        return {"new_xyz": new_xyz, "idx": idx, "inv": inv, "rows": rows, "n_src": xyz.size(1),
                "rows_src": rows_source(feats_rows) if any(r is not None for r in rows) else None}

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor], geometry=None
                ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """xyz (B,N,3), features (B,C,N)|None ->
        (new_xyz (B,npoint,3)|None, new_features (B, sum_k mlps[k][-1], npoint)).
        `geometry` = sample_and_query(xyz) computed earlier (optional; rows path only)."""
        if geometry is not None and _rows_path_ok(xyz, features):
            self._check_geometry(xyz, geometry)
            return geometry["new_xyz"], self._forward_rows(xyz, geometry["new_xyz"], features, geometry["idx"],
                                                           geometry.get("inv"), geometry.get("rows"), geometry)
        new_xyz = self._sample(xyz)
        if _rows_path_ok(xyz, features):
            return new_xyz, self._forward_rows(xyz, new_xyz, features)

        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            g = mlp(grouper(xyz, new_xyz, features))            # (B, C_out, npoint, nsample)
            g = F.max_pool2d(g, kernel_size=[1, g.size(3)])     # (B, C_out, npoint, 1)
            pooled.append(g.squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1)

    def _check_geometry(self, xyz, geometry):
        """Cheap invariants of a prefetched `sample_and_query` result: it must belong to a batch of this shape on this
        device (a stale or re-ordered geometry would otherwise gather out of range or silently mix clouds)."""
        new_xyz, idx = geometry["new_xyz"], geometry["idx"]
        B = xyz.size(0)
        if len(idx) != len(self.groupers):
            raise RuntimeError("geometry: one ball-query index tensor per scale expected")
        if self.npoint is not None:
            if new_xyz is None or tuple(new_xyz.shape) != (B, self.npoint, 3) or new_xyz.device != xyz.device:
                raise RuntimeError(f"geometry: new_xyz must be ({B}, {self.npoint}, 3) on {xyz.device}")
        if "n_src" in geometry and geometry["n_src"] != xyz.size(1):
            raise RuntimeError(f"geometry was computed for clouds of {geometry['n_src']} points, got {xyz.size(1)}")
        inv = geometry.get("inv")
        if inv is not None:
            if len(inv) != len(idx):
                raise RuntimeError("geometry: one inverse index (or None) per scale expected")
            for i, v in zip(idx, inv):
                if v is not None and (i is None or v[0].numel() != B * xyz.size(1) + 1 or v[1].numel() != i.numel()
                                      or v[0].device != xyz.device):
                    raise RuntimeError("geometry: inverse index does not belong to these neighbourhoods")
        for g, i in zip(self.groupers, idx):
            if i is None:
                continue
            if (i.dtype != torch.int32 or i.device != xyz.device or i.dim() != 3
                    or tuple(i.shape[:2]) != (B, self.npoint) or i.size(2) != g.nsample):
                raise RuntimeError(f"geometry: idx must be int32 ({B}, {self.npoint}, {g.nsample}) on {xyz.device}")
        rows = geometry.get("rows")
        if rows is not None:
            if len(rows) != len(idx):
                raise RuntimeError("geometry: one pre-grouped rows tensor (or None) per scale expected")
            for i, r in zip(idx, rows):
                if r is not None and (i is None or r.device != xyz.device or r.dim() < 2
                                      or r.numel() != i.numel() * r.size(-1)):
                    raise RuntimeError("geometry: pre-grouped rows do not belong to these neighbourhoods")

    def _forward_rows(self, xyz, new_xyz, features, idx=None, inv=None, rows=None, geometry=None):
        feats_rows = pointnet2_utils.as_rows(features)
        if rows is not None and (geometry is None or not rows_still_valid(geometry, feats_rows)):
            rows = None          # gathered from another (or since modified) feature tensor: group the given one instead
        B = xyz.size(0)
        pooled = []
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            pooled.append(sa_scale_rows(grouper, mlp, xyz, new_xyz, feats_rows,
                                        idx=None if idx is None else idx[k],
                                        inv=None if inv is None else inv[k],
                                        rows=None if rows is None else rows[k]))     # (B, npoint, C_out)
        rows = pooled[0] if len(pooled) == 1 else torch.cat(pooled, dim=2)
        return pointnet2_utils.rows_to_channels(rows)           # (B, sum C_out, npoint) view


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale-grouping set abstraction: one (radius, nsample, mlp) per scale."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, normalize_xyz=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            if npoint is not None:
                self.groupers.append(pointnet2_utils.QueryAndGroup(
                    radius, nsample, use_xyz=use_xyz, normalize_xyz=normalize_xyz))
            else:
                self.groupers.append(pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3  # in place, like the reference: callers observe the mutation
            self.mlps.append(build_shared_mlp(spec, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction; npoint=None groups the whole cloud."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True,
                 normalize_xyz=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample],
                         bn=bn, use_xyz=use_xyz, normalize_xyz=normalize_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance 3-NN interpolation + shared MLP."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = build_shared_mlp(mlp, bn=bn)

    @staticmethod
    def _weights(unknown, known):
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        recip = 1.0 / (dist + 1e-8)
        return idx, recip / torch.sum(recip, dim=2, keepdim=True)

    def interpolation(self, unknown, known):
        """Data-only part: (idx (B,n,3) int32, weight (B,n,3), inverse of idx | None) of the inverse-distance 3-NN
        interpolation.  The inverse index turns the backward's three atomicAdds per gradient element into a per-point sum
        (csrc/interpolate.hip); it is data like idx, so a prefetched geometry carries it."""
        idx, weight = self._weights(unknown, known)
        return idx, weight, pointnet2_utils.interp_inverse(idx, known.size(1))

    def forward(self, unknown, known, unknow_feats, known_feats, interp=None):
        """unknown (B,n,3), known (B,m,3)|None, unknow_feats (B,C1,n)|None,
        known_feats (B,C2,m) -> (B, mlp[-1], n).  `interp` = interpolation(unknown, known) computed earlier."""
        if _rows_path_ok(unknown, known_feats) and (unknow_feats is None or unknow_feats.dtype == torch.float32):
            return self._forward_rows(unknown, known, unknow_feats, known_feats, interp)

        if known is not None:
            idx, weight = self._weights(unknown, known)
            spread = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            spread = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        stacked = spread if unknow_feats is None else torch.cat([spread, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)

    def _forward_rows(self, unknown, known, unknow_feats, known_feats, interp=None):
        B, n = unknown.size(0), unknown.size(1)
        known_rows = pointnet2_utils.as_rows(known_feats)               # (B,m,C2)
        if known is not None:
            idx, weight, inv = (tuple(interp) + (None,))[:3] if interp is not None else self._weights(unknown, known) + (None,)
            if inv is None and known_rows.requires_grad:
                inv = pointnet2_utils.interp_inverse(idx, known.size(1))      # (not prefetched: built here, one launch)
            spread = pointnet2_utils.interpolate_concat_rows(
                known_rows, idx, weight, None if unknow_feats is None else pointnet2_utils.as_rows(unknow_feats), inv)
        else:
            spread = known_rows.expand(B, n, known_rows.size(2))
            if unknow_feats is not None:
                spread = torch.cat([spread, pointnet2_utils.as_rows(unknow_feats)], dim=2)
        h = mlp_rows(self.mlp, spread.reshape(B * n, -1))
        return pointnet2_utils.rows_to_channels(h.view(B, n, -1))
