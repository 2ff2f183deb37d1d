"""SA / FP modules with the Group-Free-3D API (keyword-only constructors, SA
returns the FPS indices).  Mirrors GF3D/pointnet2/pointnet2_modules.py:162-269
(``PointnetSAModuleVotes``) and :354-414 (``PointnetFPModule``); same parameter
names (``mlp_module.layer{i}.conv`` / ``.bn.bn``; ``mlp.layer{i}...``).  Both run
on the rows fast path of ``pointnet2_ops`` by default and fall back to the
reference's channel-major staging when it is switched off."""
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from pointnet2_ops import pointnet2_modules as _pm
from pointnet2_ops import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True, pooling: str = "max",
                 sigma: float = None, normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False):
        super().__init__()
        if pooling not in ("max", "avg", "rbf"):
            raise ValueError(f"unknown pooling {pooling!r}")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling, self.use_xyz = pooling, use_xyz
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True, normalize_xyz=normalize_xyz,
                sample_uniformly=sample_uniformly, ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        if use_xyz and len(mlp) > 0:
            mlp[0] += 3                      # in place like the reference (:205-207)
        self.mlp_module = pt_utils.SharedMLP(mlp, bn=bn)

    def sample_and_query(self, xyz: torch.Tensor, inds: torch.Tensor = None, inverse_index: bool = False,
                         feats_rows: torch.Tensor = None):
        """The data-only part of the module (no parameters, no features): FPS indices, sampled centres
        and ball-query neighbourhoods.  Lets a pipeline compute the geometry of the NEXT batch on a side
        stream while the current batch trains (see Pointnet2Backbone.precompute_geometry).
        `inverse_index`: this level's features will need a gradient — also build the inverse of a crowded
        neighbourhood index (per-point sum instead of atomics in the backward, csrc/group_csr.hip)."""
        inds, new_xyz = pointnet2_utils.sample_centres(xyz, self.npoint, inds)
        # `feats_rows` (B,N,C): this level's input features are data (input colours: no gradient) -> the query kernel emits
        # the grouped rows as well where it covers them (pn2_ball_query_group), next to idx
        rows = None
        if feats_rows is not None and self.pooling == "max":
            idx, rows = _pm._query_maybe_fused(self.grouper, xyz, new_xyz, feats_rows, data_features=True)
        else:
            idx = self.grouper.query(xyz, new_xyz)
        inv = _pm.build_inverse_indices([self.grouper], [idx], xyz.size(1))[0] if inverse_index else None
        # sample_uniformly / ret_unique_cnt: the count belongs to THIS query (the grouper's last_unique_cnt is overwritten
        # by the next call — a geometry computed ahead of time must carry its own)
        return {"inds": inds, "new_xyz": new_xyz, "idx": idx, "inv": inv, "rows": rows, "n_src": xyz.size(1),
                "rows_src": _pm.rows_source(feats_rows) if rows is not None else None,
                "unique_cnt": getattr(self.grouper, "last_unique_cnt", None) if self.ret_unique_cnt else None}

    def _check_geometry(self, xyz, geometry):
        """A prefetched sample_and_query() result must belong to a batch of this shape on this device."""
        B, dev = xyz.size(0), xyz.device
        nx, idx, inds = geometry["new_xyz"], geometry["idx"], geometry["inds"]
        ok = (tuple(nx.shape) == (B, self.npoint, 3) and nx.device == dev
              and idx.dtype == torch.int32 and tuple(idx.shape) == (B, self.npoint, self.nsample) and idx.device == dev
              and tuple(inds.shape) == (B, self.npoint) and geometry.get("n_src", xyz.size(1)) == xyz.size(1))
        if not ok:
            raise RuntimeError(f"geometry does not match this batch: expected new_xyz ({B},{self.npoint},3), "
                               f"idx int32 ({B},{self.npoint},{self.nsample}) for clouds of {xyz.size(1)} points on {dev}")

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None, geometry=None):
        """xyz (B,N,3), features (B,C,N) -> (new_xyz (B,npoint,3), new_features (B,C',npoint), inds (B,npoint)).
        `geometry` = result of sample_and_query(xyz) computed earlier (optional)."""
        if geometry is not None and self.pooling == "max" and _pm._rows_path_ok(xyz, features):
            self._check_geometry(xyz, geometry)
            feats_rows = pointnet2_utils.as_rows(features)
            pre = geometry.get("rows")
            if pre is not None and (not _pm.rows_still_valid(geometry, feats_rows)
                                    or pre.numel() != geometry["idx"].numel() * pre.size(-1)):
                pre = None       # gathered from another (or since modified) feature tensor: group the given one instead
            rows = _pm.sa_scale_rows(self.grouper, self.mlp_module, xyz, geometry["new_xyz"],
                                     feats_rows, idx=geometry["idx"], inv=geometry.get("inv"), rows=pre)
            out = (geometry["new_xyz"], pointnet2_utils.rows_to_channels(rows), geometry["inds"])
            return out + (geometry.get("unique_cnt"),) if self.ret_unique_cnt else out
        if inds is not None:
            assert inds.shape[1] == self.npoint
        new_xyz = None
        if self.npoint is not None:
            inds, new_xyz = pointnet2_utils.sample_centres(xyz, self.npoint, inds)
        elif inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)

        if self.npoint is not None and _pm._rows_path_ok(xyz, features):
            feats_rows = pointnet2_utils.as_rows(features)
            if self.pooling == "max":                # (the ball query runs inside: fused with the grouping where it can be)
                rows = _pm.sa_scale_rows(self.grouper, self.mlp_module, xyz, new_xyz, feats_rows, idx=None)
            else:
                rows = self._pool_rows(xyz, new_xyz, feats_rows, self.grouper.query(xyz, new_xyz))
            out = (new_xyz, pointnet2_utils.rows_to_channels(rows), inds)
            return out + (self.grouper.last_unique_cnt,) if self.ret_unique_cnt else out

        unique_cnt = None
        if self.ret_unique_cnt:
            grouped, grouped_xyz, unique_cnt = self.grouper(xyz, new_xyz, features)
        else:
            grouped, grouped_xyz = self.grouper(xyz, new_xyz, features)
        h = self.mlp_module(grouped)                                   # (B, C', npoint, nsample)
        if self.pooling == "max":
            h = F.max_pool2d(h, kernel_size=[1, h.size(3)])
        elif self.pooling == "avg":
            h = F.avg_pool2d(h, kernel_size=[1, h.size(3)])
        else:  # rbf-weighted mean over the neighbourhood (:244-248)
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            h = torch.sum(h * rbf.unsqueeze(1), -1, keepdim=True) / float(self.nsample)
        if self.ret_unique_cnt:
            return new_xyz, h.squeeze(-1), inds, unique_cnt
        return new_xyz, h.squeeze(-1), inds

    def _pool_rows(self, xyz, new_xyz, feats_rows, idx):
        """avg / rbf pooling (:241-248) on the rows path: fused gather kernel -> MFMA shared MLP on point-major rows ->
        (weighted) mean over each neighbourhood; the rbf weights come from the same (optionally radius-normalised)
        relative coordinates the grouper emits as `grouped_xyz`."""
        g = pointnet2_utils.group_concat_rows(xyz, new_xyz, feats_rows, idx, self.use_xyz, self.normalize_xyz, self.radius)
        B, m, ns, W = g.shape
        h = _pm.mlp_rows(self.mlp_module, g.reshape(-1, W)).view(B, m, ns, -1)
        if self.pooling == "avg":
            return h.mean(dim=2)
        rel = g[..., :3] if self.use_xyz else pointnet2_utils.group_concat_rows(
            xyz, new_xyz, None, idx, True, self.normalize_xyz, self.radius)
        rbf = torch.exp(-1 * rel.pow(2).sum(-1) / (self.sigma ** 2) / 2)          # (B, m, ns)
        return torch.sum(h * rbf.unsqueeze(-1), dim=2) / float(self.nsample)


class PointnetFPModule(_pm.PointnetFPModule):
    """Keyword-only variant: ``PointnetFPModule(mlp=[...], bn=True)``; parameters
    live under ``mlp.layer{i}`` (GF3D SharedMLP naming)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        nn.Module.__init__(self)
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)
