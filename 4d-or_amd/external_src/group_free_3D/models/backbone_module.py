"""PointNet++ SA/FP backbone over whole fused OR scans (the repo's only SA+FP
stack; BASELINE config 2 "full SA/FP stack").  Mirrors
GF3D/models/backbone_module.py:12-129: four set-abstraction levels
(2048/0.2/64, 1024/0.4/32, 512/0.8/16, 256/1.2/16, radius-normalised local xyz)
and two feature-propagation levels; same constructor, same ``end_points`` keys,
same parameter names."""
import contextlib

import torch
from typing import Optional
import torch.nn as nn

from pointnet2_ops import pointnet2_utils
from pointnet2_ops.pointnet2_modules import _capturing as capturing, confirm_rows, rows_source
from external_src.group_free_3D.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0, width=1, depth=2):
        super().__init__()
        self.depth, self.width = depth, width
        w = width
        levels = [  # (npoint, radius, nsample, c_in, hidden, c_out)
            (2048, 0.2, 64, input_feature_dim, 64 * w, 128 * w),
            (1024, 0.4, 32, 128 * w, 128 * w, 256 * w),
            (512, 0.8, 16, 256 * w, 128 * w, 256 * w),
            (256, 1.2, 16, 256 * w, 128 * w, 256 * w),
        ]
        for i, (npoint, radius, nsample, c_in, hid, c_out) in enumerate(levels, start=1):
            setattr(self, f"sa{i}", PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample,
                mlp=[c_in] + [hid] * depth + [c_out], use_xyz=True, normalize_xyz=True))
        self.fp1 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, 256 * w])
        self.fp2 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, 288])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        # (B, C, N) as a transposed VIEW of point-major rows: the rows path takes the rows back without a copy (the
        # reference's .transpose(1, 2).contiguous() followed by as_rows was two passes over the colours per step)
        features = pc[..., 3:].contiguous().transpose(1, 2) if pc.size(-1) > 3 else None
        return xyz, features

    def precompute_geometry(self, pointcloud: torch.Tensor, inference: Optional[bool] = None):
        """`inference` (default: eval mode AND the caller records no gradient): the geometry will feed a forward without a
        backward — no inverse neighbourhood indices, no pre-grouped rows (see _precompute_geometry)."""
        if inference is None:
            inference = not self.training and not torch.is_grad_enabled()
        return self._precompute_geometry(pointcloud, bool(inference))

    @torch.no_grad()
    def _precompute_geometry(self, pointcloud: torch.Tensor, infer: bool):
        """Everything in the forward that depends on the coordinates only — the FPS chain, the sampled
        centres, the ball-query neighbourhoods and the 3-NN interpolation weights of the FP levels.
        No parameters are involved, so a training pipeline can run this for batch i+1 on a side stream
        while batch i is in its forward/backward (bench.py does); pass the result as `geometry=`."""
        xyz = pointcloud[..., 0:3].contiguous()
        geo = {"sa": [], "fp": []}
        levels = [xyz]
        # the split cloud travels with the geometry: the forward takes these two tensors instead of slicing the cloud a
        # second time on the critical path (two strided copies of 19 MB each at the headline shape), after the same
        # identity check as the pre-grouped rows (`src`: storage, shape and version of the cloud they were cut from)
        geo["xyz"], geo["src"] = xyz, rows_source(pointcloud)
        # FPS shape that leaves half of the CUs to the co-running step (include/pn2_hip.h: PN2_FPS_FEW_CUS)
        # fp32 steps (13+ ms of matrix kernels) are longer than the sampling chain: give the sampling as few CUs as possible;
        # the bf16 step is shorter than the chain and wants the faster 128-CU shape
        from pointnet2_ops import eval_fused, fused_mlp
        bg = getattr(pointnet2_utils._ext, "background_geometry", None)
        # inference (eval mode, no gradient recorded): no backward will ask for the inverse neighbourhood indices, and the
        # one-kernel SA levels (pointnet2_ops/eval_fused.py) gather their own rows from idx — the query kernel need not emit
        # level 1's grouped rows (117 MB at the headline shape); the forward is short, so the sampling gets the faster shape
        with (bg(fewest=fused_mlp.mlp_dtype() == torch.float32 and not infer) if bg is not None else contextlib.nullcontext()):
            feats0 = (pointcloud[..., 3:].contiguous() if pointcloud.size(-1) > 3 and not pointcloud.requires_grad else None)
            geo["feats_rows"] = feats0
            pregroup = not (infer and eval_fused.eval_fused_enabled())
            for i in (1, 2, 3, 4):
                # levels 2-4 gather features that carry a gradient; level 1 reads the input colours: its grouped rows come
                # out of the ball query itself (pn2_ball_query_group) here, off the step's critical path
                g = getattr(self, f"sa{i}").sample_and_query(levels[-1], inverse_index=i > 1 and not infer,
                                                             feats_rows=feats0 if i == 1 and pregroup else None)
                if g.get("rows_src") is not None:
                    g["rows_src"] = rows_source(pointcloud)     # (the rows come from THIS cloud's colour columns)
                geo["sa"].append(g)
                levels.append(g["new_xyz"])
            geo["fp"].append(self.fp1.interpolation(levels[3], levels[4]))
            geo["fp"].append(self.fp2.interpolation(levels[2], levels[3]))
        return geo

    def forward(self, pointcloud: torch.Tensor, end_points=None, geometry=None):
        """pointcloud (B, N, 3 + input_feature_dim) -> end_points dict with
        sa{1..4}_xyz / _features (/ _inds for 1,2) and fp2_{features,xyz,inds}.
        `geometry` = precompute_geometry(pointcloud) (optional; identical results)."""
        end_points = end_points or {}
        src = None if geometry is None else geometry.get("src")
        if (src is not None and geometry.get("feats_rows") is not None and not pointcloud.requires_grad
                and (src.matches(pointcloud) or capturing())):
            xyz, features = geometry["xyz"], geometry["feats_rows"].transpose(1, 2)      # cut from THIS cloud by the prefetch
        else:
            xyz, features = self._break_up_pc(pointcloud)
        if geometry is not None:
            geometry = dict(geometry, sa=confirm_rows(geometry["sa"], pointcloud))
        for i in (1, 2, 3, 4):
            xyz, features, inds = getattr(self, f"sa{i}")(
                xyz, features, geometry=None if geometry is None else geometry["sa"][i - 1])
            if i <= 2:
                end_points[f"sa{i}_inds"] = inds
            end_points[f"sa{i}_xyz"] = xyz
            end_points[f"sa{i}_features"] = features
        features = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"],
                            end_points["sa3_features"], end_points["sa4_features"],
                            interp=None if geometry is None else geometry["fp"][0])
        features = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"],
                            end_points["sa2_features"], features,
                            interp=None if geometry is None else geometry["fp"][1])
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        return end_points
