"""Single-scale-grouping PointNet++ classifier backbone (base class of the MSG
encoder 4D-OR actually uses).  Mirrors PN2/models/pointnet2_ssg_cls.py:55-124
(PN2 = scene_graph_prediction/pointnet2_dir/pointnet2): ``__init__(input_dim)``,
``_build_model``, ``_break_up_pc``, ``forward(pointcloud, return_features)`` and
the parameter layout, INCLUDING the 1024->512->256->40 ``fc_layer`` head that the
MSG subclass inherits but never feeds (msg_cls.py:47 calls ``super()._build_model()``
first): the paper checkpoints contain those tensors, so they must exist for a
strict ``load_state_dict``.  Lightning / ModelNet training hooks of the upstream
file are out of scope."""
import torch
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import PointnetSAModule, confirm_rows, rows_source


class PointNet2ClassificationSSG(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self._build_model()

    def _build_model(self):
        c = self.input_dim - 3
        self.SA_modules = nn.ModuleList([
            PointnetSAModule(npoint=512, radius=0.2, nsample=64, mlp=[c, 64, 64, 128], use_xyz=True),
            PointnetSAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256], use_xyz=True),
            PointnetSAModule(mlp=[256, 256, 512, 1024], use_xyz=True),
        ])
        self.fc_layer = nn.Sequential(
            nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.ReLU(True),
            nn.Linear(512, 256, bias=False), nn.BatchNorm1d(256), nn.ReLU(True),
            nn.Dropout(0.5), nn.Linear(256, 40),
        )

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def precompute_geometry(self, pointcloud):
        """The coordinate-only part of the forward (FPS chain + ball queries of every SA level; no parameters):
        lets a loop prepare the next clouds on a side stream.  Pass the result as `geometry=`."""
        xyz = pointcloud[..., 0:3].contiguous()
        # the first level groups the INPUT features (colours / instance mask: data, no gradient): its grouped rows can be
        # emitted by the ball query itself (pn2_ball_query_group), ahead of the step like the rest of the geometry
        feats0 = pointcloud[..., 3:].contiguous() if pointcloud.size(-1) > 3 and not pointcloud.requires_grad else None
        # inference (eval mode, no gradient recorded): no backward will need the inverse neighbourhood indices, and the
        # one-kernel SA levels (pointnet2_ops/eval_fused.py) gather their own rows from idx
        from pointnet2_ops import eval_fused
        infer = not self.training and not torch.is_grad_enabled()
        if infer and eval_fused.eval_fused_enabled():
            feats0 = None
        geo = []
        for k, sa in enumerate(self.SA_modules):
            g = sa.sample_and_query(xyz, inverse_index=k > 0 and not infer, feats_rows=feats0 if k == 0 else None)
            if g.get("rows_src") is not None:
                g["rows_src"] = rows_source(pointcloud)         # (the rows come from THIS cloud's feature columns)
            geo.append(g)
            xyz = g["new_xyz"]
            if xyz is None:                      # group-all level: nothing below depends on coordinates
                break
        return geo

    def forward(self, pointcloud, return_features=False, geometry=None):
        """pointcloud (B, N, 3 + C), each point (x, y, z, features...) ->
        (B, C_last, 1) set features if `return_features` else class logits.
        `geometry` = precompute_geometry(pointcloud) (optional; identical results)."""
        xyz, features = self._break_up_pc(pointcloud)
        if geometry is not None:
            geometry = confirm_rows(geometry, pointcloud)
        for i, sa in enumerate(self.SA_modules):
            g = geometry[i] if geometry is not None and i < len(geometry) else None
            xyz, features = sa(xyz, features, geometry=g) if g is not None else sa(xyz, features)
        if return_features:
            return features
        return self.fc_layer(features.squeeze(-1))
