"""``PointNetfeat``: channel-first cloud (B, 3+C, N) -> (B, 256) descriptor through the
MSG PointNet++ encoder.  Mirrors SGH/model/pointnets/network_PointNet2.py:13-25
(`out_size` and `input_dropout` are stored but, as in the reference, unused)."""
import torch.nn as nn

from scene_graph_prediction.pointnet2_dir.pointnet2.models.pointnet2_msg_cls import PointNet2ClassificationMSG


class PointNetfeat(nn.Module):
    def __init__(self, input_dim=6, out_size=1024, input_dropout=0.0):
        super().__init__()
        self.name = "pnetenc"
        self.backbone = PointNet2ClassificationMSG(input_dim=input_dim)
        self.out_size = out_size
        self.input_dropout = input_dropout

    def precompute_geometry(self, x):
        """FPS chain + ball queries of the encoder for the channel-first clouds `x` (see the backbone)."""
        return self.backbone.precompute_geometry(x.transpose(1, 2))

    def forward(self, x, geometry=None):
        assert x.ndim > 2
        return self.backbone(x.transpose(1, 2), return_features=True, geometry=geometry)[:, :, 0]
