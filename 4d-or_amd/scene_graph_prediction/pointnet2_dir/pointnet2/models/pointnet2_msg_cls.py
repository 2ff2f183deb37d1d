"""Multi-scale-grouping PointNet++ encoder — the "max we can run" variant 4D-OR
ships (PN2/models/pointnet2_msg_cls.py:45-78): SA1 512 centres, radii
[0.1, 0.2], nsamples [16, 32]; SA2 128 centres, radii [0.2, 0.4], nsamples
[32, 64]; SA3 groups everything.  The SSG modules built by the parent are
constructed first and then replaced, exactly like the reference, so parameter
initialisation consumes the RNG in the same order and the inherited dead
``fc_layer`` stays in the ``state_dict``."""
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import PointnetSAModule, PointnetSAModuleMSG
from scene_graph_prediction.pointnet2_dir.pointnet2.models.pointnet2_ssg_cls import PointNet2ClassificationSSG


class PointNet2ClassificationMSG(PointNet2ClassificationSSG):
    def _build_model(self):
        super()._build_model()
        c = self.input_dim - 3
        sa1 = PointnetSAModuleMSG(npoint=512, radii=[0.1, 0.2], nsamples=[16, 32],
                                  mlps=[[c, 64, 64], [c, 64, 128]], use_xyz=True)
        wide = 64 + 128
        sa2 = PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4], nsamples=[32, 64],
                                  mlps=[[wide, 128, 128], [wide, 128, 128]], use_xyz=True)
        sa3 = PointnetSAModule(mlp=[128 + 128, 256, 256], use_xyz=True)
        self.SA_modules = nn.ModuleList([sa1, sa2, sa3])
