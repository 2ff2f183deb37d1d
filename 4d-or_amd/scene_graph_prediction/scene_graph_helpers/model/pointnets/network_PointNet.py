"""Classification heads of the scene-graph model (dense layers; they stay on
rocBLAS through torch — SURVEY.md §8 A14).  Mirrors ``PointNetCls``
(SGH/model/pointnets/network_PointNet.py:188-224) and ``PointNetRelCls``
(:227-271): 256 -> 512 -> 256 (dropout 0.3 before the optional BN) -> k,
log-softmax; the relation head late-fuses the image embedding and the
subject/object one-hot before ``fc3``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from scene_graph_prediction.scene_graph_helpers.model.gcns.network_TripletGCN import SegmentBatchNorm
from scene_graph_prediction.scene_graph_helpers.model.pointnets.networks_base import BaseNetwork


def scan_batch_norm(bn: nn.BatchNorm1d, x, ptr):
    """`bn(x)` for rows of S scans [ptr[s], ptr[s+1]): in training mode the batch statistics are per scan and the running
    statistics receive the S updates of S single-scan steps, in scan order (momentum EMA, unbiased variance —
    torch.nn.functional.batch_norm); in eval mode (running statistics) it is plain `bn(x)`."""
    if ptr is None or not (bn.training or bn.running_mean is None):
        return bn(x)
    y, mean, rstd = SegmentBatchNorm.apply(x, ptr, bn.weight, bn.bias, bn.eps)
    if bn.running_mean is not None:
        with torch.no_grad():
            S = mean.size(0)
            if bn.momentum is not None and x.is_cuda:                              # one launch (pn2_segment_bn_running_update)
                from pointnet2_ops import _ext
                _ext.segment_bn_running_update(mean, rstd, ptr, bn.eps, float(bn.momentum), bn.running_mean, bn.running_var,
                                               bn.num_batches_tracked)
                return y
            n = (ptr[1:] - ptr[:-1]).to(x.dtype).unsqueeze(1)                      # rows per scan
            var = (1.0 / (rstd * rstd) - bn.eps).clamp_min(0) * n / (n - 1).clamp_min(1)
            if bn.momentum is None:                                                # cumulative average
                for s in range(S):
                    bn.num_batches_tracked += 1
                    f = 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.lerp_(mean[s], f)
                    bn.running_var.lerp_(var[s], f)
            else:
                m = float(bn.momentum)
                w = m * (1.0 - m) ** torch.arange(S - 1, -1, -1, device=x.device, dtype=x.dtype)   # weight of scan s
                bn.running_mean.mul_((1.0 - m) ** S).add_((w.unsqueeze(1) * mean).sum(0))
                bn.running_var.mul_((1.0 - m) ** S).add_((w.unsqueeze(1) * var).sum(0))
                bn.num_batches_tracked += S
    return y


class _Head(BaseNetwork):
    def _make_trunk(self, in_size, batch_norm, drop_out):
        self.fc1 = nn.Linear(in_size, 512)
        self.fc2 = nn.Linear(512, 256)

    def _trunk(self, x, use_bn, scan_ptr=None):
        x = self.fc1(x)
        if use_bn:
            x = scan_batch_norm(self.bn1, x, scan_ptr)
        x = self.relu(x)
        x = self.fc2(x)
        if self.use_drop_out:
            x = self.dropout(x)
        if use_bn:
            x = scan_batch_norm(self.bn2, x, scan_ptr)
        return self.relu(x)


class PointNetCls(_Head):
    def __init__(self, k=2, in_size=1024, batch_norm=True, drop_out=True, init_weights=True):
        super().__init__()
        self.name = "pnetcls"
        self.in_size, self.k = in_size, k
        self.use_batch_norm, self.use_drop_out = batch_norm, drop_out
        self._make_trunk(in_size, batch_norm, drop_out)
        self.fc3 = nn.Linear(256, k)
        if drop_out:
            self.dropout = nn.Dropout(p=0.3)
        if batch_norm:
            self.bn1 = nn.BatchNorm1d(512)
            self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()
        if init_weights:
            self.init_weights("constant", 1, target_op="BatchNorm")
            self.init_weights("xavier_normal", 1)

    def forward(self, x, scan_ptr=None):
        """`scan_ptr` (S+1 row offsets, optional): the rows are S scans and training-mode BatchNorm statistics are per scan."""
        return F.log_softmax(self.fc3(self._trunk(x, self.use_batch_norm, scan_ptr)), dim=1)


class PointNetRelCls(_Head):
    def __init__(self, k=2, in_size=1024, batch_norm=True, drop_out=True, init_weights=True,
                 image_embedding_size=None, n_object_types=None):
        super().__init__()
        self.name = "pnetcls"
        self.in_size = in_size
        self.use_bn, self.use_drop_out = batch_norm, drop_out
        self._make_trunk(in_size, batch_norm, drop_out)
        self.fc3 = nn.Linear(256 + (image_embedding_size or 0) + n_object_types * 2, k)
        if drop_out:
            self.dropout = nn.Dropout(p=0.3)
        if batch_norm:
            self.bn1 = nn.BatchNorm1d(512)
            self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()
        if init_weights:
            self.init_weights("constant", 1, target_op="BatchNorm")
            self.init_weights("xavier_normal", 1)

    def forward(self, x, relation_objects_one_hot=None, image_embeddings=None, scan_ptr=None):
        x = self._trunk(x, self.use_bn, scan_ptr)
        if image_embeddings is not None:            # late fusion of the scene-level image embedding
            x = torch.cat([x, image_embeddings.unsqueeze(0).repeat(len(x), 1)], dim=1)
        if relation_objects_one_hot is not None:    # late fusion of subject/object classes
            x = torch.cat([x, relation_objects_one_hot], dim=1)
        return F.log_softmax(self.fc3(x), dim=1)
