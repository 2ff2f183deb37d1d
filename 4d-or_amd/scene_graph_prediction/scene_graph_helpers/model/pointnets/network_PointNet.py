"""Classification heads of the scene-graph model (dense layers; they stay on
rocBLAS through torch — SURVEY.md §8 A14).  Mirrors ``PointNetCls``
(SGH/model/pointnets/network_PointNet.py:188-224) and ``PointNetRelCls``
(:227-271): 256 -> 512 -> 256 (dropout 0.3 before the optional BN) -> k,
log-softmax; the relation head late-fuses the image embedding and the
subject/object one-hot before ``fc3``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from scene_graph_prediction.scene_graph_helpers.model.pointnets.networks_base import BaseNetwork


class _Head(BaseNetwork):
    def _make_trunk(self, in_size, batch_norm, drop_out):
        self.fc1 = nn.Linear(in_size, 512)
        self.fc2 = nn.Linear(512, 256)

    def _trunk(self, x, use_bn):
        x = self.fc1(x)
        if use_bn:
            x = self.bn1(x)
        x = self.relu(x)
        x = self.fc2(x)
        if self.use_drop_out:
            x = self.dropout(x)
        if use_bn:
            x = self.bn2(x)
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

    def forward(self, x):
        return F.log_softmax(self.fc3(self._trunk(x, self.use_batch_norm)), dim=1)


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

    def forward(self, x, relation_objects_one_hot=None, image_embeddings=None):
        x = self._trunk(x, self.use_bn)
        if image_embeddings is not None:            # late fusion of the scene-level image embedding
            x = torch.cat([x, image_embeddings.unsqueeze(0).repeat(len(x), 1)], dim=1)
        if relation_objects_one_hot is not None:    # late fusion of subject/object classes
            x = torch.cat([x, relation_objects_one_hot], dim=1)
        return F.log_softmax(self.fc3(x), dim=1)
