"""Scene-graph prediction model: two PointNet++ encoders (objects, object pairs),
a TripletGCN and two heads that emit <subject, predicate, object> triples.

API mirror of SGH/model/scene_graph_prediction_model.py (``SGPNModelWrapper``):
constructor signature (:31), sub-module names / ``state_dict`` keys
(``obj_encoder``, ``rel_encoder``, ``gcn``, ``obj_predictor``, ``rel_predictor``),
``forward(batch, return_meta_data)`` (:87-109), the loss of
``training_step`` / ``validation_step`` (:134-155), triple emission in
``predict_step`` (:157-177), per-take metric bookkeeping (:124-132, :195-238) and
``configure_optimizers`` (:240-242).  The reference derives from
pytorch_lightning.LightningModule (harness; not installed and out of scope):
this class is a plain ``nn.Module`` exposing the same step methods so any loop
(ours: bench.py / the DDP runner) can drive it.  ``IMAGE_INPUT == 'full'``
(timm EfficientNet-B5 late fusion) is rejected: stock-torch 2-D CNN, out of scope.
"""
from collections import defaultdict

import torch
import torch.nn.functional as F
import torch.optim as optim
from torch import nn

from scene_graph_prediction.scene_graph_helpers.model.gcns.network_TripletGCN import TripletGCNModel
from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet import PointNetCls, PointNetRelCls
from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet2 import PointNetfeat as PointNetfeat2


class SGPNModelWrapper(nn.Module):
    def __init__(self, config, num_class, num_rel, weights_obj, weights_rel, relationNames):
        super().__init__()
        self.config = config
        self.mconfig = config["MODEL"]
        self.n_object_types = 6
        # class weights of the two NLL terms: non-persistent buffers, so they follow .to(device) once instead
        # of being uploaded every step (reference :63-64 / :140-141) and stay out of the state_dict
        self.register_buffer("weights_obj", torch.as_tensor(weights_obj, dtype=torch.float32), persistent=False)
        self.register_buffer("weights_rel", torch.as_tensor(weights_rel, dtype=torch.float32), persistent=False)
        self.relationNames = relationNames
        self.lr = float(self.config["LR"])
        self.reset_metrics()
        if self.config["IMAGE_INPUT"] == "full":
            raise NotImplementedError("IMAGE_INPUT='full' needs timm's EfficientNet-B5 (out of scope, absent offline)")

        m = self.mconfig
        self.obj_encoder = PointNetfeat2(input_dim=6, out_size=m["point_feature_size"], input_dropout=m["INPUT_DROPOUT"])
        self.rel_encoder = PointNetfeat2(input_dim=7, out_size=m["edge_feature_size"], input_dropout=m["INPUT_DROPOUT"])
        self.gcn = TripletGCNModel(num_layers=m["N_LAYERS"], dim_node=m["point_feature_size"],
                                   dim_edge=m["edge_feature_size"], dim_hidden=m["gcn_hidden_feature_size"])
        self.obj_predictor = PointNetCls(num_class, in_size=m["point_feature_size"], batch_norm=False, drop_out=True)
        self.rel_predictor = PointNetRelCls(num_rel, in_size=m["edge_feature_size"], batch_norm=False, drop_out=True,
                                            image_embedding_size=None, n_object_types=self.n_object_types)

    # ------------------------------------------------------------------ forward
    def precompute_geometry(self, batch):
        """Sampling / grouping geometry of both encoders for `batch` (no parameters involved): a loop that already holds
        the next scan can run this on a side stream and store the result as batch["geometry"]."""
        return {"obj": self.obj_encoder.precompute_geometry(batch["obj_points"]),
                "rel": self.rel_encoder.precompute_geometry(batch["rel_points"])}

    def forward(self, batch, return_meta_data=False):
        geo = batch.get("geometry")
        obj_feature = self.obj_encoder(batch["obj_points"], geometry=None if geo is None else geo["obj"])
        rel_feature = self.rel_encoder(batch["rel_points"], geometry=None if geo is None else geo["rel"])
        gcn_obj_feature, gcn_rel_feature = self.gcn(obj_feature, rel_feature, batch["edge_indices"], batch.get("edge_csr"))
        obj_cls = self.obj_predictor(gcn_obj_feature if self.mconfig["OBJ_PRED_FROM_GCN"] else obj_feature)
        rel_cls = self.rel_predictor(gcn_rel_feature, relation_objects_one_hot=batch["relation_objects_one_hot"])
        if return_meta_data:
            return obj_cls, rel_cls, obj_feature, rel_feature, gcn_obj_feature, gcn_rel_feature, None
        return obj_cls, rel_cls

    # ------------------------------------------------------------------ steps
    def loss(self, obj_pred, rel_pred, batch):
        loss_obj = F.nll_loss(obj_pred, batch["gt_class"], weight=self.weights_obj.to(batch["gt_class"].device, non_blocking=True))
        loss_rel = F.nll_loss(rel_pred, batch["gt_rels"], weight=self.weights_rel.to(batch["gt_rels"].device, non_blocking=True))
        return self.mconfig["lambda_o"] * loss_obj + loss_rel

    def _step(self, batch, split):
        obj_pred, rel_pred, *_ = self(batch, return_meta_data=True)
        loss = self.loss(obj_pred, rel_pred, batch)
        self.update_metrics(batch, rel_pred, split=split)
        return loss

    def training_step(self, batch, batch_idx=0):
        return self._step(batch, "train")

    def validation_step(self, batch, batch_idx=0):
        return self._step(batch, "val")

    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        """-> (scan_id, [(subject_name, predicate, object_name), ...]); 'none' edges dropped."""
        _, rel_pred, *_ = self(batch, return_meta_data=True)
        predicted = torch.max(rel_pred.detach(), 1)[1].cpu().tolist()
        none_id = self.relationNames.index("none")
        edges = batch["edge_indices"].transpose(0, 1).cpu().tolist()
        triples = []
        for (start, end), rel in zip(edges, predicted):
            if rel == none_id:
                continue
            triples.append((batch["objs_json"][start + 1], self.relationNames[rel], batch["objs_json"][end + 1]))
        return batch["scan_id"], triples

    def configure_optimizers(self, capturable=False):
        """AdamW(lr=LR, weight_decay=W_DECAY) (reference :240-242); `capturable` keeps the step counters on
        the device so the update can be replayed inside a hipGraph (runtime.GraphedTrainStep)."""
        return optim.AdamW(params=self.parameters(), lr=self.lr, weight_decay=float(self.config["W_DECAY"]),
                           capturable=bool(capturable))

    def pure_training_step(self, batch):
        """(loss, rel_pred) without host-side bookkeeping: the body a graph capture needs."""
        obj_pred, rel_pred = self(batch)
        return self.loss(obj_pred, rel_pred, batch), rel_pred

    # ------------------------------------------------------------------ metrics
    def reset_metrics(self, split=None):
        if split in (None, "train"):
            self.train_take_rel_preds, self.train_take_rel_gts = defaultdict(list), defaultdict(list)
        if split in (None, "val"):
            self.val_take_rel_preds, self.val_take_rel_gts = defaultdict(list), defaultdict(list)

    def update_metrics(self, batch, rel_pred, split="train"):
        if split not in ("train", "val"):
            raise NotImplementedError()
        preds = getattr(self, f"{split}_take_rel_preds")
        gts = getattr(self, f"{split}_take_rel_gts")
        take = batch.get("take_idx", 0)
        preds[take].extend(rel_pred.detach().cpu().numpy().argmax(1))
        gts[take].extend(batch["gt_rels"].detach().cpu().numpy())

    def evaluate_predictions(self, epoch_loss, split):
        """Per-take and overall precision / recall / F1 (sklearn classification_report, like the
        reference); returns {'macro_f1', 'macro_prec', 'macro_rec', 'weighted_*', 'per_take'}."""
        from sklearn.metrics import classification_report
        if split not in ("train", "val"):
            raise NotImplementedError()
        preds = getattr(self, f"{split}_take_rel_preds")
        gts = getattr(self, f"{split}_take_rel_gts")
        labels = list(range(len(self.relationNames)))
        all_gt, all_pred, per_take = [], [], {}
        for take in sorted(preds.keys()):
            all_gt.extend(gts[take])
            all_pred.extend(preds[take])
            per_take[take] = classification_report(gts[take], preds[take], labels=labels,
                                                   target_names=self.relationNames, output_dict=True,
                                                   zero_division=0)
        res = classification_report(all_gt, all_pred, labels=labels, target_names=self.relationNames,
                                    output_dict=True, zero_division=0)
        return {"epoch_loss": float(epoch_loss), "macro_f1": res["macro avg"]["f1-score"],
                "macro_prec": res["macro avg"]["precision"], "macro_rec": res["macro avg"]["recall"],
                "weighted_f1": res["weighted avg"]["f1-score"], "weighted_prec": res["weighted avg"]["precision"],
                "weighted_rec": res["weighted avg"]["recall"], "per_take": per_take}
