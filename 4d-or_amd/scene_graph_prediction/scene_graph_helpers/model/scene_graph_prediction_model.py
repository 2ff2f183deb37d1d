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
(ours: bench.py / the DDP runner) can drive it.

``IMAGE_INPUT == 'full'`` (:47-55, :96-100): the 2-D CNN itself (timm ``tf_efficientnet_b5_ns``, stock torch, absent
offline) is out of scope, but everything after it is here — ``full_image_feature_reduction`` (Linear
``num_features -> FULL_IMAGE_EMBEDDING_SIZE // 6``), the flatten over the six views and the late fusion in
``PointNetRelCls`` — so the ``no_gt_image`` checkpoints load (``full_image_model.*`` entries are skipped unless a CNN
was attached with ``attach_image_model``).  The batch then carries either ``full_image`` (6,3,H,W; needs an attached
CNN) or ``full_image_features`` (6, num_features) computed by it ahead of time.
"""
import contextlib
import os
from collections import defaultdict

import torch
import torch.nn.functional as F
import torch.optim as optim
from torch import nn

from pointnet2_ops.pointnet2_modules import per_scan_statistics
from scene_graph_prediction.scene_graph_helpers.model.gcns.network_TripletGCN import OneScan, TripletGCNModel
from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet import PointNetCls, PointNetRelCls
from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet2 import PointNetfeat as PointNetfeat2


_ENCODER_STREAMS = {}


def _encoder_stream(device):
    """One extra stream per device for the object encoder (created once: a stream capture must not meet its creation)."""
    st = _ENCODER_STREAMS.get(device)
    if st is None:
        st = _ENCODER_STREAMS[device] = torch.cuda.Stream(device=device)
        # the object encoder's parameter gradients are produced on this stream and accumulated on the main one — on purpose
        # (see forward); torch would warn about the mismatch once per process
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
    return st


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
        m = self.mconfig
        self.with_images = self.config["IMAGE_INPUT"] == "full"
        self.full_image_model = None
        self.obj_encoder = PointNetfeat2(input_dim=6, out_size=m["point_feature_size"], input_dropout=m["INPUT_DROPOUT"])
        self.rel_encoder = PointNetfeat2(input_dim=7, out_size=m["edge_feature_size"], input_dropout=m["INPUT_DROPOUT"])
        if self.with_images:
            # registered BEFORE the GCN, like the reference (:47-57): state_dict key order and the order the constructor
            # draws from the RNG are the reference's (tests/golden/sgpn.npz, generated from the reference class).
            # EfficientNet-B5's `num_features` (2048) unless the config says otherwise (reference :57)
            self.full_image_feature_reduction = nn.Linear(int(m.get("IMAGE_MODEL_NUM_FEATURES", 2048)),
                                                          m["FULL_IMAGE_EMBEDDING_SIZE"] // 6)
        self.gcn = TripletGCNModel(num_layers=m["N_LAYERS"], dim_node=m["point_feature_size"],
                                   dim_edge=m["edge_feature_size"], dim_hidden=m["gcn_hidden_feature_size"])
        self.obj_predictor = PointNetCls(num_class, in_size=m["point_feature_size"], batch_norm=False, drop_out=True)
        self.rel_predictor = PointNetRelCls(num_rel, in_size=m["edge_feature_size"], batch_norm=False, drop_out=True,
                                            image_embedding_size=m["FULL_IMAGE_EMBEDDING_SIZE"] if self.with_images else None,
                                            n_object_types=self.n_object_types)

    # ------------------------------------------------------------------ image branch (late fusion only)
    def attach_image_model(self, cnn: nn.Module):
        """Plug in the 2-D backbone (any module mapping (6,3,H,W) -> (6, num_features)); frozen except `conv_head`
        and with its BatchNorms in eval mode, like the reference (:51-55, :74-85)."""
        self.full_image_model = cnn
        for p in cnn.parameters():
            p.requires_grad = False
        head = getattr(cnn, "conv_head", None)
        if head is not None:
            for p in head.parameters():
                p.requires_grad = True
        return self

    def freeze_image_model_batchnorm(self):
        if self.full_image_model is None:
            return
        for module in self.full_image_model.modules():
            if isinstance(module, (nn.BatchNorm2d, nn.BatchNorm1d)):
                if getattr(module, "weight", None) is not None:
                    module.weight.requires_grad_(False)
                if getattr(module, "bias", None) is not None:
                    module.bias.requires_grad_(False)
                module.eval()

    def load_state_dict(self, state_dict, strict=True, **kw):
        if self.with_images and self.full_image_model is None:
            state_dict = {k: v for k, v in state_dict.items() if not k.startswith("full_image_model.")}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _image_embedding(self, batch):
        if "full_image_features" in batch:
            feats = batch["full_image_features"]
        elif self.full_image_model is not None:
            self.freeze_image_model_batchnorm()
            feats = self.full_image_model(batch["full_image"])
        else:
            raise RuntimeError("IMAGE_INPUT='full': the batch needs 'full_image_features' (6, num_features), or attach the "
                               "2-D CNN with attach_image_model() and pass 'full_image'")
        if feats.dim() == 3:                                        # (S, 6, F): one embedding per scan of a batch
            return self.full_image_feature_reduction(feats).flatten(1)
        return self.full_image_feature_reduction(feats).flatten()

    # ------------------------------------------------------------------ forward
    def precompute_geometry(self, batch):
        """Sampling / grouping geometry of both encoders for `batch` (no parameters involved): a loop that already holds
        the next scan can run this on a side stream and store the result as batch["geometry"]."""
        return {"obj": self.obj_encoder.precompute_geometry(batch["obj_points"]),
                "rel": self.rel_encoder.precompute_geometry(batch["rel_points"])}

    #: Batched scans in TRAINING mode: BatchNorm batch statistics per scan everywhere (encoders, GCN, heads), i.e. the
    #: arithmetic of S single-scan steps of the reference (main.py:54-56) with their gradients averaged.  False: the encoders
    #: and heads normalise over the whole batch (fewer launches, a different — larger-batch — BatchNorm).
    per_scan_statistics = True
    #: a single scan on the GPU through the fused per-scan BatchNorm kernels too (38 launches fewer per step).  Off by
    #: default: measured on one box, 1 scan per step, 115.9 / 116.4 scans/s without vs 113.5 / 115.3 with — a Python autograd
    #: Function + C call per BatchNorm costs the host thread as much as torch's four native launches, and the step is bound
    #: by the host thread, not by the launch count (PN2_ONE_SCAN_SEGMENTS=1 switches it on)
    one_scan_segments = os.environ.get("PN2_ONE_SCAN_SEGMENTS", "0") == "1"

    #: object and relation encoder on two streams (forward and, through autograd, backward).  PN2_ENCODER_STREAMS=0: one stream.
    encoder_streams = os.environ.get("PN2_ENCODER_STREAMS", "1") == "1"

    def forward(self, batch, return_meta_data=False):
        geo = batch.get("geometry")
        scenes = batch.get("scenes")          # block-diagonal batch of several scans (dataset/synthetic.py::collate_scans)
        per_scan = scenes is not None and scenes.num_scenes > 1 and self.training and self.per_scan_statistics
        node_ptr = scenes.node_ptr if per_scan else None
        edge_ptr = scenes.edge_ptr if per_scan else None
        # (a step of ONE scan enters the context too, without a table: the encoders then keep the kernel routes a batch of
        # scans takes, so that S scans per step stay the arithmetic of S single-scan steps — pointnet2_ops/fused_mlp.py group[9])
        one_scan = self.training and self.per_scan_statistics and not per_scan
        with (per_scan_statistics(scenes.nodes_per_scene, scenes.edges_per_scene) if per_scan
              else (per_scan_statistics() if one_scan else contextlib.nullcontext())):
            if self.encoder_streams and batch["obj_points"].is_cuda and not torch.cuda.is_current_stream_capturing():
                # (inside a stream capture the fork only adds graph edges: replay of the one-scan step 135 -> 68 scans/s)
                # the two encoders share nothing until the GCN: the object encoder (9 small clouds per scan: kernels that
                # fill a fraction of the chip) runs on a second stream next to the relation encoder (72 clouds of 8000
                # points); autograd replays each encoder's backward on the stream its forward ran on
                main = torch.cuda.current_stream(batch["obj_points"].device)
                side = _encoder_stream(batch["obj_points"].device)
                side.wait_stream(main)
                # autograd runs a leaf's AccumulateGrad on the stream that was current when the node was CREATED (first use
                # of the parameter in a graph) and the node can outlive the step: created on the side stream it would later
                # run there even inside a stream capture of a single-stream step — outside the captured graph (replayed
                # gradients of the object encoder were garbage).  A view of every parameter taken HERE creates the nodes on
                # the main stream; the encoder's own uses then find them.
                keep = [p_.view_as(p_) for p_ in self.obj_encoder.parameters() if p_.requires_grad] if torch.is_grad_enabled() else None
                with torch.cuda.stream(side):
                    obj_feature = self.obj_encoder(batch["obj_points"], geometry=None if geo is None else geo["obj"])
                del keep
                rel_feature = self.rel_encoder(batch["rel_points"], geometry=None if geo is None else geo["rel"])
                main.wait_stream(side)
                obj_feature.record_stream(main)
            else:
                obj_feature = self.obj_encoder(batch["obj_points"], geometry=None if geo is None else geo["obj"])
                rel_feature = self.rel_encoder(batch["rel_points"], geometry=None if geo is None else geo["rel"])
        gcn_scenes = scenes
        if scenes is None and self.one_scan_segments and obj_feature.is_cuda and rel_feature.size(0) >= 2:
            # one scan on the GPU: the BatchNorm1d layers of the GCN and (training) of the heads through the per-scan kernel
            # of the batched path — same statistics, a third of the launches (network_TripletGCN.OneScan)
            gcn_scenes = OneScan.get(obj_feature.device, obj_feature.size(0), rel_feature.size(0))
            if self.training and obj_feature.size(0) >= 2:
                node_ptr, edge_ptr = gcn_scenes.node_ptr, gcn_scenes.edge_ptr
        gcn_obj_feature, gcn_rel_feature = self.gcn(obj_feature, rel_feature, batch["edge_indices"], batch.get("edge_csr"),
                                                    scenes=gcn_scenes)
        obj_cls = self.obj_predictor(gcn_obj_feature if self.mconfig["OBJ_PRED_FROM_GCN"] else obj_feature,
                                     scan_ptr=node_ptr)
        if self.with_images:
            emb = self._image_embedding(batch)
            if emb.dim() == 2:                                      # batched scans: every edge gets its own scan's embedding
                emb = emb[scenes.edge_scene]
                rel_cls = self.rel_predictor(gcn_rel_feature, relation_objects_one_hot=torch.cat(
                    [emb, batch["relation_objects_one_hot"]], dim=1), scan_ptr=edge_ptr)
            else:
                rel_cls = self.rel_predictor(gcn_rel_feature, relation_objects_one_hot=batch["relation_objects_one_hot"],
                                             image_embeddings=emb, scan_ptr=edge_ptr)
        else:
            rel_cls = self.rel_predictor(gcn_rel_feature, relation_objects_one_hot=batch["relation_objects_one_hot"],
                                         scan_ptr=edge_ptr)
        if return_meta_data:
            return obj_cls, rel_cls, obj_feature, rel_feature, gcn_obj_feature, gcn_rel_feature, None
        return obj_cls, rel_cls

    # ------------------------------------------------------------------ steps
    @staticmethod
    def _nll_per_scene(logp, target, weight, scene, num_scenes):
        """Weighted NLL averaged PER SCAN, then over the scans: what the reference's one-scan steps optimise on
        average (F.nll_loss(weight=...) divides by the sum of the target weights of that scan)."""
        w = weight[target]
        picked = -logp.gather(1, target.unsqueeze(1)).squeeze(1) * w
        num = torch.zeros(num_scenes, dtype=logp.dtype, device=logp.device).index_add_(0, scene, picked)
        den = torch.zeros(num_scenes, dtype=logp.dtype, device=logp.device).index_add_(0, scene, w)
        # a scan whose targets all carry zero class weight has no loss (F.nll_loss of that scan alone is 0/0): it is left
        # out of the mean instead of poisoning the other scans' loss and gradients with NaN
        valid = den > 0
        per = torch.where(valid, num / torch.where(valid, den, torch.ones_like(den)), torch.zeros_like(num))
        return per.sum() / valid.sum().clamp_min(1).to(per.dtype)

    def loss(self, obj_pred, rel_pred, batch):
        w_obj = self.weights_obj.to(batch["gt_class"].device, non_blocking=True)
        w_rel = self.weights_rel.to(batch["gt_rels"].device, non_blocking=True)
        scenes = batch.get("scenes")
        if scenes is not None:
            loss_obj = self._nll_per_scene(obj_pred, batch["gt_class"], w_obj, scenes.node_scene, scenes.num_scenes)
            loss_rel = self._nll_per_scene(rel_pred, batch["gt_rels"], w_rel, scenes.edge_scene, scenes.num_scenes)
        else:
            loss_obj = F.nll_loss(obj_pred, batch["gt_class"], weight=w_obj)
            loss_rel = F.nll_loss(rel_pred, batch["gt_rels"], weight=w_rel)
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
        scenes = batch.get("scenes")
        if scenes is None:
            triples = []
            for (start, end), rel in zip(edges, predicted):
                if rel == none_id:
                    continue
                triples.append((batch["objs_json"][start + 1], self.relationNames[rel], batch["objs_json"][end + 1]))
            return batch["scan_id"], triples
        # block-diagonal batch: one (scan_id, triples) per scan, node ids local to the scan again
        node_ptr, edge_ptr = scenes.node_ptr.cpu().tolist(), scenes.edge_ptr.cpu().tolist()
        out = []
        for s in range(scenes.num_scenes):
            objs, off, triples = batch["objs_jsons"][s], node_ptr[s], []
            for e in range(edge_ptr[s], edge_ptr[s + 1]):
                rel = predicted[e]
                if rel == none_id:
                    continue
                start, end = edges[e]
                triples.append((objs[start - off + 1], self.relationNames[rel], objs[end - off + 1]))
            out.append((batch["scan_ids"][s], triples))
        return out

    def configure_optimizers(self, capturable=False):
        """AdamW(lr=LR, weight_decay=W_DECAY) (reference :240-242); `capturable` keeps the step counters on
        the device so the update can be replayed inside a hipGraph (runtime.GraphedTrainStep)."""
        # fused=True on the GPU: the same update as ONE kernel over all parameter tensors — the multi-tensor form costs the
        # host thread ~1 ms of the 8 ms one-scan step (tools/host_profile.py sgp); same arithmetic
        params = list(self.parameters())
        return optim.AdamW(params=params, lr=self.lr, weight_decay=float(self.config["W_DECAY"]),
                           capturable=bool(capturable), fused=bool(params) and all(p.is_cuda for p in params))

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
        p, g = rel_pred.detach().cpu().numpy().argmax(1), batch["gt_rels"].detach().cpu().numpy()
        scenes = batch.get("scenes")
        if scenes is None:
            take = batch.get("take_idx", 0)
            preds[take].extend(p)
            gts[take].extend(g)
            return
        edge_ptr = scenes.edge_ptr.cpu().tolist()
        for s, take in enumerate(batch.get("take_idxs", [0] * scenes.num_scenes)):
            preds[take].extend(p[edge_ptr[s]:edge_ptr[s + 1]])
            gts[take].extend(g[edge_ptr[s]:edge_ptr[s + 1]])

    def evaluate_predictions(self, epoch_loss, split, print_reports=False, log=print):
        """Per-take and overall precision / recall / F1 (sklearn classification_report, like the
        reference); returns {'macro_f1', 'macro_prec', 'macro_rec', 'weighted_*', 'per_take'}.  `print_reports`: also
        print the text reports the reference prints (:216-218, :230-235: "Take k", "<split> Results:")."""
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
            if print_reports:
                log(f"\nTake {take}\n")
                log(classification_report(gts[take], preds[take], labels=labels, target_names=self.relationNames,
                                          zero_division=0))
        res = classification_report(all_gt, all_pred, labels=labels, target_names=self.relationNames,
                                    output_dict=True, zero_division=0)
        if print_reports:
            log(f"{split} Results:\n")
            log(classification_report(all_gt, all_pred, labels=labels, target_names=self.relationNames, zero_division=0))
        return {"epoch_loss": float(epoch_loss), "macro_f1": res["macro avg"]["f1-score"],
                "macro_prec": res["macro avg"]["precision"], "macro_rec": res["macro avg"]["recall"],
                "weighted_f1": res["weighted avg"]["f1-score"], "weighted_prec": res["weighted avg"]["precision"],
                "weighted_rec": res["weighted avg"]["recall"], "per_take": per_take}
