"""Thin runner for the scene-graph model on MI355X.

Counterpart of scene_graph_prediction/main.py (:36-117: ``--config``, modes train / evaluate /
infer, JSON dump of ``{scan_id: [[sub, pred, obj], ...]}`` at :111-115) without the
pytorch_lightning harness and without the 4D-OR dataset (neither ships with the reference
tree).  It drives ``SGPNModelWrapper`` on synthetic scans of the real
shape so the whole path — encoders, TripletGCN, heads, loss, triple emission, wire format —
runs end to end, and loads the paper checkpoints unchanged when given ``--weights``.

    python -m scene_graph_prediction.main --config no_gt.json --mode infer --scans 4 --out scan_relations.json
"""
import argparse
import json
import os

import torch

from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

RELATION_NAMES = ["Assisting", "Cementing", "Cleaning", "CloseTo", "Cutting", "Drilling", "Hammering", "Holding",
                  "LyingOn", "Operating", "Preparing", "Sawing", "Suturing", "Touching", "none"]   # data/relationships.txt + 'none'


def config_loader(path):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_graph_helpers", "configs")
    full = path if os.path.exists(path) else os.path.join(here, path)
    with open(full) as f:
        return json.load(f)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="no_gt.json")
    ap.add_argument("--mode", choices=["train", "evaluate", "infer"], default="infer")
    ap.add_argument("--scans", type=int, default=2, help="synthetic scans to process")
    ap.add_argument("--objects", type=int, default=9)
    ap.add_argument("--weights", type=str, default=None, help="state_dict (.pth) of the reference model")
    ap.add_argument("--out", type=str, default="scan_relations_synthetic.json")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--graphs", action="store_true",
                    help="train mode: replay each scan shape's whole step (fwd + loss + bwd + AdamW) as one hipGraph")
    args = ap.parse_args(argv)

    torch.manual_seed(42)                                     # main.py:40 seeds everything with 42
    config = config_loader(args.config)
    dcfg = config["dataset"]
    model = SGPNModelWrapper(config, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)),
                             RELATION_NAMES).cuda()
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location="cuda"))
    scans = [synthetic_scan(args.objects, dcfg["num_points_objects"], dcfg["num_points_relation"], seed=i,
                            scan_id=f"synthetic_{i:06d}") for i in range(args.scans)]

    if args.mode == "train":
        model.train()
        opt = model.configure_optimizers(capturable=args.graphs)
        if args.graphs:
            from runtime import GraphedTrainStep
            # flat gradients are dense zeros where eager autograd leaves None: keep AdamW's weight decay off the
            # classifier heads the encoders inherit but never call by freezing them
            for n, p in model.named_parameters():
                if ".backbone.fc_layer." in n:
                    p.requires_grad_(False)
            opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=model.lr,
                                    weight_decay=float(config["W_DECAY"]), capturable=True)
            graphed = GraphedTrainStep(model.pure_training_step, model.parameters(), opt)
        from runtime import GeometryPrefetcher
        for epoch in range(args.epochs):
            # batch = one scan per step, like main.py:54-56; the next scan's sampling geometry is prefetched on a side stream
            device_scans = (to_device(scan, "cuda") for scan in scans)
            for i, batch in enumerate(GeometryPrefetcher(model.precompute_geometry, device_scans)):
                if args.graphs:
                    loss, rel_pred = graphed(batch)
                    model.update_metrics(batch, rel_pred, split="train")
                else:
                    opt.zero_grad(set_to_none=True)
                    loss = model.training_step(batch, i)
                    loss.backward()
                    opt.step()
                print(f"epoch {epoch} step {i}: loss {float(loss.detach()):.4f}")
        print(json.dumps({k: v for k, v in model.evaluate_predictions(0.0, "train").items() if k != "per_take"}))
        return
    model.eval()
    if args.mode == "evaluate":
        with torch.no_grad():
            total = sum(float(model.validation_step(to_device(s, "cuda"), i)) for i, s in enumerate(scans))
        print(json.dumps({k: v for k, v in model.evaluate_predictions(total, "val").items() if k != "per_take"}))
        return
    results = {}
    with torch.no_grad():
        for i, scan in enumerate(scans):
            scan_id, rels = model.predict_step(to_device(scan, "cuda"), i)
            results[scan_id] = rels
    with open(args.out, "w") as f:                            # same wire format as main.py:111-115
        json.dump(results, f)
    print(f"wrote {args.out}: {sum(len(v) for v in results.values())} triples for {len(results)} scans")


if __name__ == "__main__":
    main()
