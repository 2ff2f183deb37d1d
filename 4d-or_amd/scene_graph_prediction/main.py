"""Thin runner for the scene-graph model on MI355X.

Counterpart of scene_graph_prediction/main.py (:36-117: ``--config``, modes train / evaluate /
infer, JSON dump of ``{scan_id: [[sub, pred, obj], ...]}`` at :111-115) without the
pytorch_lightning harness and without the 4D-OR dataset (neither ships with the reference
tree).  It drives ``SGPNModelWrapper`` on synthetic scans of the real
shape so the whole path — encoders, TripletGCN, heads, loss, triple emission, wire format —
runs end to end, and loads the paper checkpoints unchanged when given ``--weights``.

    python -m scene_graph_prediction.main --config no_gt.json --mode infer --scans 4 --out scan_relations.json

rel-F1 on real data is one command away (``evaluate_split``; the reference's main.py:68-89):

    python -m scene_graph_prediction.main --config no_gt.json --mode evaluate \\
        --cache-dir datasets/4D-OR/scene_graph_cache_no_gt --gt data/relationships_validation.json \\
        --weights scene_graph_prediction/scene_graph_helpers/paper_weights/paper_model_no_gt_no_images.pth

prints the per-take / overall classification reports and one JSON line with ``rel_f1``; with no dataset or no weights the
line reads ``"rel_f1": null, "status": "unmeasured"``.
"""
import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (what the host driver supports): must be in the
                                                            # environment before the first HIP call of the process
import torch

from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import OBJECT_NAMES, synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

RELATION_NAMES = ["Assisting", "Cementing", "Cleaning", "CloseTo", "Cutting", "Drilling", "Hammering", "Holding",
                  "LyingOn", "Operating", "Preparing", "Sawing", "Suturing", "Touching", "none"]   # data/relationships.txt + 'none'


def config_loader(path):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_graph_helpers", "configs")
    full = path if os.path.exists(path) else os.path.join(here, path)
    with open(full) as f:
        return json.load(f)


def setup_distributed(device: torch.device):
    """(rank, world).  Under torchrun (WORLD_SIZE > 1) one process per GPU: RCCL (`nccl` backend on ROCm) over xGMI for
    cuda devices, gloo for the CPU tests.  The reference trains on ONE GPU (main.py:62 `gpus=1`); scans are independent,
    so the data-parallel run shards them by rank and averages the gradients — nothing else is exchanged."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        dist.init_process_group("nccl" if device.type == "cuda" else "gloo")
    return dist.get_rank(), world


def train(model, config, scans, device, epochs=1, graphs=False, rank=0, world=1, log=print):
    """The train mode of the runner: one scan per step like main.py:54-56.  world > 1: rank r takes scans r, r + world, ...;
    every parameter's .grad is a view of one flat buffer (runtime.FlatGrads) that is averaged with ONE all-reduce between
    backward and optimizer step; the classification heads the encoders inherit but never call are frozen —
    they would otherwise sit in the buffer as zeros and take AdamW's weight decay; rank 0's initial weights are broadcast.
    len(scans) need not be a multiple of world: the incomplete last round is dropped (a rank with one scan more would wait
    forever in its extra all-reduce).  Returns the list of (step, loss) of this rank."""
    import torch.distributed as dist
    from runtime import FlatGrads, GeometryPrefetcher, GraphedTrainStep, ScheduledGC
    model.train()
    flat = None
    if graphs or world > 1:
        for n, p in model.named_parameters():
            if ".backbone.fc_layer." in n:
                p.requires_grad_(False)
        live = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(live, lr=model.lr, weight_decay=float(config["W_DECAY"]), capturable=bool(graphs),
                                fused=bool(live) and all(p.is_cuda for p in live))
    else:
        opt = model.configure_optimizers(capturable=False)
    if world > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    graphed = None
    if graphs:
        graphed = GraphedTrainStep(model.pure_training_step, model.parameters(), opt)
    elif world > 1:
        flat = FlatGrads(model.parameters())
    # every rank must run the same number of steps (one all-reduce per step): the tail that does not fill a round of
    # `world` scans is dropped, like DistributedSampler(drop_last=True)
    usable = len(scans) // world * world
    if usable == 0:
        raise ValueError(f"{len(scans)} scans cannot be sharded over {world} ranks: every rank needs at least one")
    if usable != len(scans) and rank == 0:
        log(f"[rank 0] {len(scans) - usable} of {len(scans)} scans dropped so that all {world} ranks run {usable // world} steps")
    mine = scans[:usable][rank::world]
    history = []
    # cyclic-GC passes on the loop's schedule (runtime/gc_schedule.py): an automatic generation-2 pass stops the enqueueing
    # thread for longer than its lead over the GPU
    with ScheduledGC(every=200) as sgc:
        for epoch in range(epochs):
            device_scans = (to_device(scan, device) for scan in mine)
            # the next scan's sampling geometry is prefetched on a side stream (GPU only)
            it = GeometryPrefetcher(model.precompute_geometry, device_scans) if device.type == "cuda" else device_scans
            for i, batch in enumerate(it):
                if graphed is not None:
                    loss, rel_pred = graphed(batch)
                    model.update_metrics(batch, rel_pred, split="train")
                else:
                    if flat is not None:
                        flat.zero_()
                    else:
                        opt.zero_grad(set_to_none=True)
                    loss = model.training_step(batch, i)
                    loss.backward()
                    if flat is not None:
                        flat.all_reduce_mean()
                    opt.step()
                sgc.step()
                history.append((epoch * len(mine) + i, float(loss.detach())))
                log(f"[rank {rank}] epoch {epoch} step {i}: loss {history[-1][1]:.4f}")
    return history


def evaluate_split(config, args, device, log=print):
    """main.py:68-89 of the reference — val split, `load_state_dict(paper_weight)`, `trainer.validate` — on the reference's
    own prepared-sample cache: every scan of the split through `validation_step`, then the per-take and overall sklearn
    reports exactly as `evaluate_predictions` prints them (scene_graph_prediction_model.py:195-238) and ONE JSON line whose
    `rel_f1` is the macro F1 the paper quotes.  Without data (no cache folder / no sample of the split in it) or without
    weights the line says so: `"rel_f1": null, "status": "unmeasured"`."""
    from scene_graph_prediction.scene_graph_helpers.dataset.or_dataset import ORDataset
    gt = list(args.gt or [])
    root = args.data_root or (os.path.dirname(os.path.abspath(gt[0])) if gt else None)
    line = {"metric": "rel_f1 (macro F1 over the predicates, sklearn classification_report)", "split": args.split,
            "config": args.config, "weights": args.weights, "cache_dir": args.cache_dir, "gt": gt}

    def unmeasured(reason):
        log(json.dumps(line | {"rel_f1": None, "status": "unmeasured", "reason": reason}))

    if not gt and root is None:
        return unmeasured("no relationship JSON (--gt) given")
    if args.cache_dir is None or not os.path.isdir(args.cache_dir):
        return unmeasured(f"cache folder {args.cache_dir!r} does not exist")
    names_ok = root is not None and all(os.path.exists(os.path.join(root, n)) for n in ("classes.txt", "relationships.txt"))
    ds = ORDataset(config, args.split, cache_dir=args.cache_dir, root=root if names_ok else None, gt_files=gt,
                   class_names=None if names_ok else OBJECT_NAMES, relation_names=None if names_ok else RELATION_NAMES[:-1])
    if len(ds) == 0:
        return unmeasured(f"no prepared sample of the {args.split} split under {args.cache_dir}")
    model = SGPNModelWrapper(config, len(ds.classNames), len(ds.relationNames), ds.w_cls_obj, ds.w_cls_rel,
                             ds.relationNames).to(device)
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location=device))
    model.eval()
    total, n = 0.0, 0
    with torch.no_grad():
        for i, batch in enumerate(ds):
            if args.max_scans and i >= args.max_scans:
                break
            total += float(model.validation_step(to_device(batch, device), i))
            n += 1
    res = model.evaluate_predictions(total, "val", print_reports=True, log=log)
    status = "measured" if args.weights else "unmeasured"
    out = line | {"rel_f1": res["macro_f1"] if args.weights else None, "status": status, "scans": n,
                  "scans_in_split": len(ds), "class_weights": ds.weights_source,
                  **{k: v for k, v in res.items() if k != "per_take"}}
    if not args.weights:
        out["reason"] = "no --weights: the figures below are those of a randomly initialised model"
    log(json.dumps(out))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="no_gt.json")
    ap.add_argument("--mode", choices=["train", "evaluate", "infer"], default="infer")
    ap.add_argument("--scans", type=int, default=2, help="synthetic scans to process")
    ap.add_argument("--objects", type=int, default=9)
    ap.add_argument("--weights", type=str, default=None, help="state_dict (.pth) of the reference model")
    ap.add_argument("--out", type=str, default="scan_relations_synthetic.json")
    ap.add_argument("--cache-dir", type=str, default=None,
                    help="evaluate: folder of the reference's prepared samples ({scan_id}.npz, or_dataset.py:94-120), e.g. "
                         "datasets/4D-OR/scene_graph_cache_no_gt")
    ap.add_argument("--gt", type=str, action="append", default=None,
                    help="evaluate: relationship JSON (data/relationships_validation.json); repeat for the train JSON, "
                         "which supplies the class weights of the loss")
    ap.add_argument("--data-root", type=str, default=None,
                    help="evaluate: folder with classes.txt / relationships.txt (default: next to the first --gt file)")
    ap.add_argument("--split", choices=["train", "val", "test"], default="val")
    ap.add_argument("--max-scans", type=int, default=0, help="evaluate: stop after this many scans (0 = the whole split)")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--graphs", action="store_true",
                    help="train mode: replay each scan shape's whole step (fwd + loss + bwd + AdamW) as one hipGraph")
    ap.add_argument("--device", type=str, default="cuda",
                    help="cuda (default; under torchrun: cuda:LOCAL_RANK) | cpu (tests: needs a CPU backend in pointnet2_utils._ext)")
    args = ap.parse_args(argv)

    torch.manual_seed(42)                                     # main.py:40 seeds everything with 42
    config = config_loader(args.config)
    dcfg = config["dataset"]
    if args.device == "cuda" and "LOCAL_RANK" in os.environ:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    device = torch.device(args.device if args.device != "cuda" else f"cuda:{torch.cuda.current_device()}")
    rank, world = setup_distributed(device) if args.mode == "train" else (0, 1)
    if device.type == "cuda":
        # one process per GPU: the enqueueing thread next to its GPU's NUMA node (runtime/affinity.py; PN2_PIN_NUMA=0: off)
        from runtime.affinity import pin_to_gpu_numa
        # the GPU is the SELECTED device (--device cuda:3 in a single-process run), not the local rank's ordinal; with
        # HIP_VISIBLE_DEVICES narrowing every rank to one device the peer lookup finds nothing and the even-slice fallback
        # applies, which never goes below affinity.MIN_CORES cores (num_workers + the RCCL threads inherit the mask)
        pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                        device_index=device.index if device.index is not None else torch.cuda.current_device(),
                        min_cores=max(4, int(config.get("NUM_WORKERS", 0)) + 2))
    if args.mode == "evaluate" and (args.cache_dir or args.gt):
        evaluate_split(config, args, device)
        return
    model = SGPNModelWrapper(config, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)),
                             RELATION_NAMES).to(device)
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location=device))
    scans = [synthetic_scan(args.objects, dcfg["num_points_objects"], dcfg["num_points_relation"], seed=i,
                            scan_id=f"synthetic_{i:06d}") for i in range(args.scans)]

    if args.mode == "train":
        train(model, config, scans, device, epochs=args.epochs, graphs=args.graphs, rank=rank, world=world)
        if rank == 0:
            print(json.dumps({k: v for k, v in model.evaluate_predictions(0.0, "train").items() if k != "per_take"}))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    model.eval()
    if args.mode == "evaluate":
        with torch.no_grad():
            total = sum(float(model.validation_step(to_device(s, device), i)) for i, s in enumerate(scans))
        print(json.dumps({k: v for k, v in model.evaluate_predictions(total, "val").items() if k != "per_take"}))
        return
    results = {}
    with torch.no_grad():
        for i, scan in enumerate(scans):
            scan_id, rels = model.predict_step(to_device(scan, device), i)
            results[scan_id] = rels
    with open(args.out, "w") as f:                            # same wire format as main.py:111-115
        json.dump(results, f)
    print(f"wrote {args.out}: {sum(len(v) for v in results.values())} triples for {len(results)} scans")


if __name__ == "__main__":
    main()
