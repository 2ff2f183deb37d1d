"""Synthetic batches with the exact key set / dtypes / layouts that ``ORDataset.collate_fn``
hands to the model (SGH/dataset/or_dataset.py:63-74 and data_preparation_utils.py:110-240;
SURVEY.md §8 A0): the real dataset and its open3d preparation are out of scope, so benchmarks
and tests feed the hot path with clouds of the same shape.

    obj_points               (n_obj, 6, P_obj)   xyz (zero-mean, unit sphere) + rgb in [0,1]
    rel_points               (E, 7, P_rel)       xyz + rgb + instance mask in {0,1,2}
    edge_indices             (2, E) int64        all ordered pairs n != m, n-major
    relation_objects_one_hot (E, 12)
    gt_class (n_obj,), gt_rels (E,) int64; objs_json {1-based id: name}; scan_id; take_idx
"""
import torch

# the 12 object classes of data/classes.txt (sorted, dataset_utils.py:206-207)
OBJECT_NAMES = ["Patient", "anesthesia_equipment", "human_0", "human_1", "human_2", "human_3", "human_4", "human_5",
                "instrument", "instrument_table", "operating_table", "secondary_table"]


def _unit_cloud(n, p, gen):
    x = torch.randn(n, p, 3, generator=gen)
    x = x / x.norm(dim=2, keepdim=True) * torch.rand(n, p, 1, generator=gen).pow(1.0 / 3.0)
    x = x - x.mean(dim=1, keepdim=True)                       # zero_mean (data_preparation_utils.py:12-18)
    return x / x.norm(dim=2).amax(dim=1).view(n, 1, 1)


def fully_connected_edges(n_obj):
    pairs = [(a, b) for a in range(n_obj) for b in range(n_obj) if a != b]      # :130-133
    return torch.tensor(pairs, dtype=torch.int64).t().contiguous()


def synthetic_scan(n_obj=9, points_obj=4000, points_rel=8000, num_class=12, num_rel=15, seed=0, scan_id="synthetic_000000"):
    gen = torch.Generator().manual_seed(seed)
    edges = fully_connected_edges(n_obj)
    E = edges.size(1)
    obj = torch.cat([_unit_cloud(n_obj, points_obj, gen), torch.rand(n_obj, points_obj, 3, generator=gen)], dim=2)
    rel = torch.cat([_unit_cloud(E, points_rel, gen), torch.rand(E, points_rel, 3, generator=gen),
                     torch.randint(0, 3, (E, points_rel, 1), generator=gen).float()], dim=2)
    one_hot = torch.zeros(E, 12)
    one_hot[torch.arange(E), torch.randint(0, 6, (E,), generator=gen)] = 1
    one_hot[torch.arange(E), 6 + torch.randint(0, 6, (E,), generator=gen)] = 1
    return {
        "scan_id": scan_id, "take_idx": 0,
        "obj_points": obj.permute(0, 2, 1).contiguous(),            # channel-first like collate_fn
        "rel_points": rel.permute(0, 2, 1).contiguous(),
        "edge_indices": edges,
        "relation_objects_one_hot": one_hot,
        "gt_class": torch.randint(0, num_class, (n_obj,), generator=gen),
        "gt_rels": torch.randint(0, num_rel, (E,), generator=gen),
        "objs_json": {i + 1: OBJECT_NAMES[i % len(OBJECT_NAMES)] for i in range(n_obj)},
    }


def to_device(batch, device):
    return {k: (v.to(device) if (torch.is_tensor(v) or hasattr(v, "node_ptr")) else v) for k, v in batch.items()}


def collate_scans(scans):
    """Several scans -> ONE block-diagonal batch (MI355X-first: the reference's DataLoader has batch_size=1,
    main.py:54-56, and leaves 90 % of the GPU idle on 9 + 72 small clouds).  Clouds are concatenated along the batch
    axis, `edge_indices` are offset by each scan's first node row, and `scenes` (a SceneBatch) records the node / edge
    row ranges so that the GCN's BatchNorm statistics and the loss average stay PER SCAN.  Per-scan metadata becomes
    lists (`scan_ids`, `objs_jsons`, `take_idxs`)."""
    from scene_graph_prediction.scene_graph_helpers.model.gcns.network_TripletGCN import SceneBatch
    node_ptr, edge_ptr = [0], [0]
    for s in scans:
        node_ptr.append(node_ptr[-1] + s["obj_points"].size(0))
        edge_ptr.append(edge_ptr[-1] + s["rel_points"].size(0))
    batch = {k: torch.cat([s[k] for s in scans], dim=0) for k in
             ("obj_points", "rel_points", "relation_objects_one_hot", "gt_class", "gt_rels")}
    batch["edge_indices"] = torch.cat([s["edge_indices"] + off for s, off in zip(scans, node_ptr)], dim=1).contiguous()
    batch["scenes"] = SceneBatch(torch.tensor(node_ptr), torch.tensor(edge_ptr))
    batch["scan_ids"] = [s["scan_id"] for s in scans]
    batch["objs_jsons"] = [s["objs_json"] for s in scans]
    batch["take_idxs"] = [s.get("take_idx", 0) for s in scans]
    if all("full_image_features" in s for s in scans):
        batch["full_image_features"] = torch.stack([s["full_image_features"] for s in scans])     # (S, 6, F)
    return batch
