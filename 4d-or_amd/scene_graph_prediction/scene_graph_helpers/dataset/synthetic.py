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
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
