"""GPU data preparation: a fused OR scan + instance mask -> the batch dict the model consumes.

Counterpart of ``data_preparation`` (SGH/dataset/data_preparation_utils.py:52-240) and ``ORDataset.collate_fn``
(SGH/dataset/or_dataset.py:63-74) without open3d and without the host: per-object crops (instance mask, :110-125),
per-pair crops (strict inside test against the union of the two padded boxes, mask channel 1 = subject / 2 = object,
:173-224), down / up-sampling to ``num_points_objects`` / ``num_points_relation`` (:37-49) and ``zero_mean`` (:12-18) run
as four HIP kernels (csrc/data_prep.hip) on the scan already resident in HBM.  Edges are all ordered pairs n != m,
n-major (:127-133); the subject / object one-hot follows ``objname_to_onehot`` (:21-34).

What is and is not identical to the reference: boxes, filters, mask channel, edge order, one-hot and zero_mean are
restated exactly (tests compare against a numpy restatement of those lines).  The sub-sampling cannot be: the
reference draws from numpy's global generator on top of an open3d voxel trace.  Here it is a seeded counter-based
sampler with the same two regimes (fewer members than the target: draws with replacement; more: distinct members, one
per stratum of the member order) — statistical, not bitwise, parity.
"""
from typing import Dict, List, Optional, Sequence

import torch

from pointnet2_ops import _ext

OBJ_NAME_TO_INDEX = {"anesthesia_equipment": 0, "operating_table": 1, "instrument_table": 2, "secondary_table": 3,
                     "instrument": 4, "human": 5}            # data_preparation_utils.py:22-29


def objname_to_onehot(name: str) -> torch.Tensor:
    if "human" in name or "Patient" in name:                 # :30-31
        name = "human"
    v = torch.zeros(len(OBJ_NAME_TO_INDEX))
    v[OBJ_NAME_TO_INDEX[name]] = 1
    return v


def fully_connected_edges(n_obj: int) -> torch.Tensor:
    pairs = [(a, b) for a in range(n_obj) for b in range(n_obj) if a != b]      # :127-133
    return torch.tensor(pairs, dtype=torch.int64).t().contiguous() if pairs else torch.zeros(2, 0, dtype=torch.int64)


def prepare_scan(points: torch.Tensor, masks: torch.Tensor, n_obj: int, num_points_objects: int = 4000,
                 num_points_relation: int = 8000, padding: float = 0.2, seed: int = 0,
                 object_names: Optional[Sequence[str]] = None, gt_class: Optional[torch.Tensor] = None,
                 gt_rels: Optional[torch.Tensor] = None, scan_id: str = "scan", take_idx: int = 0) -> Dict:
    """points (P, 6) fp32 on the GPU (xyz + rgb in [0,1]), masks (P,) int32 with object ids 1..n_obj (0 = context)
    -> the A0 batch: obj_points (n_obj, 6, T_o), rel_points (E, 7, T_r), edge_indices (2, E) int64,
    relation_objects_one_hot (E, 12) [needs object_names], plus the optional labels passed through."""
    if not points.is_cuda:
        raise RuntimeError("prepare_scan: the scan must be resident on the GPU (there is no host path)")
    points = points.contiguous().float()
    masks = masks.contiguous().to(torch.int32)
    edges64 = fully_connected_edges(n_obj).to(points.device)
    edges = edges64.to(torch.int32).contiguous()
    obj, rel, boxes, sel, counts = _ext.prepare_scan_crops(points, masks, edges, n_obj, num_points_objects,
                                                          num_points_relation, padding, seed)
    batch = {
        "scan_id": scan_id, "take_idx": take_idx,
        "obj_points": obj.permute(0, 2, 1).contiguous(),       # channel-first, like collate_fn (or_dataset.py:67-68)
        "rel_points": rel.permute(0, 2, 1).contiguous(),
        "edge_indices": edges64,
        "prep": {"boxes": boxes, "selection": sel, "members": counts},
    }
    if object_names is not None:
        onehot = torch.stack([torch.cat([objname_to_onehot(object_names[a]), objname_to_onehot(object_names[b])])
                              for a, b in edges64.t().tolist()]) if edges64.size(1) else torch.zeros(0, 12)
        batch["relation_objects_one_hot"] = onehot.to(points.device)
        batch["objs_json"] = {i + 1: n for i, n in enumerate(object_names)}
    if gt_class is not None:
        batch["gt_class"] = gt_class
    if gt_rels is not None:
        batch["gt_rels"] = gt_rels
    return batch


def synthetic_fused_scan(n_obj=9, points=300000, seed=0, device="cuda"):
    """A room-sized cloud with `n_obj` blob-shaped instances + context points, metric coordinates (metres), for tests and
    the end-to-end benchmark (the 4D-OR scans are not in the tree)."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(n_obj, 3, generator=g) * torch.tensor([4.0, 4.0, 1.5]) + torch.tensor([0.5, 0.5, 0.2])
    sizes = torch.rand(n_obj, 3, generator=g) * 0.4 + 0.15
    per_obj = torch.randint(points // (4 * n_obj), points // (2 * n_obj), (n_obj,), generator=g)
    per_obj[0] = 2500                                            # one object below the 4000-point target: up-sampling regime
    xyz, mask = [], []
    for i in range(n_obj):
        n = int(per_obj[i])
        xyz.append(centres[i] + torch.randn(n, 3, generator=g) * sizes[i])
        mask.append(torch.full((n,), i + 1, dtype=torch.int32))
    rest = points - sum(int(v) for v in per_obj)
    xyz.append(torch.rand(rest, 3, generator=g) * torch.tensor([5.0, 5.0, 2.5]))
    mask.append(torch.zeros(rest, dtype=torch.int32))
    xyz, mask = torch.cat(xyz), torch.cat(mask)
    perm = torch.randperm(xyz.size(0), generator=g)
    pts = torch.cat([xyz[perm], torch.rand(xyz.size(0), 3, generator=g)], dim=1)
    return pts.to(device), mask[perm].to(device)
