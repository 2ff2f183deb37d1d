"""The reference's per-scan sample cache (SGH/dataset/or_dataset.py:94-120): ``np.savez_compressed(path, sample)`` of ONE
python dict per scan — read back with ``np.load(path, allow_pickle=True)['arr_0'].item()`` — holding the prepared crops so
that ``data_preparation`` (open3d, ~seconds per scan) runs once per scan.  Same file format and keys, so caches written by
the reference load here and the other way round:

    scan_id, objs_json, instance2mask, obj_points, rel_points, gt_class, gt_rels, edge_indices,
    relation_objects_one_hot, rel_hand_points

(or_dataset.py:101-118).  Tensors are stored as the torch CPU tensors / numpy arrays they are in the sample; keys this
build does not produce (``instance2mask``, ``rel_hand_points``: hand locations for the augmentations) are written as None
unless the caller supplies them."""
import os
from pathlib import Path
from typing import Callable, Dict, Optional

import numpy as np
import torch

CACHE_KEYS = ("scan_id", "objs_json", "instance2mask", "obj_points", "rel_points", "gt_class", "gt_rels", "edge_indices",
              "relation_objects_one_hot", "rel_hand_points")           # or_dataset.py:101-118


def cache_path(folder, scan_id: str) -> Path:
    return Path(folder) / f"{scan_id}.npz"                             # or_dataset.py:94


def save_sample(folder, sample: Dict) -> Path:
    """Write `sample` in the reference's cache format (tensors moved to the host; GPU-side extras dropped)."""
    path = cache_path(folder, sample["scan_id"])
    os.makedirs(path.parent, exist_ok=True)
    out = {}
    for k in CACHE_KEYS:
        v = sample.get(k)
        out[k] = v.detach().cpu() if torch.is_tensor(v) else v
    np.savez_compressed(str(path), out)                                # or_dataset.py:120: one pickled dict under 'arr_0'
    return path


def load_sample(folder, scan_id: str) -> Optional[Dict]:
    path = cache_path(folder, scan_id)
    if not path.exists():
        return None
    return np.load(str(path), allow_pickle=True)["arr_0"].item()       # or_dataset.py:96


def cached(folder, scan_id: str, prepare: Callable[[], Dict], device=None) -> Dict:
    """or_dataset.py:94-120: the cached sample if there is one, else `prepare()` (e.g. gpu_preparation.prepare_scan on the
    resident scan) written to the cache.  `device`: move the tensors there (the model wants them on the GPU)."""
    sample = load_sample(folder, scan_id)
    if sample is None:
        sample = prepare()
        sample.setdefault("scan_id", scan_id)
        save_sample(folder, sample)
        sample = {k: sample.get(k) for k in CACHE_KEYS} | {k: v for k, v in sample.items() if k in ("take_idx",)}
    if device is not None:
        sample = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}
    return sample
