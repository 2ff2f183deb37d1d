"""The reference's per-scan sample cache (SGH/dataset/or_dataset.py:94-120): ``np.savez_compressed(path, sample)`` of ONE
python dict per scan — read back with ``np.load(path, allow_pickle=True)['arr_0'].item()`` — holding the prepared crops so
that ``data_preparation`` (open3d, ~seconds per scan) runs once per scan.  Same file format, keys AND tensor layout, so
caches written by the reference load here and the other way round:

    scan_id, objs_json, instance2mask, obj_points, rel_points, gt_class, gt_rels, edge_indices,
    relation_objects_one_hot, rel_hand_points

(or_dataset.py:101-118).  What the file holds is the PRE-collate sample of ``ORDataset.__getitem__`` — point-major
``obj_points (n_obj, P, 6)``, ``rel_points (E, P, 7)``, ``edge_indices (E, 2)``, ``gt_class`` as ``data_preparation``
returned it — and ``ORDataset.collate_fn`` (or_dataset.py:63-74) turns it into what the model consumes: channel-first
clouds, ``edge_indices (2, E)``, flat int64 ``gt_class``, ``take_idx`` from the scan id.  The tensors this build works with
(``gpu_preparation.prepare_scan``, ``synthetic.synthetic_scan``) are the POST-collate ones, so ``save_sample`` undoes the
collate before writing and ``load_sample`` / ``cached`` apply it after reading (``collate_sample``).  Keys this build does
not produce (``instance2mask``, ``rel_hand_points``: hand locations for the augmentations) are written as None unless the
caller supplies them."""
import os
from pathlib import Path
from typing import Callable, Dict, Optional

import numpy as np
import torch

CACHE_KEYS = ("scan_id", "objs_json", "instance2mask", "obj_points", "rel_points", "gt_class", "gt_rels", "edge_indices",
              "relation_objects_one_hot", "rel_hand_points")           # or_dataset.py:101-118


def cache_path(folder, scan_id: str) -> Path:
    return Path(folder) / f"{scan_id}.npz"                             # or_dataset.py:94


def take_of(scan_id) -> int:
    """or_dataset.py:72: ``int(scan_id.split('_')[0])``; ids that do not start with a take number (synthetic scans) -> 0."""
    try:
        return int(str(scan_id).split("_")[0])
    except ValueError:
        return 0


#: channel counts a stored cloud can have: xyz | xyz + rgb | xyz + rgb + mask (data_preparation_utils.py:110-125)
_POINT_CHANNELS = (3, 6, 7)


def _check_stored_layout(sample: Dict, from_disk: bool = True) -> None:
    """A cache file holds the reference's PRE-collate layout: clouds (n, P, C) with C in {3, 6, 7} last, `edge_indices`
    (E, 2).  Files written by rounds 1-3 of this build stored the post-collate layout (channel-first clouds, (2, E)
    edges); permuting those a second time gives wrong edges without an error when E == 2, so a stale file is refused
    here instead (delete it: `ORDataset` regenerates the sample)."""
    sid = sample.get("scan_id", "?")
    for k in ("obj_points", "rel_points"):
        v = sample.get(k)
        if v is None:
            continue
        shp = tuple(v.shape)
        # (a cloud of P in {3, 6, 7} POINTS with P < C is not evidence of the old layout when the sample is fresh: the stale-
        # layout heuristic only applies to samples that came from a file — `from_disk`)
        stale = from_disk and shp[1] in _POINT_CHANNELS and shp[1] < shp[2] if len(shp) == 3 else False
        if len(shp) != 3 or shp[2] not in _POINT_CHANNELS or stale:
            raise ValueError(f"cache sample {sid}: `{k}` has shape {shp}, expected the reference's pre-collate layout "
                             f"(n, points, channels in {_POINT_CHANNELS}); the file predates the round-4 cache format — "
                             "delete it so that it is regenerated")
    ei = sample.get("edge_indices")
    if ei is not None and int(torch.as_tensor(ei).numel()) == 0:
        ei = None                    # a scan without edges: torch.tensor([]) is 1-D (0,), and the reference's .t() accepts it
    if ei is not None:
        shp = tuple(ei.shape)
        n_obj = None if sample.get("obj_points") is None else int(sample["obj_points"].shape[0])
        n_edge = None if sample.get("rel_points") is None else int(sample["rel_points"].shape[0])
        bad = len(shp) != 2 or shp[1] != 2 or (n_edge is not None and shp[0] != n_edge)
        if not bad and n_obj is not None and shp[0]:
            bad = int(torch.as_tensor(ei).max()) >= n_obj
        if bad:
            raise ValueError(f"cache sample {sid}: `edge_indices` has shape {shp}, expected (E, 2) with E = "
                             f"{n_edge} pair clouds; the file predates the round-4 cache format — delete it")


def collate_sample(sample: Dict) -> Dict:
    """``ORDataset.collate_fn`` (or_dataset.py:63-74) on one cached (pre-collate) sample; returns a new dict."""
    out = dict(sample)
    _check_stored_layout(out, from_disk=bool(out.pop("_from_cache", True)))
    for k in ("obj_points", "rel_points"):
        if out.get(k) is not None:
            out[k] = torch.as_tensor(out[k]).permute(0, 2, 1).contiguous()      # (n, P, C) -> (n, C, P)
    if out.get("gt_class") is not None:
        out["gt_class"] = torch.as_tensor(out["gt_class"]).flatten().long()
    if out.get("edge_indices") is not None:
        ei_t = torch.as_tensor(out["edge_indices"])
        out["edge_indices"] = (ei_t.reshape(0, 2) if ei_t.numel() == 0 else ei_t).t().contiguous()   # (E, 2) -> (2, E)
    for k in ("gt_rels", "relation_objects_one_hot"):
        if out.get(k) is not None:
            out[k] = torch.as_tensor(out[k])
    out["take_idx"] = take_of(out.get("scan_id", ""))
    return out


def uncollate_sample(sample: Dict) -> Dict:
    """The inverse of ``collate_sample`` for the keys of the cache file: model-ready tensors -> the reference's stored
    layout (host tensors)."""
    out = {}
    for k in CACHE_KEYS:
        v = sample.get(k)
        v = v.detach().cpu() if torch.is_tensor(v) else v
        if v is not None and k in ("obj_points", "rel_points"):
            v = v.permute(0, 2, 1).contiguous()                        # (n, C, P) -> (n, P, C)
        if v is not None and k == "edge_indices":
            v = v.t().contiguous()                                     # (2, E) -> (E, 2)
        out[k] = v
    return out


def save_sample(folder, sample: Dict) -> Path:
    """Write the model-ready `sample` in the reference's cache format (pre-collate layout, tensors on the host; GPU-side
    extras dropped)."""
    path = cache_path(folder, sample["scan_id"])
    os.makedirs(path.parent, exist_ok=True)
    np.savez_compressed(str(path), uncollate_sample(sample))           # or_dataset.py:120: one pickled dict under 'arr_0'
    return path


def load_raw(folder, scan_id: str) -> Optional[Dict]:
    """The stored dict exactly as the reference's ``__getitem__`` reads it (or_dataset.py:96); None on a miss."""
    path = cache_path(folder, scan_id)
    if not path.exists():
        return None
    return np.load(str(path), allow_pickle=True)["arr_0"].item()


def load_sample(folder, scan_id: str) -> Optional[Dict]:
    """The cached scan as the model consumes it (collated); None on a miss."""
    raw = load_raw(folder, scan_id)
    return None if raw is None else collate_sample(raw)


def cached(folder, scan_id: str, prepare: Callable[[], Dict], device=None) -> Dict:
    """or_dataset.py:94-120: the cached sample if there is one, else `prepare()` (e.g. gpu_preparation.prepare_scan on the
    resident scan; model-ready layout) written to the cache.  Either way the result is model-ready.  `device`: move the
    tensors there (the model wants them on the GPU)."""
    sample = load_sample(folder, scan_id)
    if sample is None:
        sample = prepare()
        sample.setdefault("scan_id", scan_id)
        save_sample(folder, sample)
        sample = {k: sample.get(k) for k in CACHE_KEYS} | {"take_idx": sample.get("take_idx", take_of(scan_id))}
    if device is not None:
        sample = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}
    return sample
