"""``ORDataset`` over the reference's prepared-sample cache: what ``main.py --mode evaluate`` needs to turn a cache folder,
the ground-truth relationship JSON and a checkpoint into rel-F1 (SURVEY.md 7 hard part 6).

Counterpart of SGH/dataset/or_dataset.py:16-132 + dataset_utils.py:14-64,196-287 for the part that is data handling and
not open3d: the class / relationship name lists (``classes.txt`` / ``relationships.txt`` + ``'none'``, sorted), the scan list of
a split (``TAKE_SPLIT``, helpers/configurations.py:47; a scan needs >= 3 objects of known classes, dataset_utils.py:232-241),
the ``{scan_id}_{split}`` ids, ``objs_json`` / ``relationship_json`` per scan, the class weights of the two NLL terms
(``get_weights`` + data_processing/compute_weight_occurrences.py) and ``collate_fn``.  The samples themselves come from
the ``.npz`` cache the reference writes on its first pass over a scan (or_dataset.py:94-120; ``dataset/cache.py`` — same
files); a cache miss calls ``prepare(scan_id, objs_json, rel_json)`` if one was given (e.g. ``gpu_preparation.prepare_scan`` on
the fused scan) and raises otherwise: cropping from the raw ``.pcd`` files + instance labels is the reference's open3d
code and out of scope.

Differences, stated: the reference lists a split's scans from the ``.pcd`` files under ``datasets/4D-OR``
(dataset_utils.py:33-37); here the scans of a split are those of the JSON whose take is in the split (and, with
``only_cached=True``, whose sample is in the cache folder).  The train JSON is not in the reference tree: without it the
class weights are ones (they scale the reported loss, never a prediction or F1)."""
import json
import os
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import cache

TAKE_SPLIT = {"train": [1, 3, 5, 7, 9, 10], "val": [4, 8], "test": [2, 6]}      # helpers/configurations.py:47
_SPLIT_ID = {"train": 0, "val": 1, "test": 2}                                   # dataset_utils.py:218-223
_JSON_NAMES = ("relationships_train.json", "relationships_validation.json", "relationships_test_dummy.json")


def read_lines(path) -> List[str]:
    with open(path) as f:                                           # utils/util.py:9-15, :32-38
        return [line.rstrip() for line in f]


def occurrences(class_names, relation_names, scans) -> (np.ndarray, np.ndarray):
    """compute_weight_occurrences.compute (:25-103): how often every object class / predicate occurs in `scans`."""
    o_obj, o_rel = np.zeros(len(class_names)), np.zeros(len(relation_names))
    for scan in scans:
        names = {int(k): v for k, v in scan["objects"].items()}
        for v in names.values():
            o_obj[class_names.index(v)] += 1
        for r in scan["relationships"]:
            if r[3] not in relation_names:
                continue
            if r[0] == 0 or r[1] == 0:
                raise RuntimeError("found obj or sub is 0")
            if r[0] not in names or r[1] not in names:
                continue
            o_rel[relation_names.index(r[3])] += 1
    return o_obj, o_rel


def class_weights(class_names, relation_names, scans):
    """dataset_utils.get_weights (:259-270): log weighting for objects, linear for predicates, 1e-4 for 'none'."""
    o_obj, o_rel = occurrences(class_names, relation_names, scans)
    with np.errstate(divide="ignore"):
        w_obj = torch.abs(1.0 / (torch.log(torch.from_numpy(o_obj).float()) + 1))
        w_rel = 1.0 / torch.from_numpy(o_rel).float()
    w_rel[-1] = 0.0001
    return w_obj, w_rel


class ORDataset:
    """``ORDataset(config, split)`` of the reference, fed from a cache folder + relationship JSON files.

    root        folder holding classes.txt / relationships.txt (+ the relationship JSONs found there), like
                ``config['dataset']['root'][0]`` = ``data/`` of the reference tree
    gt_files    relationship JSON files to use instead of / on top of those under `root`
    cache_dir   folder of ``{scan_id}.npz`` samples (``datasets/4D-OR/scene_graph_cache{suffix}`` in the reference)
    """

    def __init__(self, config, split="val", cache_dir=None, root=None, gt_files: Optional[Sequence[str]] = None,
                 only_cached=True, prepare: Optional[Callable] = None, class_names=None, relation_names=None):
        assert split in ("train", "val", "test")
        self.config, self.split, self.prepare = config, split, prepare
        self.mconfig = config.get("dataset", {})
        self.caching_folder = None if cache_dir is None else Path(cache_dir)
        root = root if root is not None else (self.mconfig.get("root") or [None])[0]
        if class_names is None:
            class_names = read_lines(os.path.join(root, "classes.txt"))
        if relation_names is None:
            relation_names = read_lines(os.path.join(root, "relationships.txt"))
        self.classNames = sorted(class_names)
        self.relationNames = sorted(relation_names)
        if "none" not in self.relationNames:
            self.relationNames.append("none")                       # dataset_utils.py:207-211
        files = [str(f) for f in (gt_files or [])]
        if root is not None:
            files += [os.path.join(root, n) for n in _JSON_NAMES if os.path.exists(os.path.join(root, n))]
        seen, self.data = set(), {"scans": []}
        for f in files:
            if os.path.abspath(f) in seen:
                continue
            seen.add(os.path.abspath(f))
            with open(f) as fh:
                self.data["scans"] += json.load(fh)["scans"]
        self.relationship_json, self.objs_json, self.scans = {}, {}, []
        in_split = [s for s in self.data["scans"] if s["take_idx"] in TAKE_SPLIT[split]]
        for scan in in_split:                                        # dataset_utils.get_relationships (:214-250)
            objects = {int(k): v for k, v in scan["objects"].items()}
            if sum(v in self.classNames for v in objects.values()) < 3:
                continue
            sid = f'{scan["take_idx"]}_{scan["scan"]}_{_SPLIT_ID[split]}'
            if sid in self.objs_json:
                continue
            if only_cached and (self.caching_folder is None or not cache.cache_path(self.caching_folder, sid).exists()):
                continue
            self.relationship_json[sid] = [list(r) for r in scan["relationships"]]
            self.objs_json[sid] = objects
            self.scans.append(sid)
        train = [s for s in self.data["scans"] if s["take_idx"] in TAKE_SPLIT["train"]]
        #: 'train json' = from the training takes like the reference's train_dataset.w_cls_* (main.py:74-76); 'ones' when
        #: no training scan is in the JSON files (the train JSON is not part of the reference tree)
        self.weights_source = "train json" if train else "ones"
        if train:
            self.w_cls_obj, self.w_cls_rel = class_weights(self.classNames, self.relationNames, train)
        else:
            self.w_cls_obj, self.w_cls_rel = torch.ones(len(self.classNames)), torch.ones(len(self.relationNames))

    def __len__(self):
        return len(self.scans)

    def gt_rels(self, scan_id) -> torch.Tensor:
        """The per-edge predicate ids `data_preparation` builds (data_preparation_utils.py:139-191) for the fully connected
        n-major edge list over the scan's objects in sorted id order: 'none' unless the JSON names a relationship."""
        ids = sorted(self.objs_json[scan_id])
        pos = {inst: i for i, inst in enumerate(ids)}
        none_id = self.relationNames.index("none")
        adj = np.full((len(ids), len(ids)), none_id, dtype=np.int64)
        for r in self.relationship_json[scan_id]:
            if r[0] in pos and r[1] in pos and r[3] in self.relationNames:
                adj[pos[r[0]], pos[r[1]]] = self.relationNames.index(r[3])
        return torch.tensor([adj[n, m] for n in range(len(ids)) for m in range(len(ids)) if n != m], dtype=torch.int64)

    def __getitem__(self, index) -> Dict:
        """The PRE-collate sample, like the reference's __getitem__ (cache hit: or_dataset.py:95-96)."""
        sid = self.scans[index]
        sample = None if self.caching_folder is None else cache.load_raw(self.caching_folder, sid)
        if sample is None:
            if self.prepare is None:
                raise FileNotFoundError(
                    f"{sid}: no cached sample under {self.caching_folder} and no `prepare` callback (cropping the fused "
                    "scan is the reference's open3d preparation; run the reference once to fill the cache or pass prepare=)")
            ready = self.prepare(sid, self.objs_json[sid], self.relationship_json[sid])
            ready.setdefault("scan_id", sid)
            ready.setdefault("objs_json", self.objs_json[sid])
            if self.caching_folder is not None:
                cache.save_sample(self.caching_folder, ready)
            sample = cache.uncollate_sample(ready)
            sample["_from_cache"] = False            # fresh: the stale-file heuristic of collate_sample does not apply
        return sample

    def collate_fn(self, batch):
        return cache.collate_sample(batch[0])                        # or_dataset.py:63-74 (batch_size = 1)

    def __iter__(self):
        for i in range(len(self)):
            yield self.collate_fn([self[i]])
