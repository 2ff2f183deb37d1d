"""The real-data `evaluate` path on CPU: the reference's prepared-sample cache layout (or_dataset.py:94-120, pre-collate),
`ORDataset` over cache + relationship JSON (dataset_utils.py), and `main.py --mode evaluate` printing rel-F1 — or
"unmeasured" without data / weights (SURVEY.md 7 hard part 6; reference main.py:68-89)."""
import json
import os

import numpy as np
import pytest
import torch

from scene_graph_prediction.scene_graph_helpers.dataset import cache
from scene_graph_prediction.scene_graph_helpers.dataset.or_dataset import ORDataset, class_weights, occurrences
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import OBJECT_NAMES, synthetic_scan

RELS = ["Assisting", "Cementing", "Cleaning", "CloseTo", "Cutting", "Drilling", "Hammering", "Holding", "LyingOn",
        "Operating", "Preparing", "Sawing", "Suturing", "Touching"]


def _reference_layout_sample(scan_id, n_obj, p_obj, p_rel, seed):
    """A sample exactly as the reference's __getitem__ builds and caches it (or_dataset.py:101-120): point-major clouds,
    (E, 2) edges, gt_class as data_preparation returns it."""
    g = torch.Generator().manual_seed(seed)
    E = n_obj * (n_obj - 1)
    edges = torch.tensor([[n, m] for n in range(n_obj) for m in range(n_obj) if n != m], dtype=torch.long)
    return {"scan_id": scan_id, "objs_json": {i + 1: OBJECT_NAMES[i] for i in range(n_obj)}, "instance2mask": {i + 1: i + 1 for i in range(n_obj)},
            "obj_points": torch.rand(n_obj, p_obj, 6, generator=g), "rel_points": torch.rand(E, p_rel, 7, generator=g),
            "gt_class": torch.arange(n_obj, dtype=torch.int64), "gt_rels": torch.randint(0, 15, (E,), generator=g),
            "edge_indices": edges, "relation_objects_one_hot": torch.rand(E, 12, generator=g),
            "rel_hand_points": torch.zeros(E, 2, 3, dtype=torch.float16)}


def test_a_reference_written_cache_loads_collated_and_ours_is_stored_pre_collate(tmp_path):
    ref = _reference_layout_sample("4_000009_1", 2, 30, 40, 0)             # 2 objects: edge_indices is 2 x 2
    np.savez_compressed(str(tmp_path / "4_000009_1.npz"), ref)              # what the reference does (or_dataset.py:120)
    got = cache.cached(tmp_path, "4_000009_1", lambda: (_ for _ in ()).throw(AssertionError("cache miss")))
    assert got["obj_points"].shape == (2, 6, 30) and got["rel_points"].shape == (2, 7, 40)
    assert torch.equal(got["obj_points"], ref["obj_points"].permute(0, 2, 1))
    assert torch.equal(got["edge_indices"], ref["edge_indices"].t()) and got["edge_indices"].tolist() == [[0, 1], [1, 0]]
    assert got["take_idx"] == 4 and got["gt_class"].dtype == torch.int64
    # the other way round: a model-ready scan written here is stored in the reference's layout
    scan = synthetic_scan(3, 50, 60, seed=1, scan_id="8_000001_1")
    cache.save_sample(tmp_path, scan)
    stored = np.load(str(tmp_path / "8_000001_1.npz"), allow_pickle=True)["arr_0"].item()
    assert tuple(stored) == cache.CACHE_KEYS
    assert stored["obj_points"].shape == (3, 50, 6) and stored["rel_points"].shape == (6, 60, 7)
    assert stored["edge_indices"].shape == (6, 2) and stored["edge_indices"][:2].tolist() == [[0, 1], [0, 2]]
    # ... and the reference's own collate_fn (or_dataset.py:63-74, restated) turns it back into the scan
    back = {"obj_points": stored["obj_points"].permute(0, 2, 1), "edge_indices": stored["edge_indices"].t().contiguous()}
    assert torch.equal(back["obj_points"], scan["obj_points"]) and torch.equal(back["edge_indices"], scan["edge_indices"])


def test_occurrences_and_class_weights_known_answers():
    names = sorted(OBJECT_NAMES)
    rels = RELS + ["none"]
    scans = [{"take_idx": 1, "scan": "000001", "objects": {"1": "Patient", "2": "human_0", "3": "instrument"},
              "relationships": [[2, 1, 3, "CloseTo"], [2, 3, 7, "Holding"], [1, 2, 0, "NotAPredicate"]]},
             {"take_idx": 1, "scan": "000002", "objects": {"1": "Patient", "2": "human_0"},
              "relationships": [[2, 1, 3, "CloseTo"], [5, 1, 3, "CloseTo"]]}]       # subject 5 is not in the scan: skipped
    o_obj, o_rel = occurrences(names, rels, scans)
    assert o_obj[names.index("Patient")] == 2 and o_obj[names.index("instrument")] == 1 and o_obj.sum() == 5
    assert o_rel[rels.index("CloseTo")] == 2 and o_rel[rels.index("Holding")] == 1 and o_rel.sum() == 3
    w_obj, w_rel = class_weights(names, rels, scans)
    assert abs(float(w_obj[names.index("Patient")]) - 1 / (np.log(2) + 1)) < 1e-6       # |1 / (log n + 1)|
    assert float(w_obj[names.index("instrument")]) == 1.0
    assert float(w_rel[rels.index("CloseTo")]) == 0.5 and float(w_rel[-1]) == pytest.approx(1e-4)


def _write_dataset(tmp_path, n_scans=3):
    """A 3-scan synthetic 'val' cache (takes 4 and 8) + GT JSON + name files."""
    root, cdir = tmp_path / "data", tmp_path / "cache"
    root.mkdir(); cdir.mkdir()
    (root / "classes.txt").write_text("\n".join(OBJECT_NAMES))
    (root / "relationships.txt").write_text("\n".join(RELS))
    rels = sorted(RELS) + ["none"]
    scans_json = []
    for i in range(n_scans):
        take, scan = (4, 8)[i % 2], f"{i:06d}"
        n_obj = 3 + i % 2
        objects = {str(k + 1): OBJECT_NAMES[k] for k in range(n_obj)}
        relationships = [[1, 2, rels.index("CloseTo"), "CloseTo"], [2, 3, rels.index("Holding"), "Holding"]][: 1 + i % 2]
        scans_json.append({"take_idx": take, "scan": scan, "objects": objects, "relationships": relationships,
                           "human_idx_to_name": {}})
    scans_json.append({"take_idx": 4, "scan": "000077", "objects": {"1": "Patient", "2": "human_0"}, "relationships": []})
    scans_json.append({"take_idx": 2, "scan": "000001", "objects": {"1": "Patient", "2": "human_0", "3": "human_1"},
                       "relationships": []})                                  # test take: not in the val split
    (root / "relationships_validation.json").write_text(json.dumps({"scans": scans_json}))
    ds = ORDataset({"dataset": {}}, "val", cache_dir=cdir, root=str(root), only_cached=False)
    assert len(ds) == n_scans and ds.scans[0] == "4_000000_1"                 # 2-object scan and test take filtered out
    for i, sid in enumerate(ds.scans):
        n_obj = len(ds.objs_json[sid])
        s = synthetic_scan(n_obj, 300, 400, seed=10 + i, scan_id=sid)
        s["objs_json"] = ds.objs_json[sid]
        s["gt_rels"] = ds.gt_rels(sid)
        cache.save_sample(cdir, s)
    return root, cdir, rels


def test_dataset_over_cache_and_gt_json(tmp_path):
    root, cdir, rels = _write_dataset(tmp_path)
    ds = ORDataset({"dataset": {}}, "val", cache_dir=cdir, root=str(root))
    assert ds.relationNames == rels and ds.classNames == sorted(OBJECT_NAMES) and len(ds) == 3
    assert ds.weights_source == "ones"                                        # no training take in the JSON
    # gt_rels: n-major fully connected edges, 'none' unless named (data_preparation_utils.py:127-191)
    g = ds.gt_rels("8_000001_1")                                              # 4 objects, CloseTo 1->2, Holding 2->3
    edges = [(n, m) for n in range(4) for m in range(4) if n != m]
    want = [rels.index("CloseTo") if e == (0, 1) else rels.index("Holding") if e == (1, 2) else rels.index("none") for e in edges]
    assert g.tolist() == want
    batch = next(iter(ds))
    assert batch["obj_points"].shape == (3, 6, 300) and batch["edge_indices"].shape == (2, 6) and batch["take_idx"] == 4
    raw = ds[0]
    assert raw["obj_points"].shape == (3, 300, 6)                             # __getitem__ is pre-collate like the reference
    with pytest.raises(FileNotFoundError):
        ORDataset({"dataset": {}}, "val", cache_dir=tmp_path / "nowhere", root=str(root), only_cached=False)[0]


def test_evaluate_mode_prints_rel_f1_like_the_reference(tmp_path, oracle_backend, capsys):
    """`--mode evaluate --cache-dir D --gt J --weights W` over a 3-scan cache == sklearn on the model's own predictions."""
    from sklearn.metrics import classification_report
    from scene_graph_prediction import main as runner
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    root, cdir, rels = _write_dataset(tmp_path)
    cfg = runner.config_loader("no_gt.json")
    torch.manual_seed(3)
    model = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), rels)
    wpath = str(tmp_path / "w.pth")
    torch.save(model.state_dict(), wpath)
    argv = ["--config", "no_gt.json", "--mode", "evaluate", "--device", "cpu", "--cache-dir", str(cdir),
            "--gt", str(root / "relationships_validation.json"), "--weights", wpath]
    runner.main(argv)
    out = capsys.readouterr().out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["status"] == "measured" and line["scans"] == 3 and line["split"] == "val"
    assert "Take 4" in out and "Take 8" in out and "val Results:" in out      # the reference's printed reports
    # the same numbers straight from sklearn on the model's predictions
    ds = ORDataset(cfg, "val", cache_dir=cdir, root=str(root))
    model.eval()
    gts, preds = [], []
    with torch.no_grad():
        for b in ds:
            _, rel = model(to_device(b, torch.device("cpu")))
            preds += rel.argmax(1).tolist()
            gts += b["gt_rels"].tolist()
    want = classification_report(gts, preds, labels=list(range(15)), target_names=rels, output_dict=True, zero_division=0)
    assert line["rel_f1"] == pytest.approx(want["macro avg"]["f1-score"], abs=1e-12)
    assert line["weighted_f1"] == pytest.approx(want["weighted avg"]["f1-score"], abs=1e-12)
    # no weights: figures of a random model are not a measurement; no data: nothing to measure
    runner.main(argv[:-2])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["rel_f1"] is None and line["status"] == "unmeasured"
    runner.main(["--config", "no_gt.json", "--mode", "evaluate", "--device", "cpu", "--cache-dir", str(tmp_path / "none"),
                 "--gt", str(root / "relationships_validation.json"), "--weights", wpath])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["rel_f1"] is None and line["status"] == "unmeasured" and "does not exist" in line["reason"]


def test_the_in_tree_validation_json_of_the_reference_parses():
    """data/relationships_validation.json of the reference tree (1332 scans) through the dataset's filters — only where
    /root/reference exists (this container); the GPU box skips."""
    root = "/root/reference/data"
    if not os.path.exists(os.path.join(root, "relationships_validation.json")):
        pytest.skip("reference tree not present")
    ds = ORDataset({"dataset": {}}, "val", cache_dir=None, root=root, only_cached=False)
    assert ds.relationNames[-1] == "none" and len(ds.relationNames) == 15 and len(ds.classNames) == 12
    assert 1000 < len(ds) <= 1332 and all(s.endswith("_1") for s in ds.scans)
    sid = next(s for s in ds.scans if ds.relationship_json[s])
    g = ds.gt_rels(sid)
    n = len(ds.objs_json[sid])
    assert g.numel() == n * (n - 1) and int((g != 14).sum()) >= 1


def test_collate_accepts_a_scan_without_edges_and_fresh_tiny_clouds():
    """`torch.tensor([])` edge lists are 1-D (0,) and the reference's collate `.t()` accepts them; a FRESH sample
    whose clouds have 3 / 6 points with fewer points than channels is not a stale cache file."""
    import numpy as np
    import torch
    from scene_graph_prediction.scene_graph_helpers.dataset import cache
    lonely = {"scan_id": "1_000001", "obj_points": np.zeros((1, 40, 6), np.float32), "rel_points": np.zeros((0, 50, 7), np.float32),
              "edge_indices": torch.tensor([]), "gt_class": np.array([3]), "gt_rels": np.zeros((0,), np.int64),
              "relation_objects_one_hot": np.zeros((0, 12), np.float32)}
    out = cache.collate_sample(lonely)
    assert tuple(out["edge_indices"].shape) == (2, 0) and tuple(out["obj_points"].shape) == (1, 6, 40)
    tiny = {"scan_id": "1_000002", "obj_points": np.zeros((2, 3, 6), np.float32), "rel_points": np.zeros((2, 6, 7), np.float32),
            "edge_indices": np.array([[0, 1], [1, 0]]), "_from_cache": False}
    out = cache.collate_sample(tiny)
    assert tuple(out["obj_points"].shape) == (2, 6, 3) and "_from_cache" not in out
    import pytest
    with pytest.raises(ValueError, match="predates"):
        cache.collate_sample(dict(tiny, _from_cache=True))          # the same shapes read from a FILE are refused
