"""CPU checks of the (f)3 pieces that need no GPU: the numpy restatement of the voxel trace against hand-derived known
answers (open3d is absent: these are what pins it), and the reference's .npz sample-cache format."""
import numpy as np
import torch

import prep_oracle


def test_voxel_trace_known_answers():
    # size 10, min_bound (0,0,0): voxel origin (-5,-5,-5); a point p lies in voxel floor((p+5)/10), upper octant half when
    # the fractional part >= 0.5
    pts = np.array([[0.0, 0.0, 0.0],      # ref 0.5 -> voxel (0,0,0), octant 7 (all three >= 0.5)
                    [4.9, 0.0, 0.0],      # ref x 0.99 -> voxel 0, octant 7
                    [5.0, 0.0, 0.0],      # ref x 1.0  -> voxel (1,0,0), x lower half: octant 6
                    [12.0, 0.0, 0.0],     # ref x 1.7  -> voxel (1,0,0), octant 7
                    [0.0, 0.0, 0.0]],     # duplicate of point 0: the LAST index wins the slot
                   dtype=np.float32)
    tr = prep_oracle.voxel_trace(pts, 10.0, pts.min(0))
    assert tr.shape == (2, 8)
    rows = {tuple(r) for r in tr.tolist()}
    assert (-1, -1, -1, -1, -1, -1, -1, 4) in rows            # voxel 0: points 0, 1, 4 in octant 7 -> 4
    assert (-1, -1, -1, -1, -1, -1, 2, 3) in rows             # voxel 1: point 2 in octant 6, point 3 in octant 7
    assert np.unique(tr)[1:].tolist() == [2, 3, 4]            # what calculate_downsample_indices keeps (:44)


def test_ladder_picks_the_coarsest_rung_above_the_target():
    rng = np.random.default_rng(0)
    pts = (rng.random((6000, 3)) * np.array([900.0, 600.0, 400.0])).astype(np.float32)      # millimetres
    best, rung = prep_oracle.downsample_candidates(pts, 1000)
    assert rung >= 0 and len(best) > 1000 and np.all(np.diff(best) > 0)
    nxt = np.unique(prep_oracle.voxel_trace(pts, 15 + 5 * (rung + 1), pts.min(0)))[1:] if rung + 1 < 17 else []
    assert len(nxt) <= 1000                                   # the next rung is the one that stopped the ladder
    # a cloud the first rung already thins below the target keeps every point (:41)
    few, r2 = prep_oracle.downsample_candidates(pts[:50] * 0.001, 40)
    assert r2 == -1 and len(few) == 50


def test_sample_cache_has_the_reference_format(tmp_path):
    from scene_graph_prediction.scene_graph_helpers.dataset import cache
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan
    scan = synthetic_scan(3, 50, 60, seed=1, scan_id="4_000131")
    calls = []

    def prepare():
        calls.append(1)
        return dict(scan)

    a = cache.cached(tmp_path, "4_000131", prepare)
    b = cache.cached(tmp_path, "4_000131", prepare)
    assert len(calls) == 1                                    # second call is a cache hit (or_dataset.py:95-96)
    raw = np.load(str(tmp_path / "4_000131.npz"), allow_pickle=True)
    assert raw.files == ["arr_0"]                             # np.savez_compressed(path, sample): one pickled dict
    stored = raw["arr_0"].item()
    assert tuple(stored) == cache.CACHE_KEYS
    for k in ("obj_points", "rel_points", "edge_indices", "gt_class", "gt_rels", "relation_objects_one_hot"):
        assert torch.equal(torch.as_tensor(a[k]), scan[k]) and torch.equal(torch.as_tensor(b[k]), scan[k])   # model-ready both times
    # on disk: the reference's PRE-collate layout (or_dataset.py:101-118 before collate_fn :63-74)
    assert torch.equal(stored["obj_points"], scan["obj_points"].permute(0, 2, 1))
    assert torch.equal(stored["rel_points"], scan["rel_points"].permute(0, 2, 1))
    assert torch.equal(stored["edge_indices"], scan["edge_indices"].t())
    for k in ("gt_class", "gt_rels", "relation_objects_one_hot"):
        assert torch.equal(stored[k], scan[k])
    assert stored["scan_id"] == "4_000131" and stored["objs_json"] == scan["objs_json"] and a["scan_id"] == b["scan_id"]
