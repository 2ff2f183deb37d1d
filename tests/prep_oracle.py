"""CPU restatement (numpy) of the scan preparation — TEST INFRASTRUCTURE.

Deterministic parts follow the reference line by line (SGH/dataset/data_preparation_utils.py): padded object boxes
:113-115, union box + strict inside test :203-208, mask channel :200-202, zero_mean :12-18.  The sub-sampling restates
the product's seeded sampler (csrc/data_prep.hip: prep_mix + strata), because the reference's (open3d voxel trace +
numpy's global generator, :37-49) is not reproducible."""
import numpy as np

M = 0xFFFFFFFF


def mix(seed, a, b):
    h = (seed ^ ((a * 0x9E3779B9) & M) ^ ((b * 0x85EBCA6B) & M)) & M
    h ^= h >> 16; h = (h * 0x7FEB352D) & M
    h ^= h >> 15; h = (h * 0x846CA68B) & M
    h ^= h >> 16
    return h


def zero_mean(xyz):
    xyz = xyz.astype(np.float32)
    mean = xyz.mean(axis=0, dtype=np.float32)
    xyz = xyz - mean
    far = np.sqrt((xyz ** 2).sum(1).max())
    return xyz / far if far > 0 else xyz


def prepare(points, masks, n_obj, t_obj, t_rel, padding, seed):
    points = points.astype(np.float32)
    boxes = []
    for i in range(n_obj):
        sel = points[masks == i + 1, :3]
        boxes.append((sel.min(0) - np.float32(padding), sel.max(0) + np.float32(padding)))
    edges = [(a, b) for a in range(n_obj) for b in range(n_obj) if a != b]
    crops, members = [], []
    for i in range(n_obj):
        members.append(np.where(masks == i + 1)[0])
    for a, b in edges:
        lo, hi = np.minimum(boxes[a][0], boxes[b][0]), np.maximum(boxes[a][1], boxes[b][1])
        inside = np.ones(len(points), dtype=bool)
        for d in range(3):
            inside &= (points[:, d] > lo[d]) & (points[:, d] < hi[d])
        members.append(np.where(inside)[0])
    sel_all, obj_out, rel_out = [], [], []
    for c, mem in enumerate(members):
        target = t_obj if c < n_obj else t_rel
        count = len(mem)
        pick = np.empty(target, dtype=np.int64)
        for t in range(target):
            h = mix(seed, c, t)
            if count < target:
                q = h % count
            else:
                s0, s1 = t * count // target, (t + 1) * count // target
                q = s0 + h % (s1 - s0)
            pick[t] = mem[q]
        sel_all.append(pick)
        rows = points[pick]
        if c >= n_obj:
            a, b = edges[c - n_obj]
            mk = (masks[pick] == a + 1).astype(np.float32) + 2 * (masks[pick] == b + 1).astype(np.float32)
            rows = np.concatenate([rows, mk[:, None]], 1)
        rows = rows.copy()
        rows[:, :3] = zero_mean(rows[:, :3])
        (obj_out if c < n_obj else rel_out).append(rows)
    return (np.stack(obj_out), np.stack(rel_out), np.array([np.concatenate(b) for b in boxes]), np.concatenate(sel_all),
            np.array([len(m) for m in members]), np.array(edges).T)


# ---------------------------------------------------------------------------------------------- voxel ladder (:37-49)
def voxel_trace(xyz, size, min_bound):
    """open3d's PointCloud.voxel_down_sample_and_trace(size, min_bound, max_bound)[1], restated from the library's
    documented algorithm (open3d is not installed: "parity unpinned" for this function — pinned only by the hand-derived
    known answers in tests/test_prep_cpu.py): double precision, voxel origin = min_bound - size / 2, one row per occupied
    voxel, eight columns = octants (bit c set when the point lies in the upper half of the voxel along axis c), each
    holding the LAST point index that fell into it, -1 where none did."""
    origin = np.asarray(min_bound, dtype=np.float64) - size * 0.5
    ref = (np.asarray(xyz, dtype=np.float64) - origin) / size
    vox = np.floor(ref)
    octant = ((ref - vox) >= 0.5).astype(np.int64) @ np.array([1, 2, 4])
    rows = {}
    for i, (v, o) in enumerate(zip(map(tuple, vox.astype(np.int64)), octant)):
        rows.setdefault(v, [-1] * 8)[o] = i
    return np.array(list(rows.values()), dtype=np.int64).reshape(-1, 8)


def downsample_candidates(pointset, target_N):
    """data_preparation_utils.py:41-47: (best_choice, rung) — the candidate set the final np.random.choice draws from."""
    xyz = np.asarray(pointset)[:, :3].astype(np.float32)
    mn = xyz.min(0)
    best, rung = np.arange(len(xyz)), -1
    for r, size in enumerate(range(15, 100, 5)):
        choice = np.unique(voxel_trace(xyz, size, mn))[1:]
        if len(choice) > target_N:
            best, rung = choice, r
        else:
            break
    return best, rung
