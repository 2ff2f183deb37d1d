"""Host-side pieces added in round 5: rank placement (runtime/affinity.py) and the identity check of pre-grouped rows in a
prefetched geometry (pointnet2_modules.RowsSource)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]

from pointnet2_ops import pointnet2_modules as pm  # noqa: E402
from runtime import affinity  # noqa: E402


def test_parse_cpulist():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.parse_cpulist("") == []
    assert affinity.parse_cpulist("5") == [5]


def test_even_slice_covers_disjointly():
    cpus = list(range(64))
    parts = [affinity.even_slice(cpus, r, 8) for r in range(8)]
    assert all(len(p) == 8 for p in parts)
    assert sorted(sum(parts, [])) == cpus
    # more ranks than cores: every rank still gets a core
    assert all(len(affinity.even_slice([0, 1], r, 8)) == 1 for r in range(8))
    # a remainder goes to the last rank
    assert affinity.even_slice(list(range(10)), 2, 3) == [6, 7, 8, 9]


def test_plan_affinity_prefers_the_gpu_node(tmp_path):
    allowed = list(range(16))
    assert affinity.plan_affinity(allowed, [8, 9, 10, 11, 99], 0, 1) == [8, 9, 10, 11]
    # two ranks on one node split it
    assert affinity.plan_affinity(allowed, [8, 9, 10, 11], 1, 2, ranks_on_node=2, index_on_node=1) == [10, 11]
    # no NUMA information: one rank keeps everything, several ranks take even slices
    assert affinity.plan_affinity(allowed, [], 0, 1) == allowed
    assert affinity.plan_affinity(allowed, [], 3, 4) == [12, 13, 14, 15]


def test_sysfs_readers(tmp_path):
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:05:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("32-63\n")
    assert affinity.numa_node_of_pci("0000:05:00.0", str(tmp_path)) == 1
    assert affinity.numa_node_of_pci("0000:06:00.0", str(tmp_path)) == -1
    assert affinity.cpus_of_node(1, str(tmp_path)) == list(range(32, 64))
    assert affinity.cpus_of_node(0, str(tmp_path)) == []


def test_pin_is_a_noop_without_gpu_and_switchable(monkeypatch):
    before = sorted(os.sched_getaffinity(0))
    info = affinity.pin_to_gpu_numa(0, 1)
    assert sorted(os.sched_getaffinity(0)) == before and not info["pinned"]
    monkeypatch.setenv("PN2_PIN_NUMA", "0")
    assert affinity.pin_to_gpu_numa(0, 4)["why"] == "PN2_PIN_NUMA=0"
    monkeypatch.delenv("PN2_PIN_NUMA")
    # several ranks, no NUMA information: an even slice; restore afterwards
    if len(before) >= 2:
        info = affinity.pin_to_gpu_numa(1, 2)
        try:
            assert info["pinned"] and len(os.sched_getaffinity(0)) == info["cpus"] < len(before)
        finally:
            os.sched_setaffinity(0, before)


def test_rows_source_tracks_identity_and_version():
    f = torch.rand(2, 10, 3)
    tok = pm.rows_source(f)
    assert tok.matches(f) and not tok.matches(f.clone()) and not tok.matches(None)
    f.mul_(2.0)                                    # in-place edit after the prefetch: the rows are stale
    assert not tok.matches(f)
    geo = {"rows": [torch.zeros(1)], "rows_src": pm.rows_source(f)}
    assert pm.rows_still_valid(geo, f)
    assert not pm.rows_still_valid(geo, f.clone())
    assert not pm.rows_still_valid({"rows": None, "rows_src": None}, f)


def test_confirm_rows_marks_or_drops():
    pc = torch.rand(2, 10, 6)
    lvl = {"idx": [torch.zeros(2, 4, 3, dtype=torch.int32)], "rows": [torch.zeros(24, 6)], "rows_src": pm.rows_source(pc)}
    plain = {"idx": [None], "rows": [None], "rows_src": None}
    ok = pm.confirm_rows([lvl, plain, None], pc)
    assert ok[0]["rows_src"].confirmed and ok[0]["rows"] is lvl["rows"] and ok[1] is plain and ok[2] is None
    feats = pc[..., 3:].contiguous()               # the model's own slice: another tensor, accepted because confirmed
    assert pm.rows_still_valid(ok[0], feats)
    other = pm.confirm_rows([lvl], pc.clone())
    assert other[0]["rows"] is None and other[0]["rows_src"] is None
    assert not lvl["rows_src"].confirmed           # the prefetched dict itself is left alone
