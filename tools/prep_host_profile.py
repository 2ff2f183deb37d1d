"""cProfile of the host thread over the GPU preparation of 8 scans (what `bench.py --with-prep` adds to a step):
python tools/prep_host_profile.py"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch  # noqa: E402

from scene_graph_prediction.scene_graph_helpers.dataset import gpu_preparation as gp  # noqa: E402
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, to_device  # noqa: E402

dev = torch.device("cuda", 0)
S = 8
fused = [gp.synthetic_fused_scan(9, 300000, seed=100 + i, device=dev) for i in range(S)]
names = ["Patient", "operating_table", "human_0", "human_1", "instrument", "secondary_table", "instrument_table",
         "anesthesia_equipment", "human_2"]
g = torch.Generator().manual_seed(7)
labels = [(torch.randint(0, 12, (9,), generator=g).to(dev), torch.randint(0, 15, (72,), generator=g).to(dev)) for _ in range(S)]


def prepare(step_idx):
    scans = [gp.prepare_scan(p, m, 9, 4000, 8000, seed=step_idx * S + i, object_names=names, gt_class=labels[i][0],
                             gt_rels=labels[i][1], scan_id=f"prep_{i:06d}") for i, (p, m) in enumerate(fused)]
    for sc in scans:
        sc.pop("prep")
    return to_device(collate_scans(scans), dev)


for i in range(3):
    prepare(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    prepare(10 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(30)
