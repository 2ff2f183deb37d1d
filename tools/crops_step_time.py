"""Scene-graph training step on PREPARED crops (object / pair crops cut out of 300k-point fused scans by the GPU preparation:
the ball densities of real data — half-full balls, a few points owning hundreds of rows), preparation OUTSIDE the timed loop.
    [PN2_BF16_LIFT=0] [PN2_LIFT_SPARSE=0] python tools/crops_step_time.py [bf16|f32] [scans]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import fused_mlp
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset import gpu_preparation as gp
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
fused_mlp.set_mlp_dtype(dtype)
cfg = config_loader("no_gt.json")
model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).train()
for n, p in model.named_parameters():
    if ".backbone.fc_layer." in n:
        p.requires_grad_(False)
model.per_scan_statistics = os.environ.get("PER_SCAN") == "1"
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
names = ["Patient", "operating_table", "human_0", "human_1", "instrument", "secondary_table", "instrument_table",
         "anesthesia_equipment", "human_2"]
g = torch.Generator().manual_seed(7)
scans = []
for i in range(S):
    p, m = gp.synthetic_fused_scan(9, 300000, seed=100 + i, device=dev)
    sc = gp.prepare_scan(p, m, 9, 4000, 8000, seed=i, object_names=names, gt_class=torch.randint(0, 12, (9,), generator=g).to(dev),
                         gt_rels=torch.randint(0, 15, (72,), generator=g).to(dev), scan_id=f"prep_{i:06d}")
    sc.pop("prep")
    scans.append(sc)
batch = scans[0] if S == 1 else to_device(collate_scans(scans), dev)


def step():
    opt.zero_grad(set_to_none=True)
    obj, rel = model(batch)
    model.loss(obj, rel, batch).backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(15):
    step()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 15
print(json.dumps({"dtype": dtype, "scans": S, "per_scan": model.per_scan_statistics, "BF16_LIFT": fused_mlp.BF16_LIFT,
                  "LIFT_SPARSE": fused_mlp.LIFT_SPARSE, "ms_per_step": round(ms, 2), "scans_per_s": round(S / ms * 1e3, 1)}))
