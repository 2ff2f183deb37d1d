"""Where does a full scene-graph step spend its wall time? (enqueue vs GPU)"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"))
import torch
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

dev = torch.device("cuda:0")
cfg = config_loader("no_gt.json")
torch.manual_seed(0)
model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
scan = to_device(synthetic_scan(9, 4000, 8000, seed=100), dev)

def step():
    opt.zero_grad(set_to_none=True)
    obj, rel = model(scan)
    model.loss(obj, rel, scan).backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue ms/step", (t1 - t0) * 100, "total ms/step", (t2 - t0) * 100)
import warnings
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
    import traceback
torch.cuda.set_sync_debug_mode("default")
print("sync warnings in one step:", len(w))
for x in w[:12]:
    print("  ", x.filename, x.lineno, str(x.message)[:100])
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=50))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
