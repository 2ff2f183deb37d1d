"""Where do the torch-native copy / fill kernels of a backbone step come from?  (torch profiler with python stacks)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.build_model(dev); opt = torch.optim.AdamW(model.parameters(), lr=3e-5)
pc = bench.synthetic_scenes(32, 50000, 1, dev)
pf = bench.GeometryPrefetcher(model, dev)
bench.run_steps(model, model, opt, pc, 3, pf)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.run_steps(model, model, opt, pc, 2, pf)
    torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul", "aten::add", "aten::contiguous", "aten::clone"):
        st = [s for s in ev.stack if "/root/repo" in s or "4d-or_amd" in s][:2]
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:70] for s in st))
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[0]:4d} {v[1]/2:9.1f} us/step  {k}")
