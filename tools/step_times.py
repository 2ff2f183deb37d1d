"""Per-step GPU time of the default bench loop (event per step on the main stream, no host sync inside the loop): does the
step time drift over the first steps (clock ramp, pipeline alignment)?   python tools/step_times.py [steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(dev)
opt = torch.optim.AdamW(model.parameters(), lr=3e-5)
pc = bench.synthetic_scenes(32, 50000, 1, dev)
pre = bench.GeometryPrefetcher(model, dev)
evs = []
def on_step(i):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
bench.run_steps(model, model, opt, pc, 5, pre)
torch.cuda.synchronize()
import time; time.sleep(float(os.environ.get("IDLE_S", "0")))
bench.run_steps(model, model, opt, pc, n, pre, on_step)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
t = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print("steps 0-9  :", " ".join("%.1f" % v for v in t[:10]))
print("steps 10-19:", " ".join("%.1f" % v for v in t[10:20]))
print("steps 20-39:", " ".join("%.1f" % v for v in t[20:40]))
print("mean first 20 %.2f, 20-40 %.2f, 40+ %.2f" % (sum(t[:20]) / 20, sum(t[20:40]) / 20, sum(t[40:]) / max(1, len(t) - 40)))
