import os, sys, time
REPO = "/root/repo"
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.build_model(dev); opt = torch.optim.AdamW(model.parameters(), lr=3e-5)
pc = bench.synthetic_scenes(32, 50000, 1, dev)
pf = bench.GeometryPrefetcher(model, dev)
bench.run_steps(model, model, opt, pc, 5, pf)
torch.cuda.synchronize()
# instrumented loop
tl = ta = tt = 0.0
geo = pf.pending; pf.pending = None
t0 = time.perf_counter()
for i in range(20):
    a = time.perf_counter(); cur = pf.acquire(geo); b = time.perf_counter(); nxt = pf.launch(pc); c = time.perf_counter()
    bench.train_step(model, opt, pc, cur); d = time.perf_counter()
    ta += b - a; tl += c - b; tt += d - c; geo = nxt
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print("per step host ms: acquire %.3f launch-geometry %.3f train_step %.3f total %.3f; drain %.1f ms" % (ta/20*1e3, tl/20*1e3, tt/20*1e3, (t1-t0)/20*1e3, (t2-t1)*1e3))
