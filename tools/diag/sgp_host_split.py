"""Host time of the phases of the 1-scan scene-graph training step (no GPU syncs inside; the step is host-bound)."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "4d-or_amd"))
import torch
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
from runtime.affinity import pin_to_gpu_numa
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = config_loader("no_gt.json")
torch.manual_seed(0)
model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).train()
for n, p in model.named_parameters():
    if ".backbone.fc_layer." in n:
        p.requires_grad_(False)
trainable = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(trainable, lr=1e-4, fused=True)
pin_to_gpu_numa(0, 1)
model.per_scan_statistics = True
scan = to_device(synthetic_scan(9, 4000, 8000, seed=100), dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
geo = None
def step(measure):
    global geo
    t = time.perf_counter()
    if geo is None:
        side.wait_stream(main)
        with torch.cuda.stream(side): geo = model.precompute_geometry(scan)
    main.wait_stream(side)
    batch = dict(scan, geometry=geo)
    side.wait_stream(main)
    with torch.cuda.stream(side): geo = model.precompute_geometry(scan)
    if measure: t = tick("geometry", t)
    opt.zero_grad(set_to_none=True)
    obj, rel = model(batch)
    if measure: t = tick("forward", t)
    loss = model.loss(obj, rel, batch)
    if measure: t = tick("loss", t)
    loss.backward()
    if measure: t = tick("backward", t)
    opt.step()
    if measure: t = tick("optimizer", t)
for _ in range(10): step(False)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(json.dumps({"host_ms": host / N * 1e3, "wall_ms": tot / N * 1e3, **{k: round(v / N * 1e3, 3) for k, v in acc.items()}}))
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N): step(False)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats(key).print_stats(60)
        print(st.getvalue())
    sys.exit(0)
# forward detail by module: hooks on top-level children
import collections
times = collections.defaultdict(float)
def pre(name):
    def f(m, i): m._t0 = time.perf_counter()
    return f
def post(name):
    def f(m, i, o): times[name] += time.perf_counter() - m._t0
    return f
for name, m in model.named_children():
    m.register_forward_pre_hook(pre(name)); m.register_forward_hook(post(name))
for _ in range(N): step(False)
torch.cuda.synchronize()
print(json.dumps({k: round(v / N * 1e3, 3) for k, v in times.items()}))
