"""Can a whole scene-graph training step (fwd + loss + bwd + AdamW) be captured in one hipGraph?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"))
import torch
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper

dev = torch.device("cuda:0")
cfg = config_loader("no_gt.json")

def make():
    torch.manual_seed(0)
    m = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0          # identical eager / graphed trajectories for the comparison
    return m

scans = [to_device(synthetic_scan(9, 4000, 8000, seed=100 + i), dev) for i in range(4)]
TENS = [k for k, v in scans[0].items() if torch.is_tensor(v)]

# eager trajectory
model = make()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
eager_losses = []
for i in range(8):
    opt.zero_grad(set_to_none=True)
    obj, rel = model(scans[i % 4])
    loss = model.loss(obj, rel, scans[i % 4])
    loss.backward()
    opt.step()
    eager_losses.append(loss.item())

# graphed trajectory
model = make()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True)
static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in scans[0].items()}
def load(b):
    for k in TENS:
        static[k].copy_(b[k])
graph_losses = []
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):                      # warm-up steps count as training steps 0..2
        load(scans[i % 4])
        opt.zero_grad(set_to_none=True)
        obj, rel = model(static)
        loss = model.loss(obj, rel, static)
        loss.backward()
        opt.step()
        graph_losses.append(loss.item())
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
load(scans[3])
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    obj, rel = model(static)
    static_loss = model.loss(obj, rel, static)
    static_loss.backward()
    opt.step()
# NOTE: capture does not execute; replay for step 3 onwards
for i in range(3, 8):
    load(scans[i % 4])
    g.replay()
    graph_losses.append(static_loss.item())
print("eager ", ["%.5f" % x for x in eager_losses])
print("graph ", ["%.5f" % x for x in graph_losses])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    load(scans[i % 4])
    g.replay()
torch.cuda.synchronize()
print("graphed ms/step", (time.perf_counter() - t0) * 50)
