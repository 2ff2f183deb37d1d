"""Experiment: the eval-mode forward of the scene-graph model (1 scan) as a replayed hipGraph vs eager."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SGPNModelWrapper(config_loader("no_gt.json"), 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).eval()
batch = to_device(synthetic_scan(9, 4000, 8000, seed=100), dev)

def fwd(with_geo):
    with torch.no_grad():
        b = dict(batch, geometry=model.precompute_geometry(batch)) if with_geo else batch
        return model(b)

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref = fwd(True)
print("eager, geometry inline: %.3f ms" % timeit(lambda: fwd(True)))
model.encoder_streams = False
print("eager, one stream:      %.3f ms" % timeit(lambda: fwd(True)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fwd(True)
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = fwd(True)
    g.replay(); torch.cuda.synchronize()
    print("replay matches eager:", all(torch.allclose(a, b, atol=1e-5) for a, b in zip(out, ref)))
    print("graph replay:           %.3f ms" % timeit(g.replay))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
