"""Do two hipGraphs replayed on two streams overlap?  Forward of the object and of the relation encoder (train-mode BatchNorm,
no autograd) captured separately; replayed back to back on one stream vs side by side on two."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "4d-or_amd"))
import torch
from scene_graph_prediction.main import RELATION_NAMES, config_loader
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper, per_scan_statistics
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = config_loader("no_gt.json")
torch.manual_seed(0)
model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)), RELATION_NAMES).to(dev).train()
scan = to_device(synthetic_scan(9, 4000, 8000, seed=100), dev)
geo = model.precompute_geometry(scan)
torch.cuda.synchronize()
def fwd_obj():
    with torch.no_grad(), per_scan_statistics():
        return model.obj_encoder(scan["obj_points"], geometry=geo["obj"])
def fwd_rel():
    with torch.no_grad(), per_scan_statistics():
        return model.rel_encoder(scan["rel_points"], geometry=geo["rel"])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
graphs = []
for f in (fwd_obj, fwd_rel):
    with torch.cuda.stream(s1):
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s1):
            out = f()
    torch.cuda.synchronize()
    graphs.append((g, out))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
go, gr = graphs[0][0], graphs[1][0]
def serial():
    with torch.cuda.stream(s1): go.replay(); gr.replay()
def only_o():
    with torch.cuda.stream(s1): go.replay()
def only_r():
    with torch.cuda.stream(s1): gr.replay()
def par():
    s2.wait_stream(s1)
    with torch.cuda.stream(s2): go.replay()
    with torch.cuda.stream(s1): gr.replay()
    s1.wait_stream(s2)
def eager_serial():
    fwd_obj(); fwd_rel()
def eager_par():
    s2.wait_stream(s1)
    with torch.cuda.stream(s2): fwd_obj()
    with torch.cuda.stream(s1): fwd_rel()
    s1.wait_stream(s2)
print(json.dumps({"graph_obj_ms": timeit(only_o), "graph_rel_ms": timeit(only_r), "graphs_serial_ms": timeit(serial), "graphs_two_streams_ms": timeit(par),
                  "eager_serial_ms": timeit(eager_serial), "eager_two_streams_ms": timeit(eager_par)}))
