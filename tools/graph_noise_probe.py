import os, sys, copy
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import torch
from test_gpu_graphed_step import _model, _rel
from runtime import GraphedTrainStep
from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device

model = _model()
params = [p for p in model.parameters() if p.requires_grad]
names = [n for n, p in model.named_parameters() if p.requires_grad]
scan = to_device(synthetic_scan(3, 1024, 2048, seed=1), "cuda")

def eager_grad():
    model.zero_grad(set_to_none=True)
    l, _ = model.pure_training_step(scan)
    l.backward()
    return torch.cat([p.grad.flatten() for p in params]).clone()

e = [eager_grad() for _ in range(4)]
print("eager vs eager:", [round(_rel(e[i], e[0]), 5) for i in range(1, 4)])
opt = torch.optim.SGD(params, lr=0.0)
st = GraphedTrainStep(model.pure_training_step, params, opt)
g = []
for i in range(5):
    st(scan)
    g.append(st.grads.flat.clone())
print("graphs:", st.num_graphs)
print("call_i vs eager0:", [round(_rel(x, e[0]), 5) for x in g])
print("replay vs replay:", [round(_rel(g[i], g[2]), 5) for i in range(3, 5)])
# per-parameter
off = 0
rows = []
for n, p in zip(names, params):
    a, b = g[2][off:off + p.numel()], e[0][off:off + p.numel()]
    rows.append((_rel(a, b), n, float(b.norm())))
    off += p.numel()
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.4f  %-70s |g|=%.3e" % r)
