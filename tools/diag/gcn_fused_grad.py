"""Diagnostic: fused vs unfused TripletGCN gradients against an fp64 torch restatement on the triplet_gcn fixture."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
import copy
import numpy as np, torch
import fixture_checks as fc
from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn

z = fc.load("triplet_gcn.npz")
torch.manual_seed(71)
model = gcn.TripletGCNModel(num_layers=2, dim_node=256, dim_edge=256, dim_hidden=512)
ei = torch.from_numpy(z["l2/ei"])

def plain(m, x, e, ei):
    for i, l in enumerate(m.gconvs):
        x_i, x_j = x.index_select(0, ei[1]), x.index_select(0, ei[0])
        h = l.nn1(torch.cat([x_i, e, x_j], 1))
        dh, de = l.dim_hidden, l.dim_edge
        msg = h[:, :dh] + h[:, dh + de:]
        agg = torch.zeros(x.size(0), dh, dtype=x.dtype, device=x.device).index_add_(0, ei[1], msg)
        x, e = l.nn2(agg), h[:, dh:dh + de]
        if i < len(m.gconvs) - 1:
            x, e = torch.relu(x), torch.relu(e)
    return x, e

def run(fn, dev, dtype):
    m = copy.deepcopy(model).to(dev).to(dtype).train()
    x = torch.from_numpy(z["l2/x"]).to(dev).to(dtype).requires_grad_(True)
    e = torch.from_numpy(z["l2/e"]).to(dev).to(dtype).requires_grad_(True)
    ox, oe = fn(m, x, e, ei.to(dev))
    wx = torch.linspace(0.5, 1.5, ox.numel(), device=dev, dtype=torch.float32).view_as(ox).to(dtype)
    we = torch.linspace(-1.0, 1.0, oe.numel(), device=dev, dtype=torch.float32).view_as(oe).to(dtype)
    ((ox * wx).sum() + (oe * we).sum()).backward()
    return [t.detach().double().cpu() for t in (ox, oe, x.grad, e.grad)]

ref64 = run(plain, "cpu", torch.float64)
ref32 = run(plain, "cpu", torch.float32)
def product(m, x, e, ei): return m(x, e, ei)
res = {"cpu plain fp32": ref32, "fixture": [torch.from_numpy(z[k]).double() for k in ("l2/out_x", "l2/out_e", "l2/grad_x", "l2/grad_e")]}
for name, fused, lift in (("fused", True, 1 << 60), ("lifted", False, 0), ("concat", False, 1 << 60)):
    gcn.FUSED_LAYER, gcn.LIFT_MIN_EDGES = fused, lift
    res["gpu " + name] = run(product, "cuda", torch.float32)
for name, r in res.items():
    print(name, " ".join(f"{k}:{float((a - b).abs().max()):.2e}/{float(b.abs().max()):.2f}" for k, a, b in zip(("ox", "oe", "gx", "ge"), r, ref64)))
d = (res["gpu fused"][3] - ref64[3]).abs()
cols = d.max(0).values
top = torch.topk(cols, 8)
print("worst grad_e columns (fused):", top.indices.tolist(), [f"{v:.1e}" for v in top.values.tolist()])
d2 = (res["gpu concat"][3] - ref64[3]).abs().max(0).values
print("same columns, concat:", [f"{float(d2[i]):.1e}" for i in top.indices])
d3 = (ref32[3] - ref64[3]).abs().max(0).values
print("same columns, cpu fp32:", [f"{float(d3[i]):.1e}" for i in top.indices])
for name in ("gpu fused", "gpu concat", "cpu plain fp32"):
    d = (res[name][3] - ref64[3]).abs()
    rows = d.max(1).values
    t = torch.topk(rows, 4)
    print(name, "grad_e worst rows", t.indices.tolist(), [f"{v:.1e}" for v in t.values.tolist()], "median row err", f"{float(rows.median()):.1e}")
# smallest |pre-activation| in fp64 at every ReLU
m = copy.deepcopy(model).double().train()
x = torch.from_numpy(z["l2/x"]).double(); e = torch.from_numpy(z["l2/e"]).double()
acts = []
hooks = [mod.register_forward_hook(lambda mod, i, o: acts.append(i[0].detach())) for mod in m.modules() if isinstance(mod, torch.nn.ReLU)]
ox, oe = plain(m, x, e, ei)
acts += [ox.detach() * 0 + 1]  # placeholder
for i, a in enumerate(acts[:-1]):
    v, idx = a.abs().flatten().min(0)
    print(f"relu {i}: shape {tuple(a.shape)} min |pre| {float(v):.2e} at row {int(idx) // a.size(1)} col {int(idx) % a.size(1)}")
