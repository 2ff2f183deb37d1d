"""TripletGCN (2 layers, 256/256/512) forward+backward time on the GPU: fused per-scan layer kernels vs the unfused path
(lifted first Linear / literal concat form),
for one scan (9 objects / 72 edges) and for block-diagonal batches of 8 / 32 / 64 scans.
    python tools/gcn_time.py            (GPU box; prints one JSON line per case)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"))
from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn  # noqa: E402


def case(n_scans, lifted, iters=30, fused=False, layer_call=True):
    gcn.FUSED_LAYER = fused
    gcn.LAYER_CALL = layer_call
    gcn.FUSED_MAX_SCANS = 1 << 30            # time the fused kernels at every scan count (the model routes by FUSED_MAX_SCANS)
    gcn.LIFT_MIN_EDGES = 0 if lifted else 1 << 60
    torch.manual_seed(0)
    net = gcn.TripletGCNModel(2, dim_node=256, dim_edge=256, dim_hidden=512).cuda()
    n = 9
    eis, node_ptr, edge_ptr = [], [0], [0]
    for s in range(n_scans):
        ei = torch.tensor([[a, b] for a in range(n) for b in range(n) if a != b]).t() + node_ptr[-1]
        eis.append(ei); node_ptr.append(node_ptr[-1] + n); edge_ptr.append(edge_ptr[-1] + ei.size(1))
    ei = torch.cat(eis, 1).contiguous().cuda()
    scenes = gcn.SceneBatch(torch.tensor(node_ptr), torch.tensor(edge_ptr)).to("cuda") if n_scans > 1 else None
    x = torch.randn(node_ptr[-1], 256, device="cuda", requires_grad=True)
    e = torch.randn(edge_ptr[-1], 256, device="cuda", requires_grad=True)
    csr = gcn.EdgeCSR(ei, x.size(0))

    params = list(net.parameters())

    def step():
        for q in params:            # like the runner's zero_grad(set_to_none=True): the first accumulation keeps the kernel's
            q.grad = None           # tensor (left to accumulate, every parameter costs one more small launch per step)
        x.grad = e.grad = None
        a, b = net(x, e, ei, csr=csr, scenes=scenes)
        (a.sum() + b.sum()).backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):              # best of three timing loops (a loop occasionally catches a ~50 ms host stall)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            step()
        t1.record(); torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / iters)
    return best


if __name__ == "__main__":
    for n_scans in [int(v) for v in sys.argv[1:]] or (1, 8, 32, 64):
        row = {"scans": n_scans, "edges": 72 * n_scans}
        for lifted in (False, True):
            row["lifted_ms" if lifted else "concat_ms"] = round(case(n_scans, lifted), 3)
        row["fused_ms"] = round(case(n_scans, False, fused=True), 3)        # csrc/gcn_fused.hip (round 4), one C call per layer and direction
        row["fused_blocks_ms"] = round(case(n_scans, False, fused=True, layer_call=False), 3)   # the same kernels, one python -> C call per block
        print(json.dumps(row))
