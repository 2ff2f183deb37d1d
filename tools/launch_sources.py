"""Where do the torch-side launches (fills, copies, cats, element-wise glue) of a step come from?  One eager step under a
TorchDispatchMode: every aten op on a device tensor is attributed to the innermost python frame inside this repo (the mode
follows the autograd engine's threads).
python tools/launch_sources.py [backbone|sgp|sgp8]"""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sgp"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if which.startswith("sgp"):
    from scene_graph_prediction.main import RELATION_NAMES, config_loader
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    cfg = config_loader("no_gt.json")
    model = SGPNModelWrapper(cfg, 12, len(RELATION_NAMES), torch.ones(12), torch.ones(len(RELATION_NAMES)),
                             RELATION_NAMES).to(dev).train()
    opt = torch.optim.AdamW([p for p in model.parameters()], lr=1e-4)
    S = int(which[3:] or 1)
    scan = to_device(synthetic_scan(9, 4000, 8000, seed=1), dev) if S == 1 else to_device(
        collate_scans([synthetic_scan(9, 4000, 8000, seed=i) for i in range(S)]), dev)

    def step():
        opt.zero_grad(set_to_none=True)
        obj, rel = model(scan)
        model.loss(obj, rel, scan).backward()
        opt.step()
else:
    model = bench.build_model(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-5)
    pc = bench.synthetic_scenes(32, 50000, 1, dev)

    def step():
        bench.train_step(model, opt, pc)

for _ in range(3):
    step()
torch.cuda.synchronize()

SKIP = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.as_strided", "aten.detach", "aten.t.", "aten.transpose", "aten.slice",
        "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.alias", "aten.permute", "aten.reshape", "aten.unbind",
        "aten.split", "aten._local_scalar_dense", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.resize_", "aten.set_",
        "aten.record_stream", "aten.new_empty", "aten.empty_like", "aten.unfold", "aten.narrow", "aten.chunk", "aten._reshape_alias")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(SKIP):
            return out
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        flat += [o for o in (out if isinstance(out, (tuple, list)) else [out]) if isinstance(o, torch.Tensor)]
        if not any(t.is_cuda for t in flat):
            return out
        if all(t.numel() == 0 for t in flat):
            return out
        frame = "(no repo frame: autograd engine / optimizer)"
        for fs in reversed(traceback.extract_stack()):
            fn = fs.filename
            if fn.startswith(REPO) and "launch_sources" not in fn:
                frame = f"{fn.replace(REPO + '/', '')}:{fs.lineno} {fs.name}"
                break
        self.sites[(name, frame)] += 1
        self.ops[name] += 1
        return out


mode = Sites()
with mode:
    step()
torch.cuda.synchronize()
print(f"[{which}] aten ops on device tensors in one step (views / allocations excluded): {sum(mode.ops.values())}")
for (name, fr), n in mode.sites.most_common(90):
    print(f"{n:4d}  {name:34s} {fr[:160]}")
print(dict(mode.ops.most_common()))
