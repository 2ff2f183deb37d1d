"""cProfile of the host thread over a few eager steps (where does the enqueue time go?):
python tools/host_profile.py [backbone|sgp]"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "backbone"
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
    S = int(which[3:] or 1)                      # "sgp8": eight scans per step, per-scan statistics
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

if os.environ.get("PN2_MLP_DTYPE"):                 # PN2_MLP_DTYPE=bf16 python tools/host_profile.py sgp8
    from pointnet2_ops import fused_mlp
    fused_mlp.set_mlp_dtype(os.environ["PN2_MLP_DTYPE"])
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(45)
if os.environ.get("PN2_PROFILE_CALLERS"):           # e.g. PN2_PROFILE_CALLERS="contiguous|torch.cat|torch.zeros"
    st.print_callers(os.environ["PN2_PROFILE_CALLERS"])
