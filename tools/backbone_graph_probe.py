"""Whole-step hipGraph replay of the backbone step (no geometry pipeline): how much are the inter-kernel gaps worth?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4d-or_amd")]
import torch
import bench
from runtime import GraphedTrainStep
dev = torch.device("cuda:0")
model = bench.build_model(dev)
pc = bench.synthetic_scenes(32, 50000, seed=1, device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=3e-5, weight_decay=1e-3)
for _ in range(3):
    bench.train_step(model, opt, pc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    bench.train_step(model, opt, pc)
torch.cuda.synchronize()
print("eager   ms/step", (time.perf_counter() - t0) * 100)

opt2 = torch.optim.AdamW(model.parameters(), lr=3e-5, weight_decay=1e-3, capturable=True)
def step_fn(batch):
    feats = model(batch["pc"])["fp2_features"]
    return feats.square().mean(), None
stepper = GraphedTrainStep(step_fn, model.parameters(), opt2)
batch = {"pc": pc}
for _ in range(4):
    stepper(batch)
torch.cuda.synchronize()
print("graphs:", stepper.num_graphs)
t0 = time.perf_counter()
for _ in range(10):
    stepper(batch)
torch.cuda.synchronize()
print("graphed ms/step", (time.perf_counter() - t0) * 100)
