"""Multi-process path on CPU: 2 ranks, gloo, each rank its own scenes (weak scaling),
gradients all-reduced by DDP exactly like bench.py does over RCCL.  Checks that the
averaged gradients equal a single-process run over the concatenated batch *per rank
statistics* (BatchNorm stays per rank: no SyncBN), and that parameters stay in sync."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ext
    from pointnet2_ops import pointnet2_modules as pm, pointnet2_utils as pu
    pu._ext = oracle_ext.OracleRowsExt
    torch.manual_seed(0)                       # identical initial weights on every rank
    net = torch.nn.ModuleList([
        pm.PointnetSAModuleMSG(npoint=32, radii=[0.3, 0.6], nsamples=[4, 8], mlps=[[3, 8, 8], [3, 8, 16]]),
        pm.PointnetFPModule(mlp=[24 + 3, 16]),
    ])

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sa, self.fp = net[0], net[1]

        def forward(self, pc):
            xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
            nx, nf = self.sa(xyz, feats)
            return self.fp(xyz, nx, feats, nf)

    model = Net()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(100 + rank)          # different scenes per rank
    pc = torch.rand(2, 200, 6, generator=g) * 2 - 1
    local = Net()
    local.load_state_dict(model.state_dict())
    local(pc).square().mean().backward()                    # this rank's own gradient, no communication
    local_grads = [p.grad.clone() for p in local.parameters()]
    for _ in range(2):
        opt.zero_grad()
        ddp(pc).square().mean().backward()
        if _ == 0:
            first = [p.grad.clone() for p in model.parameters()]
        opt.step()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    lg = torch.cat([x.flatten() for x in local_grads])
    lgs = [torch.zeros_like(lg) for _ in range(world)]
    dist.all_gather(lgs, lg)
    if rank == 0:
        torch.save({"params": gathered, "ddp_grad": torch.cat([x.flatten() for x in first]), "local_grads": lgs}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["params"][0], res["params"][1])                    # ranks stay in sync
    mean_local = (res["local_grads"][0] + res["local_grads"][1]) / 2          # DDP averages gradients
    torch.testing.assert_close(res["ddp_grad"], mean_local, atol=1e-6, rtol=1e-5)


def _flat_worker(rank, world, port, out):
    """bench.py's default gradient exchange (FlatGradSync: one all-reduce of a flat buffer, no DDP) on 2 gloo ranks."""
    sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import oracle_ext
    from pointnet2_ops import pointnet2_modules as pm, pointnet2_utils as pu
    pu._ext = oracle_ext.OracleRowsExt
    torch.manual_seed(rank)                    # DIFFERENT initial weights: the constructor must broadcast rank 0's
    sa = pm.PointnetSAModuleMSG(npoint=32, radii=[0.3, 0.6], nsamples=[4, 8], mlps=[[3, 8, 8], [3, 8, 16]])

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sa = sa

        def forward(self, pc, geometry=None):
            xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
            return {"fp2_features": self.sa(xyz, feats)[1]}

    model = Net()
    sync = bench.FlatGradSync(model.parameters(), world)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(100 + rank)
    pc = torch.rand(2, 200, 6, generator=g) * 2 - 1
    local = Net()
    local.sa = __import__("copy").deepcopy(model.sa)
    local(pc)["fp2_features"].square().mean().backward()
    lg = torch.cat([p.grad.flatten() for p in local.parameters()])
    for i in range(2):
        bench.train_step(model, opt, pc, sync=sync)
        if i == 0:
            first = sync.flat.flat.clone()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    lgs = [torch.zeros_like(lg) for _ in range(world)]
    dist.all_gather(lgs, lg)
    if rank == 0:
        torch.save({"params": gathered, "sync_grad": first, "local_grads": lgs}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_flat_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["params"][0], res["params"][1])                    # broadcast at start + identical updates
    mean_local = (res["local_grads"][0] + res["local_grads"][1]) / 2
    torch.testing.assert_close(res["sync_grad"], mean_local, atol=1e-6, rtol=1e-5)


def _runner_worker(rank, world, port, out):
    """scene_graph_prediction.main.train on 2 gloo ranks: the full SGPN model (two MSG encoders + TripletGCN + heads) on the
    CPU oracle backend, scans sharded by rank, ONE flat gradient all-reduce per step."""
    sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import copy
    import oracle_ext
    from pointnet2_ops import pointnet2_utils as pu
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    pu._ext = gcn._ext = oracle_ext.OracleRowsExt
    from scene_graph_prediction import main as runner
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    device = torch.device("cpu")
    r, w = runner.setup_distributed(device)
    assert (r, w) == (rank, world)
    cfg = runner.config_loader("no_gt.json")
    torch.manual_seed(7 + rank)                  # DIFFERENT initial weights per rank: train() must broadcast rank 0's
    model = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), runner.RELATION_NAMES)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                            # deterministic steps (the local reference below re-runs them)
    scans = [synthetic_scan(3, 300, 400, seed=i, scan_id=f"s{i}") for i in range(5)]     # ODD count on 2 ranks

    # reference: what this rank's first step computes locally, from rank 0's weights
    torch.manual_seed(7)
    ref = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), runner.RELATION_NAMES)
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ref.train()
    loss = ref.training_step(to_device(scans[rank], device), 0)
    loss.backward()
    names = [n for n, p in ref.named_parameters() if p.grad is not None and ".backbone.fc_layer." not in n]
    local = torch.cat([dict(ref.named_parameters())[n].grad.flatten() for n in names])

    seen = {}
    orig_step = torch.optim.AdamW.step

    def spy(self, *a, **k):                      # gradients as the optimizer sees them at the first step
        if "g" not in seen:
            byname = {n: p for n, p in model.named_parameters()}
            seen["g"] = torch.cat([byname[n].grad.flatten() for n in names]).clone()
        return orig_step(self, *a, **k)

    torch.optim.AdamW.step = spy
    try:
        hist = runner.train(model, cfg, scans, device, epochs=1, rank=rank, world=world, log=lambda *_: None)
    finally:
        torch.optim.AdamW.step = orig_step
    assert len(hist) == 2                        # 5 scans over 2 ranks: the fifth is dropped, both ranks run 2 steps (a
                                                 # rank with a third step would hang in its all-reduce: ADVICE r03)
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen and all(".backbone.fc_layer." in n for n in frozen)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    if rank == 0:
        torch.save({"params": gathered, "seen": seen["g"], "locals": locals_}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_runner_train_mode_on_two_gloo_ranks(tmp_path):
    """4d-or_amd/scene_graph_prediction/main.py --mode train under torchrun (here: 2 gloo ranks on the CPU oracle
    backend): rank 0's weights everywhere, scans sharded by rank, the gradient the optimizer sees = the mean of the ranks'
    local gradients, the inherited dead classification heads frozen, parameters in sync after the epoch."""
    out = str(tmp_path / "runner.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_runner_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["params"][0], res["params"][1])
    mean_local = (res["locals"][0] + res["locals"][1]) / 2
    torch.testing.assert_close(res["seen"], mean_local, atol=1e-6, rtol=1e-4)


def _graphed_worker(rank, world, port, out):
    """bench.py --workload sgp --graphs under torchrun: runtime.GraphedTrainStep over the full SGPN model's
    `pure_training_step` with a process group — flat gradient buffer, ONE all-reduce between the backward and the optimizer
    (on a CPU/gloo job the eager schedule of the same class: capture=False), two block-diagonal scans per step and rank."""
    sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ext
    from pointnet2_ops import pointnet2_utils as pu
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as gcn
    pu._ext = gcn._ext = oracle_ext.OracleRowsExt
    from runtime import GraphedTrainStep
    from scene_graph_prediction import main as runner
    from scene_graph_prediction.scene_graph_helpers.dataset.synthetic import collate_scans, synthetic_scan, to_device
    from scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model import SGPNModelWrapper
    device = torch.device("cpu")
    cfg = runner.config_loader("no_gt.json")

    def build():
        torch.manual_seed(0)                     # bench_sgp seeds every rank alike: identical initial weights
        m = SGPNModelWrapper(cfg, 12, 15, torch.ones(12), torch.ones(15), runner.RELATION_NAMES).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        for n, p in m.named_parameters():
            if ".backbone.fc_layer." in n:
                p.requires_grad_(False)
        return m

    model, ref = build(), build()
    batch = to_device(collate_scans([synthetic_scan(3, 300, 400, seed=10 * rank + i, scan_id=f"s{rank}_{i}") for i in range(2)]),
                      device)
    # this rank's own gradient, no communication
    loss, _ = ref.pure_training_step(batch)
    loss.backward()
    names = [n for n, p in ref.named_parameters() if p.requires_grad]
    local = torch.cat([(dict(ref.named_parameters())[n].grad if dict(ref.named_parameters())[n].grad is not None
                        else torch.zeros_like(dict(ref.named_parameters())[n])).flatten() for n in names])

    trainable = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(trainable, lr=float(cfg["LR"]), weight_decay=float(cfg["W_DECAY"]))
    stepper = GraphedTrainStep(model.pure_training_step, trainable, opt, capture=False)
    assert stepper.distributed
    seen = {}
    orig_step = torch.optim.AdamW.step

    def spy(self, *a, **k):
        if "g" not in seen:
            seen["g"] = stepper.grads.flat.clone()
        return orig_step(self, *a, **k)

    torch.optim.AdamW.step = spy
    try:
        for _ in range(2):
            stepper(batch)
    finally:
        torch.optim.AdamW.step = orig_step
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    if rank == 0:
        torch.save({"params": gathered, "seen": seen["g"], "locals": locals_}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_step_flat_gradients_on_two_gloo_ranks(tmp_path):
    """The `--graphs` schedule of bench_sgp / main.py on two ranks: the optimizer sees the MEAN of the ranks' local
    gradients (one flat all-reduce), and parameters stay identical across ranks."""
    out = str(tmp_path / "graphed.pt")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_graphed_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["params"][0], res["params"][1])
    mean_local = (res["locals"][0] + res["locals"][1]) / 2
    torch.testing.assert_close(res["seen"], mean_local, atol=1e-6, rtol=1e-4)
