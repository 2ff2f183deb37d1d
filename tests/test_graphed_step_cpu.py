"""runtime.GraphedTrainStep on CPU: the eager schedule (flat gradient buffer, one all-reduce between
backward and optimizer) single-process and as a 2-rank gloo job.  Graph capture itself needs a GPU
(tests/test_gpu_graphed_step.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from runtime import FlatGrads, GraphedTrainStep, batch_signature

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _step_fn(net):
    def f(batch):
        out = net(batch["x"])
        return torch.nn.functional.mse_loss(out, batch["y"]), out
    return f


def _batch(seed, n=32):
    g = torch.Generator().manual_seed(seed)
    return {"x": torch.randn(n, 6, generator=g), "y": torch.randn(n, 3, generator=g), "scan_id": f"s{seed}"}


def test_batch_signature_separates_shapes_and_ignores_metadata():
    a, b = _batch(0), _batch(1)
    assert batch_signature(a) == batch_signature(b)
    assert batch_signature(a) != batch_signature(_batch(0, n=33))


def test_flat_grads_are_views_and_accumulate_in_place():
    net = _net()
    fg = FlatGrads(net.parameters())
    assert fg.flat.numel() == sum(p.numel() for p in net.parameters())
    ptrs = [p.grad.data_ptr() for p in net.parameters()]
    loss, _ = _step_fn(net)(_batch(0))
    loss.backward()
    assert [p.grad.data_ptr() for p in net.parameters()] == ptrs          # autograd accumulated into the views
    ref = _net()
    _step_fn(ref)(_batch(0))[0].backward()
    torch.testing.assert_close(fg.flat, torch.cat([p.grad.flatten() for p in ref.parameters()]))
    fg.zero_()
    assert all(float(p.grad.abs().sum()) == 0 for p in net.parameters())


def test_eager_schedule_equals_plain_training_loop():
    a, b = _net(), _net()
    oa, ob = torch.optim.AdamW(a.parameters(), lr=1e-2), torch.optim.AdamW(b.parameters(), lr=1e-2)
    stepper = GraphedTrainStep(_step_fn(a), a.parameters(), oa, capture=False)
    for i in range(4):
        la, _ = stepper(_batch(i))
        ob.zero_grad()
        lb, _ = _step_fn(b)(_batch(i))
        lb.backward()
        ob.step()
        torch.testing.assert_close(la, lb.detach())
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb)


def test_capture_on_cpu_fails_loudly():
    net = _net()
    with pytest.raises(RuntimeError, match="needs the model on a GPU"):
        GraphedTrainStep(_step_fn(net), net.parameters(), torch.optim.SGD(net.parameters(), lr=0.1))


def _worker(rank, world, port, out):
    sys.path[:0] = [os.path.join(REPO, "4d-or_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from runtime import GraphedTrainStep as GTS
    net = _net()
    local = _net()
    batch = _batch(100 + rank)                                  # each rank its own scans (weak scaling)
    _step_fn(local)(batch)[0].backward()
    lg = torch.cat([p.grad.flatten() for p in local.parameters()])
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    stepper = GTS(_step_fn(net), net.parameters(), opt, capture=False)
    stepper(batch)
    reduced = stepper.grads.flat.clone()
    opt2 = torch.optim.AdamW(net.parameters(), lr=1e-2)
    stepper2 = GTS(_step_fn(net), net.parameters(), opt2, capture=False)
    for i in range(3):
        stepper2(_batch(200 + 10 * i + rank))
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    params = [torch.zeros_like(flat) for _ in range(world)]
    lgs = [torch.zeros_like(lg) for _ in range(world)]
    dist.all_gather(params, flat)
    dist.all_gather(lgs, lg)
    if rank == 0:
        torch.save({"params": params, "local": lgs, "reduced": reduced}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_flat_gradient_allreduce(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    torch.testing.assert_close(res["reduced"], (res["local"][0] + res["local"][1]) / 2, atol=1e-7, rtol=1e-6)
    assert torch.equal(res["params"][0], res["params"][1])


def test_scheduled_gc_freezes_disables_and_restores():
    """runtime.ScheduledGC: automatic collector off inside, young-generation pass every `every` steps, state restored."""
    import gc
    from runtime import ScheduledGC
    assert gc.isenabled()
    with ScheduledGC(every=2) as sgc:
        assert not gc.isenabled() and gc.get_freeze_count() > 0

        class Node:
            pass
        a, b = Node(), Node()
        a.other, b.other = b, a            # a reference cycle only the collector can free
        import weakref
        w = weakref.ref(a)
        del a, b
        sgc.step()
        assert w() is not None             # no pass yet, and the automatic collector is off
        sgc.step()
        assert w() is None                 # collected at the second step boundary
    assert gc.isenabled() and gc.get_freeze_count() == 0


def test_scheduled_gc_full_pass_frees_promoted_cycles_and_keeps_a_disabled_collector_off():
    """a cycle that survives a young pass sits in the oldest generation, which gc.collect(1) never visits — every
    `full_every`-th scheduled pass is a full one; __exit__ restores gc.isenabled() instead of enabling unconditionally."""
    import gc
    import weakref
    from runtime import ScheduledGC

    class Node:
        pass
    with ScheduledGC(every=1, full_every=3) as sgc:
        a, b = Node(), Node()
        a.other, b.other = b, a
        w = weakref.ref(a)
        sgc.step()                          # young pass while the cycle is ALIVE: promoted to the oldest generation
        sgc.step()                          # (twice: generation 0 -> 1 -> 2)
        del a, b
        gc.collect(1)
        assert w() is not None              # what the old schedule did forever: the young pass cannot see it
        sgc.step()                          # third scheduled pass = full collection
        assert w() is None
    gc.disable()
    try:
        with ScheduledGC(every=5):
            pass
        assert not gc.isenabled()           # the caller had it off: stays off
    finally:
        gc.enable()
