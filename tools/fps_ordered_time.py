"""Sampling of clouds that are a sampling order (the centres of the SA level above): pn2_furthest_point_sampling against
pn2_furthest_point_sampling_ordered (verification on the device, rounds only from the first unverified one) at the three
lower levels of the headline, and how often a cloud's sampling deviates from 0 .. m-1 (an exact fp32 tie that the other
point wins).  A call from python costs the host ~50 us here (two allocations, two C calls): the small shapes show that floor.
    python tools/fps_ordered_time.py            (GPU box; one JSON line per row)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4d-or_amd"))
from pointnet2_ops import _ext  # noqa: E402


def clouds(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    return (p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)).cuda()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


def first_unverified(x, m):
    lib = _ext._lib
    B, n = x.size(0), x.size(1)
    nb = int(lib.pn2_fps_ordered_workspace_bytes(B, n, m))
    ws = torch.zeros(nb // 4, dtype=torch.float32, device="cuda")
    out = torch.zeros(B, m, dtype=torch.int32, device="cuda")
    _ext._call("pn2_furthest_point_sampling_ordered", x, B, n, m, x.data_ptr(), ws.data_ptr(), nb, out.data_ptr(), 0)
    torch.cuda.synchronize()
    base = (int(lib.pn2_fps_workspace_bytes(B, n, m)) + 255) // 256 * 256
    off = (base + (B * m * 4 + 255) // 256 * 256) // 4
    return [int(v) for v in ws.view(torch.int32)[off:off + B].cpu().tolist() if v < m]


if __name__ == "__main__":
    p = clouds(32, 50000, 0)
    order = torch.gather(p, 1, _ext.furthest_point_sampling(p, 2048).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    for n, m in ((2048, 1024), (1024, 512), (512, 256)):
        x = order[:, :n].contiguous()
        assert torch.equal(_ext.furthest_point_sampling(x, m), _ext.furthest_point_sampling(x, m, ordered=True))
        print(json.dumps({"clouds": 32, "points": n, "samples": m, "plain_us": round(timeit(lambda: _ext.furthest_point_sampling(x, m)), 1),
                          "ordered_us": round(timeit(lambda: _ext.furthest_point_sampling(x, m, ordered=True)), 1),
                          "first_unverified_rounds": first_unverified(x, m)}), flush=True)
    dev = tot = 0
    ar = torch.arange(1024, dtype=torch.int32, device="cuda")
    for rep in range(8):
        q = clouds(32, 20000, 100 + rep)
        oo = torch.gather(q, 1, _ext.furthest_point_sampling(q, 2048).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        plain = _ext.furthest_point_sampling(oo, 1024)
        assert torch.equal(plain, _ext.furthest_point_sampling(oo, 1024, ordered=True))
        dev += int((plain != ar).any(dim=1).sum()); tot += 32
    print(json.dumps({"clouds_sampled_1024_of_2048": tot, "deviate_from_prefix": dev}))
