"""The one-kernel eval SA level (pn2_sa_eval_x3, csrc/x3_chain.hip) alone at the headline's four SA shapes:
time, fp32-equivalent TF/s, fraction of the f32x3 ceiling (dense bf16 MFMA / 6).  python tools/sa_eval_bench.py [out.jsonl] [level]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch  # noqa: E402

from pointnet2_ops import _ext as e  # noqa: E402

dev = "cuda"
X3_PEAK = 2500.0 / 6.0


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


LEVELS = [("SA1", 32, 50000, 2048, 64, 3, 0.2, (64, 64, 128), 0), ("SA2", 32, 2048, 1024, 32, 128, 0.4, (128, 128, 256), 1),
          ("SA3", 32, 1024, 512, 16, 256, 0.8, (128, 128, 256), 1), ("SA4", 32, 512, 256, 16, 256, 1.2, (128, 128, 256), 1)]
only = sys.argv[2] if len(sys.argv) > 2 else None
rows = []
for name, B, N, m, ns, C, r, (c1, c_mid, c_out), mode in LEVELS:
    if only and name != only:
        continue
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    new_xyz = xyz[:, :m].contiguous()
    idx = e.ball_query(new_xyz, xyz, r, ns)
    W1 = (torch.randn(c_mid, c1, generator=g) * 0.1).to(dev)
    W2 = (torch.randn(c_out, c_mid, generator=g) * 0.1).to(dev)
    stream = torch.cat([e.x3_pack_weight(W1, True if mode == 0 else False), e.x3_pack_weight(W2, True)])
    b1, b2 = torch.randn(c_mid, generator=g).to(dev), torch.randn(c_out, generator=g).to(dev)
    out = torch.empty(B, m, c_out, device=dev)
    if mode == 0:
        f = torch.rand(B, N, C, generator=g).to(dev)
        M0 = torch.zeros(c1, 16, device=dev)
        M0[:, :3 + C + 1] = torch.randn(c1, 3 + C + 1, generator=g).to(dev) * 0.3
        w0 = e.x3_pack_weight(M0, False)
        fn = lambda: e.sa_eval_x3(0, xyz, new_xyz, idx, f, None, c1, w0, c_mid, stream, b1, b2, out)
    else:
        Pq = torch.randn(B * N, c1, generator=g).to(dev)
        Q = torch.randn(B * m, c1, generator=g).to(dev) * 0.5
        fn = lambda: e.sa_eval_x3(1, xyz, new_xyz, idx, Pq, Q, c1, None, c_mid, stream, b1, b2, out)
    ms = timed(fn)
    rows_n = B * m * ns
    flops = 2 * rows_n * ((16 * c1 if mode == 0 else 0) + c1 * c_mid + c_mid * c_out)
    t = {"level": name, "rows": rows_n, "widths": [c1, c_mid, c_out], "ns": ns, "ms": round(ms, 4), "TFLOPps_fp32_equivalent": round(flops / ms / 1e9, 1),
         "frac_of_x3_peak": round(flops / ms / 1e9 / X3_PEAK, 3)}
    rows.append(t)
    print(json.dumps(t))
if len(sys.argv) > 1 and sys.argv[1] != "-":
    with open(sys.argv[1], "w") as fh:
        for t in rows:
            fh.write(json.dumps(t) + "\n")
