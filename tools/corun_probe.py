"""Where does a forward GEMM lose time when the cooperative FPS co-runs?  (needs the -DPN2_EXP_CFG build:
PN2_HIP_LIB=build/exp/libpn2_exp.so python tools/corun_probe.py)  Per-workgroup wall-clock spans and hardware ids of
one SA1-l3-shaped GEMM launch, alone and next to the FPS of 32 x 50k points, split by whether the workgroup's CU
also hosts an FPS workgroup."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
from pointnet2_ops import _ext  # noqa: E402

lib = ctypes.CDLL(_ext.LIB_PATH)


def cu_key(hw, xcc):
    # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
    return (int(xcc) & 0xF, (int(hw) >> 13) & 0x7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xF)


def gemm_dump(nwg):
    buf = np.zeros(4 * 4096, dtype=np.uint64)
    assert lib.pn2_dbg_gemm_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    b = buf.reshape(4096, 4)[:nwg]
    return b


def fps_dump(nwg):
    buf = np.zeros(2 * 1024, dtype=np.uint32)
    assert lib.pn2_dbg_fps_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    return buf.reshape(1024, 2)[:nwg]


def main():
    dev = torch.device("cuda")
    M, K, N = [int(v) for v in os.environ.get("PROBE_SHAPE", "4194304,64,128").split(",")]
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.1
    p = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1)
    xyz = torch.rand(32, int(os.environ.get("PROBE_N", "50000")), 3, device=dev)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def gemm():
        stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
        return _ext.mlp_gemm(x, W, pro=_ext.PRO_BNRELU, epi=_ext.EPI_STATS, p=p, stats=stats)

    def run(corun):
        torch.cuda.synchronize()
        if corun:
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                _ext.furthest_point_sampling(xyz, 2048)
            time.sleep(0.0005)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        for i in range(3):
            gemm()
            ev[i + 1].record()
        torch.cuda.synchronize()
        print("   six launches, us:", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(3)])
        return ev[2].elapsed_time(ev[3])

    for _ in range(2):
        run(False)
    for label, corun in (("alone", False), ("co-run", True)):
        ms = run(corun)
        nwg = 768 if os.environ.get("PN2_GEMM_GRID") is None else int(os.environ["PN2_GEMM_GRID"])
        g = gemm_dump(4096)
        g = g[g[:, 0] != 0]
        dur = (g[:, 1] - g[:, 0]).astype(np.float64) / 100.0        # us (100 MHz wall clock)
        start = (g[:, 0] - g[:, 0].min()).astype(np.float64) / 100.0
        keys = [cu_key(h, xc) for h, xc in zip(g[:, 2], g[:, 3])]
        late = int((start > 0.25 * dur.max()).sum())
        print(f"== {label}: {late} of {len(dur)} workgroups start late; kernel {ms * 1e3:.0f} us; {len(set(keys))} distinct CUs; per-WG span us: "
              f"min {dur.min():.0f} p50 {np.median(dur):.0f} max {dur.max():.0f}; last start {start.max():.0f}")
        if corun:
            f = fps_dump(256 if os.environ.get("PN2_FPS_G") != "4" else 128)
            fk = set(cu_key(h, xc) for h, xc in f)
            on = np.array([k in fk for k in keys])
            print(f"   FPS workgroups on {len(fk)} CUs; GEMM WGs sharing a CU with FPS: {on.sum()} / {len(on)}")
            for name, m in (("with FPS", on), ("without FPS", ~on)):
                if m.any():
                    print(f"   {name:12s}: n {m.sum():4d}  span p50 {np.median(dur[m]):.0f} mean {dur[m].mean():.0f} "
                          f"max {dur[m].max():.0f}  start p50 {np.median(start[m]):.0f}")
            per_cu = {}
            for k in keys:
                per_cu[k] = per_cu.get(k, 0) + 1
            cnt_on = [v for k, v in per_cu.items() if k in fk]
            cnt_off = [v for k, v in per_cu.items() if k not in fk]
            print(f"   GEMM WGs per CU: with FPS {np.bincount(cnt_on).tolist() if cnt_on else []}, "
                  f"without {np.bincount(cnt_off).tolist() if cnt_off else []}; CUs with FPS and no GEMM WG: "
                  f"{len(fk - set(keys))}")


main()
