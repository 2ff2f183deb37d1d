"""Phase breakdown of the cooperative FPS round (needs the instrumented experiment build, PN2_HIP_LIB)."""
import os, sys, ctypes
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext
sys.path.insert(0, os.path.join(REPO, "tools"))
from microbench import unit_ball
dev = torch.device("cuda:0")
B, N, m = 32, 50000, 2048
x = unit_ball(B, N, 3).to(dev)
lib = _ext._lib
ws_bytes = int(lib.pn2_fps_workspace_bytes(B, N, m))
ws = torch.zeros(ws_bytes // 8, dtype=torch.int64, device=dev)
out = torch.zeros(B, m, dtype=torch.int32, device=dev)
for _ in range(2):
    rc = lib.pn2_furthest_point_sampling(B, N, m, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws_bytes),
                                         ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
rc = lib.pn2_furthest_point_sampling(B, N, m, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws_bytes),
                                     ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
e.record(); torch.cuda.synchronize()
print("rc", rc, "ms", s.elapsed_time(e))
cloud_bytes = (ws_bytes - 256) // B
base = (B * cloud_bytes + 8) // 8
names = ["scan+wave-reduce", "barrier1", "block-reduce+publish", "sweep(poll)", "final-reduce", "barrier2"]
for w, off in (("wave0 (exchanger)", 0), ("wave5 (worker)", 8)):
    prof = ws[base + off: base + off + 6].cpu().tolist()
    tot = sum(prof)
    print(w, "ticks/round:", {n: round(p / (m - 1)) for n, p in zip(names, prof)}, "total", round(tot / (m - 1)), "(100 MHz ticks => x10 ns)")
