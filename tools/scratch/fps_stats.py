import os, sys, ctypes
REPO = "/root/repo"
os.environ["PN2_HIP_LIB"] = REPO + "/4d-or_amd/csrc/exp/libpn2_fpsstats.so"
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from pointnet2_ops import _ext as e
from microbench import unit_ball
B, N, m = 32, 50000, 2048
x = unit_ball(B, N, 3).cuda()
nb = int(e._lib.pn2_fps_workspace_bytes(B, N, m))
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
idx = torch.empty(B, m, dtype=torch.int32, device="cuda")
e._lib.pn2_fps_set_plan_override(1, 2, 1, 1024, 0)
rc = e._lib.pn2_furthest_point_sampling_ex(B, N, m, e._ptr(x), e._ptr(ws), nb, e._ptr(idx), 0, None)
torch.cuda.synchronize()
off = int(e._lib.pn2_fps_status_offset(B, N, m))
st = ws[off:off + 16].view(torch.int32).cpu()
print("rc", rc, "status", st.tolist(), "active fraction", st[1].item() / max(1, st[2].item()))
