import os, sys
sys.path[:0] = ["/root/repo/4d-or_amd", "/root/repo", "/root/repo/tools"]
import torch
from pointnet2_ops import _ext
from microbench import unit_ball, timeit
dev = torch.device("cuda:0")
xyz = unit_ball(32, 50000).to(dev)
sel = _ext.furthest_point_sampling(xyz, 2048)
new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for cpw in (1, 2, 4, 8):
    os.environ["PN2_BQ_CPW"] = str(cpw)
    t = timeit(lambda: _ext.ball_query(new_xyz, xyz, 0.2, 64))
    print("cpw", cpw, "ms", round(t * 1e3, 4))
# how far do scans go?  first index of the 64th hit
idx = _ext.ball_query(new_xyz, xyz, 0.2, 64).long()
last = idx.max(dim=2).values.float()
print("mean last-hit index", float(last.mean()), "max", float(last.max()), "p90", float(last.flatten().kthvalue(int(0.9 * last.numel())).values))
full = (idx[:, :, -1] != idx[:, :, 0]).float().mean()
print("fraction of centres with a full 64-neighbourhood", float(full))
