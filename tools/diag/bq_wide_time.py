import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "4d-or_amd"), REPO]
import torch
from pointnet2_ops import _ext as e
B, N, m, ns, r = 32, 50000, 2048, 64, 0.2
g = torch.Generator().manual_seed(0)
p = torch.randn(B, N, 3, generator=g)
xyz = (p / p.norm(dim=2, keepdim=True) * torch.rand(B, N, 1, generator=g).pow(1 / 3)).cuda()
sel = e.furthest_point_sampling(xyz, m).long()
new_xyz = torch.gather(xyz, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
feats = torch.rand(B, N, 3, device="cuda")
for w in (4, 8):
    for _ in range(5):
        e.ball_query_group(new_xyz, xyz, feats, r, ns, True, True, slab_w=w)
    torch.cuda.synchronize()
