"""Generate golden fixtures by running the REFERENCE's own python layer
(/root/reference, read-only, only available in the build container) on top of the
CPU oracle injected as its native extension.  The fixtures (inputs + expected
outputs, never reference source) are committed next to this script; the tests
that consume them run anywhere.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

What the fixtures pin: the build's python layer (pointnet2_ops mirror, MSG encoder,
GF3D SA/FP backbone) == the reference's python layer given identical native-op
results, including parameter names/shapes/initialisation order.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [REF, os.path.join(REF, "scene_graph_prediction/pointnet2_dir/pointnet2_ops_lib"), REPO]

from oracle.oracle import OracleExt  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# the reference's native extension modules -> the oracle
ext = _stub("pointnet2_ops._ext", **{k: getattr(OracleExt, k) for k in dir(OracleExt) if not k.startswith("_")})
_stub("pointnet2")
sys.modules["pointnet2._ext"] = ext
# third-party packages the reference imports at module scope but never uses on this path
_stub("pytorch_lightning", LightningModule=torch.nn.Module, seed_everything=lambda *a, **k: None)
_stub("torchvision", transforms=_stub("torchvision.transforms", Compose=lambda x: x))
for missing in ("lmdb", "msgpack_numpy", "h5py"):
    _stub(missing, patch=lambda: None)
_stub("tqdm", tqdm=lambda x, *a, **k: x)

import warnings  # noqa: E402
warnings.simplefilter("ignore")

from pointnet2_ops import pointnet2_modules as ref_pm  # noqa: E402  (the REFERENCE package)
assert ref_pm.__file__.startswith(REF)


def cloud(B, N, C, seed):
    g = torch.Generator().manual_seed(seed)
    pc = torch.rand(B, N, 3 + C, generator=g) * 2 - 1
    return pc


def sums(sd):
    return {k: float(v.double().sum()) for k, v in sd.items()}


def pack_sd(sd, prefix):
    return {prefix + k: v.detach().numpy() for k, v in sd.items()}


def sd_manifest(sd, prefix):
    """keys / shapes / per-tensor sums: the consumer rebuilds the module under the same seed (same initialisation
    order => same weights) instead of shipping megabytes of random numbers."""
    return {prefix + "keys": np.array(list(sd.keys())), prefix + "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]),
            prefix + "sums": np.array([float(v.double().sum()) for v in sd.values()])}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def fixture_sa_msg():
    pc = cloud(2, 600, 3, 1)
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    torch.manual_seed(11)
    sa = ref_pm.PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[8, 16],
                                    mlps=[[3, 16, 16], [3, 16, 32]], use_xyz=True)
    sd0 = {k: v.clone() for k, v in sa.state_dict().items()}
    sa.train()
    f = feats.clone().requires_grad_(True)
    nx, nf = sa(xyz, f)
    (nf * torch.linspace(0.5, 1.5, nf.numel()).view_as(nf)).sum().backward()
    sd1 = sa.state_dict()
    sa.eval()
    _, nf_eval = sa(xyz, feats)
    save("sa_msg.npz", pc=pc.numpy(), new_xyz=nx.detach().numpy(), new_features_train=nf.detach().numpy(),
         grad_features=f.grad.numpy(), new_features_eval=nf_eval.detach().numpy(),
         **pack_sd(sd0, "sd0/"), **{"sd1/" + k: v.numpy() for k, v in sd1.items() if "running" in k or "num_b" in k})


def fixture_fp():
    u = cloud(2, 200, 5, 2)
    k = cloud(2, 48, 9, 3)
    unknown, uf = u[..., :3].contiguous(), u[..., 3:].transpose(1, 2).contiguous()
    known, kf = k[..., :3].contiguous(), k[..., 3:].transpose(1, 2).contiguous()
    torch.manual_seed(12)
    fp = ref_pm.PointnetFPModule(mlp=[14, 16, 8])
    sd0 = {kk: v.clone() for kk, v in fp.state_dict().items()}
    fp.train()
    kfr = kf.clone().requires_grad_(True)
    out = fp(unknown, known, uf, kfr)
    out.square().sum().backward()
    save("fp.npz", unknown=u.numpy(), known=k.numpy(), out_train=out.detach().numpy(), grad_known=kfr.grad.numpy(),
         **pack_sd(sd0, "sd0/"))


def fixture_msg_encoder():
    from scene_graph_prediction.scene_graph_helpers.model.pointnets.network_PointNet2 import PointNetfeat
    out = {}
    for dim, n, seed in ((6, 1024, 21), (7, 1536, 22)):
        torch.manual_seed(seed)
        enc = PointNetfeat(input_dim=dim, out_size=256, input_dropout=0.0)
        enc.eval()
        pc = cloud(2, n, dim - 3, seed)
        if dim == 7:
            pc[..., 6] = torch.randint(0, 3, pc[..., 6].shape).float()
        x = pc.transpose(1, 2).contiguous()                     # (B, dim, N) like collate_fn
        with torch.no_grad():
            y = enc(x)
        sd = enc.state_dict()
        out[f"d{dim}/x"] = x.numpy()
        out[f"d{dim}/y"] = y.numpy()
        out[f"d{dim}/keys"] = np.array(list(sd.keys()))
        out[f"d{dim}/shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[f"d{dim}/sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    # config-1 plumbing case: one 20k-point cloud (BASELINE.json configs[0])
    torch.manual_seed(23)
    enc = PointNetfeat(input_dim=6, out_size=256, input_dropout=0.0).eval()
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1, 20000, 3, generator=g)
    p = p / p.norm(dim=2, keepdim=True) * torch.rand(1, 20000, 1, generator=g).pow(1 / 3)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).max()
    pc = torch.cat([p, torch.rand(1, 20000, 3, generator=g)], dim=2)
    with torch.no_grad():
        y = enc(pc.transpose(1, 2).contiguous())
    out["cfg1/seed_cloud"] = np.array([0, 20000])
    out["cfg1/y"] = y.numpy()
    out["cfg1/pc_checksum"] = np.array([float(pc.double().sum())])
    save("msg_encoder.npz", **out)


def fixture_gf3d_backbone():
    # import backbone_module WITHOUT running GF3D/models/__init__.py (it pulls the detector,
    # losses and the dataset, which need open3d): pre-seed the package with its search path only
    import external_src.group_free_3D  # noqa: F401
    pkg = _stub("external_src.group_free_3D.models")
    pkg.__path__ = [os.path.join(REF, "external_src/group_free_3D/models")]
    from external_src.group_free_3D.models.backbone_module import Pointnet2Backbone
    torch.manual_seed(31)
    net = Pointnet2Backbone(input_feature_dim=3)
    sd = net.state_dict()
    sd_keys, sd_shapes = list(sd.keys()), [str(tuple(v.shape)) for v in sd.values()]
    sd_sums = [float(v.double().sum()) for v in sd.values()]      # BEFORE the training forward touches BN stats
    pc = cloud(2, 2600, 3, 32)
    net.train()
    x = pc.clone()
    ep = net(x)
    feats = ep["fp2_features"]
    loss = (feats * torch.linspace(0.5, 1.5, feats.numel()).view_as(feats)).mean()
    loss.backward()
    gw = net.sa1.mlp_module.layer0.conv.weight.grad
    gw2 = net.fp2.mlp.layer1.conv.weight.grad
    save("gf3d_backbone.npz", pc=pc.numpy(), keys=np.array(sd_keys), shapes=np.array(sd_shapes),
         sums=np.array(sd_sums),
         sa1_inds=ep["sa1_inds"][:, :256].numpy(), sa2_inds=ep["sa2_inds"][:, :64].numpy(),
         sa4_xyz=ep["sa4_xyz"].detach().numpy(), sa4_features=ep["sa4_features"].detach().numpy()[:, ::4],
         fp2_features=feats.detach().numpy()[:, ::8, ::4], loss=np.array([float(loss)]),
         grad_sa1_conv0=gw.numpy(), grad_fp2_conv1=gw2.numpy()[::4])


def fixture_votes_pooling():
    """GF3D PointnetSAModuleVotes with the three pooling modes (GF3D/pointnet2/pointnet2_modules.py:236-248), the
    normalize_xyz flag and a caller-supplied `inds`; train-mode forward + backward."""
    from external_src.group_free_3D.pointnet2 import pointnet2_modules as ref_gf
    assert ref_gf.__file__.startswith(REF)
    out = {}
    pc = cloud(2, 700, 5, 41)
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    out["pc"] = pc.numpy()
    for pooling, norm, sigma in (("max", True, None), ("avg", False, None), ("rbf", True, None), ("rbf", False, 0.11)):
        tag = f"{pooling}_{int(norm)}_{'d' if sigma is None else 's'}"
        torch.manual_seed(42)
        sa = ref_gf.PointnetSAModuleVotes(mlp=[5, 16, 24], npoint=48, radius=0.35, nsample=12, use_xyz=True,
                                          pooling=pooling, sigma=sigma, normalize_xyz=norm)
        sd0 = {k: v.clone() for k, v in sa.state_dict().items()}
        sa.train()
        f = feats.clone().requires_grad_(True)
        nx, nf, inds = sa(xyz, f)
        (nf * torch.linspace(0.5, 1.5, nf.numel()).view_as(nf)).sum().backward()
        out[f"{tag}/new_xyz"], out[f"{tag}/new_features"] = nx.detach().numpy(), nf.detach().numpy()
        out[f"{tag}/inds"], out[f"{tag}/grad_features"] = inds.numpy(), f.grad.numpy()
        out[f"{tag}/grad_w0"] = sa.mlp_module.layer0.conv.weight.grad.numpy()
        out.update(pack_sd(sd0, f"{tag}/sd0/"))
    save("votes_pooling.npz", **out)


def fixture_heads():
    """PointNetCls / PointNetRelCls (SGH/model/pointnets/network_PointNet.py:188-271): initialisation (xavier_normal
    under a fixed seed), eval-mode outputs, and a train-mode forward/backward of the BatchNorm variant without dropout;
    the relation head with the subject/object one-hot and with the image late fusion."""
    from scene_graph_prediction.scene_graph_helpers.model.pointnets import network_PointNet as ref_pn
    assert ref_pn.__file__.startswith(REF)
    out = {}
    g = torch.Generator().manual_seed(51)
    x = torch.randn(11, 256, generator=g)
    onehot = torch.zeros(11, 12)
    onehot[torch.arange(11), torch.randint(0, 6, (11,), generator=g)] = 1
    onehot[torch.arange(11), 6 + torch.randint(0, 6, (11,), generator=g)] = 1
    img = torch.randn(768, generator=g)
    out["x"], out["onehot"], out["img"] = x.numpy(), onehot.numpy(), img.numpy()

    torch.manual_seed(52)
    cls = ref_pn.PointNetCls(12, in_size=256, batch_norm=False, drop_out=True).eval()
    out.update(sd_manifest(cls.state_dict(), "cls/"))
    out["cls/y_eval"] = cls(x).detach().numpy()

    torch.manual_seed(53)
    rel = ref_pn.PointNetRelCls(15, in_size=256, batch_norm=False, drop_out=True, image_embedding_size=None,
                                n_object_types=6).eval()
    out.update(sd_manifest(rel.state_dict(), "rel/"))
    out["rel/y_eval"] = rel(x, relation_objects_one_hot=onehot).detach().numpy()

    torch.manual_seed(54)
    reli = ref_pn.PointNetRelCls(15, in_size=256, batch_norm=False, drop_out=True, image_embedding_size=768,
                                 n_object_types=6).eval()
    out.update(sd_manifest(reli.state_dict(), "reli/"))
    out["reli/y_eval"] = reli(x, relation_objects_one_hot=onehot, image_embeddings=img).detach().numpy()

    torch.manual_seed(55)
    bn = ref_pn.PointNetRelCls(15, in_size=256, batch_norm=True, drop_out=False, image_embedding_size=None,
                               n_object_types=6).train()
    out.update(sd_manifest(bn.state_dict(), "relbn/"))
    xx = x.clone().requires_grad_(True)
    y = bn(xx, relation_objects_one_hot=onehot)
    (y * torch.linspace(0.5, 1.5, y.numel()).view_as(y)).sum().backward()
    out["relbn/y_train"], out["relbn/grad_x"] = y.detach().numpy(), xx.grad.numpy()
    out["relbn/grad_fc1"] = bn.fc1.weight.grad.numpy()[::8, ::8]
    save("heads.npz", **out)


def fixture_sample_uniformly():
    """GF3D QueryAndGroup(sample_uniformly=True, ret_unique_cnt=True) (GF3D/pointnet2/pointnet2_utils.py:327-339): the
    reference's host loop under torch.manual_seed.  The product resamples on the device with its own counter-based
    generator, so only the deterministic parts are compared exactly (leading unique indices, unique_cnt); the fixture
    also keeps the reference's full index tensor for the distribution check."""
    from external_src.group_free_3D.pointnet2 import pointnet2_utils as ref_gu
    assert ref_gu.__file__.startswith(REF)
    pc = cloud(2, 900, 2, 61)
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    new_xyz = xyz[:, :40].contiguous()
    qg = ref_gu.QueryAndGroup(0.3, 16, use_xyz=True, ret_grouped_xyz=True, sample_uniformly=True, ret_unique_cnt=True)
    torch.manual_seed(62)
    grouped, grouped_xyz, cnt = qg(xyz, new_xyz, feats)
    plain = ref_gu.ball_query(0.3, 16, xyz, new_xyz)
    save("sample_uniformly.npz", pc=pc.numpy(), unique_cnt=cnt.numpy(), ball_idx=plain.numpy(),
         grouped=grouped.numpy(), grouped_xyz=grouped_xyz.numpy())


if __name__ == "__main__":
    which = sys.argv[1:] or ["sa_msg", "fp", "msg_encoder", "gf3d_backbone", "votes_pooling", "heads", "sample_uniformly"]
    for name in which:
        globals()["fixture_" + name]()
