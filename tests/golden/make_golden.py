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


def fixture_gf3d_backbone_eval():
    """The reference's Pointnet2Backbone in EVAL mode (BatchNorm on running statistics: the inference path, main.py:84-117):
    weights from the seed, running statistics / affine parameters from tests/fixture_checks.py::seed_running_stats (a host
    generator: the consumer rebuilds the same module instead of loading megabytes)."""
    import external_src.group_free_3D  # noqa: F401
    pkg = sys.modules.get("external_src.group_free_3D.models") or _stub("external_src.group_free_3D.models")
    pkg.__path__ = [os.path.join(REF, "external_src/group_free_3D/models")]
    from external_src.group_free_3D.models.backbone_module import Pointnet2Backbone
    import importlib.util
    spec = importlib.util.spec_from_file_location("fixture_checks", os.path.join(REPO, "tests", "fixture_checks.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    torch.manual_seed(31)
    net = Pointnet2Backbone(input_feature_dim=3)
    assert type(net).__module__.startswith("external_src") and sys.modules[type(net).__module__].__file__.startswith(REF)
    fc.seed_running_stats(net, 77)
    net.eval()
    pc = cloud(2, 2600, 3, 33)
    with torch.no_grad():
        ep = net(pc.clone())
    save("gf3d_backbone_eval.npz", pc=pc.numpy(), sa1_inds=ep["sa1_inds"][:, :256].numpy(), sa4_xyz=ep["sa4_xyz"].numpy(),
         sa4_features=ep["sa4_features"].numpy()[:, ::4], fp2_features=ep["fp2_features"].numpy()[:, ::8, ::4])


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


# ---------------------------------------------------------------------------------------------------------------------
# third-party semantics the reference's GCN / model files import at module scope and that are absent from the image AND
# from /root/reference (pins: torch_geometric 2.0.2, torch_scatter 2.0.9 — README.md:87; pytorch_lightning 1.6.0,
# timm 0.4.12 — SGP/requirements.txt:17,19).  They are RESTATED here from their published behaviour, in the few lines
# the reference's call sites use (network_TripletGCN.py:7-8,30-32,41,57); everything else in the fixtures below is the
# reference's own python (network_TripletGCN.py, scene_graph_prediction_model.py, network_PointNet*.py) running unchanged.
def _third_party_stubs():
    import inspect

    class MessagePassing(torch.nn.Module):
        """torch_geometric.nn.conv.MessagePassing 2.0.2, the part `propagate` runs for a dense `edge_index`:
        flow='source_to_target' => (i, j) = (1, 0); an argument `<name>_i` / `<name>_j` of `message` receives
        `kwargs[name].index_select(node_dim, edge_index[i | j])`; other arguments pass through; `aggregate(msg, index =
        edge_index[i], ptr=None, dim_size = number of nodes)`; `update` is the identity."""

        def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
            super().__init__()
            self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

        def propagate(self, edge_index, size=None, **kwargs):
            i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
            n_nodes = None
            args = {}
            for name in inspect.signature(self.message).parameters:
                if name.endswith(("_i", "_j")):
                    data = kwargs[name[:-2]]
                    n_nodes = data.size(self.node_dim)
                    args[name] = data.index_select(self.node_dim, edge_index[i if name.endswith("_i") else j])
                else:
                    args[name] = kwargs[name]
            out = self.message(**args)
            out = self.aggregate(out, index=edge_index[i], ptr=None, dim_size=n_nodes)
            return self.update(out)

        def update(self, inputs):
            return inputs

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        """torch_scatter.scatter 2.0.9 for reduce in {'add', 'sum'} along a leading node dimension."""
        assert reduce in ("add", "sum") and src.dim() == 2 and dim in (0, -2)
        res = torch.zeros(dim_size, src.size(1), dtype=src.dtype)
        return res.index_add_(0, index, src)

    tg = _stub("torch_geometric")
    tg.nn = _stub("torch_geometric.nn")
    tg.nn.conv = _stub("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    _stub("torch_scatter", scatter=scatter)

    class _NoCNN(torch.nn.Module):
        """Stands where timm's EfficientNet-B5 would: `num_features` and a parameter-free `conv_head`, so that building it
        draws nothing from the RNG and adds no state_dict entry (the product skips `full_image_model.*` the same way)."""
        num_features = 2048

        def __init__(self):
            super().__init__()
            self.conv_head = torch.nn.Identity()

        def forward(self, x):
            raise RuntimeError("the 2-D CNN is out of scope: the fixture feeds pre-computed image features")

    timm = _stub("timm", create_model=lambda *a, **k: _NoCNN())
    timm.data = _stub("timm.data", resolve_data_config=lambda *a, **k: {}, create_transform=lambda **k: None)


def _ref_gcn():
    _third_party_stubs()
    from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as ref_gcn
    assert ref_gcn.__file__.startswith(REF)
    return ref_gcn


def _min_relu_margin(model, x, e, ei):
    """Smallest |input| any ReLU (module or functional) sees in a float64 forward of a copy of `model`."""
    import copy
    import torch.nn.functional as F
    m64 = copy.deepcopy(model).double().train()
    seen = []
    real = F.relu

    def spy(inp, inplace=False):
        if float(inp.detach().min()) < 0:        # (the edge output of nn1 is already rectified when the model rectifies it again)
            seen.append(float(inp.detach().abs().min()))
        return real(inp, inplace=inplace)

    F.relu = spy
    try:
        with torch.no_grad():
            m64(x.double(), e.double(), ei)
    finally:
        F.relu = real
    return min(seen)


def _full_edges(n):
    return torch.tensor([[a, b] for a in range(n) for b in range(n) if a != b]).t().contiguous()


def fixture_triplet_gcn():
    """The reference's TripletGCN / TripletGCNModel (SGH/model/gcns/network_TripletGCN.py:30-80) on the restated
    MessagePassing / scatter: initialisation order, forward, gradients; 2 and 3 layers; eval mode (BatchNorm1d with
    track_running_stats=False keeps batch statistics); the survey's hand case edge_index = [[0,1,2],[2,1,0]]
    (network_util.py:86-94) and an irregular edge list with an isolated node and a repeated edge."""
    ref_gcn = _ref_gcn()
    out = {}
    for tag, layers, n, dn, de, dh, seed in (("l2", 2, 9, 256, 256, 512, 71), ("l3", 3, 6, 64, 48, 96, 72)):
        torch.manual_seed(seed)
        model = ref_gcn.TripletGCNModel(num_layers=layers, dim_node=dn, dim_edge=de, dim_hidden=dh)
        sd0 = model.state_dict()
        out.update(sd_manifest(sd0, f"{tag}/"))
        ei = _full_edges(n)
        # inputs are drawn until no ReLU input of the (float64) forward lies within 2e-5 of the kink: a pre-activation of
        # 3e-7 takes either side in two correct fp32 implementations and moves a whole gradient row by 1e-2 (measured:
        # the first draw had |pre| = 2.5e-7 and 6.7e-7, and the fp32 reference itself was 4.7e-3 off the fp64 gradient)
        for draw in range(200):
            g = torch.Generator().manual_seed(seed + 100 + draw)
            x = torch.randn(n, dn, generator=g)
            e = torch.randn(ei.size(1), de, generator=g)
            if _min_relu_margin(model, x, e, ei) > 2e-5:
                break
        else:
            raise RuntimeError("no kink-free draw")
        out[f"{tag}/input_seed"] = np.array([seed + 100 + draw])
        x.requires_grad_(True)
        e.requires_grad_(True)
        model.train()
        ox, oe = model(x, e, ei)
        wx = torch.linspace(0.5, 1.5, ox.numel()).view_as(ox)
        we = torch.linspace(-1.0, 1.0, oe.numel()).view_as(oe)
        ((ox * wx).sum() + (oe * we).sum()).backward()
        out[f"{tag}/x"], out[f"{tag}/e"], out[f"{tag}/ei"] = x.detach().numpy(), e.detach().numpy(), ei.numpy()
        out[f"{tag}/out_x"], out[f"{tag}/out_e"] = ox.detach().numpy(), oe.detach().numpy()
        out[f"{tag}/grad_x"], out[f"{tag}/grad_e"] = x.grad.numpy(), e.grad.numpy()
        names = [k for k, _ in model.named_parameters()]
        out[f"{tag}/grad_names"] = np.array(names)
        out[f"{tag}/grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        out[f"{tag}/grad_sums"] = np.array([float(p.grad.double().sum()) for _, p in model.named_parameters()])
        last = f"gconvs.{layers - 1}.nn2.3.weight"
        out[f"{tag}/grad_last_w"] = dict(model.named_parameters())[last].grad.numpy()[::4, ::8]
        out[f"{tag}/grad_first_w"] = dict(model.named_parameters())["gconvs.0.nn1.0.weight"].grad.numpy()[::16, ::16]
        model.eval()
        with torch.no_grad():
            ex, ee = model(x.detach(), e.detach(), ei)
        out[f"{tag}/eval_x"], out[f"{tag}/eval_e"] = ex.numpy(), ee.numpy()
    # one layer, small, irregular edges (node 4 isolated as a target, edge (0 -> 2) twice) + the hand case
    torch.manual_seed(73)
    layer = ref_gcn.TripletGCN(dim_node=16, dim_edge=12, dim_hidden=24)
    out.update(sd_manifest(layer.state_dict(), "one/"))
    g = torch.Generator().manual_seed(173)
    for tag, ei in (("irr", torch.tensor([[0, 1, 0, 3, 4, 2, 0], [2, 0, 2, 1, 0, 3, 1]])),
                    ("hand", torch.tensor([[0, 1, 2], [2, 1, 0]]))):
        n = 5 if tag == "irr" else 3
        x = torch.randn(n, 16, generator=g).requires_grad_(True)
        e = torch.randn(ei.size(1), 12, generator=g).requires_grad_(True)
        layer.zero_grad()
        ox, oe = layer(x, e, ei)
        (ox.square().sum() + 2 * oe.sum()).backward()
        out[f"one/{tag}/x"], out[f"one/{tag}/e"], out[f"one/{tag}/ei"] = x.detach().numpy(), e.detach().numpy(), ei.numpy()
        out[f"one/{tag}/out_x"], out[f"one/{tag}/out_e"] = ox.detach().numpy(), oe.detach().numpy()
        out[f"one/{tag}/grad_x"], out[f"one/{tag}/grad_e"] = x.grad.numpy(), e.grad.numpy()
        out[f"one/{tag}/grad_w"] = layer.nn1[0].weight.grad.numpy().copy()
    save("triplet_gcn.npz", **out)


def _scan_batch(n_obj, pts_obj, pts_rel, seed):
    """A synthetic scan with the batch-dict keys of ORDataset.collate_fn (or_dataset.py:63-74)."""
    g = torch.Generator().manual_seed(seed)
    E = n_obj * (n_obj - 1)
    obj = torch.rand(n_obj, 6, pts_obj, generator=g) * 2 - 1
    rel = torch.rand(E, 7, pts_rel, generator=g) * 2 - 1
    rel[:, 6] = torch.randint(0, 3, (E, pts_rel), generator=g).float()
    onehot = torch.zeros(E, 12)
    onehot[torch.arange(E), torch.randint(0, 6, (E,), generator=g)] = 1
    onehot[torch.arange(E), 6 + torch.randint(0, 6, (E,), generator=g)] = 1
    return dict(obj_points=obj, rel_points=rel, edge_indices=_full_edges(n_obj), relation_objects_one_hot=onehot,
                gt_class=torch.randint(0, 12, (n_obj,), generator=g), gt_rels=torch.randint(0, 15, (E,), generator=g),
                full_image_features=torch.randn(6, 2048, generator=g))


def fixture_sgpn():
    """The reference's SGPNModelWrapper (SGH/model/scene_graph_prediction_model.py:31-109,134-141) for `no_gt.json` and
    `no_gt_image.json`: ORDERED state_dict keys / shapes / sums under a fixed seed (registration order = RNG consumption
    order), an eval-mode forward and a train-mode forward + loss + gradients (Dropout modules switched off: the device
    draws other random numbers) of one small synthetic scan.  Image config: the CNN is replaced by pre-computed features
    `(6, num_features)` fed through `full_image_feature_reduction` exactly as :97-99 does after the CNN."""
    import json
    _third_party_stubs()
    from scene_graph_prediction.scene_graph_helpers.model import scene_graph_prediction_model as ref_sgm
    assert ref_sgm.__file__.startswith(REF)
    names = [f"r{i}" for i in range(14)] + ["none"]
    out = {}
    batch = _scan_batch(3, 640, 768, 81)
    for k, v in batch.items():
        out["batch/" + k] = v.numpy()
    w_obj = torch.linspace(0.5, 2.0, 12)
    w_rel = torch.linspace(0.25, 3.0, 15)
    out["weights_obj"], out["weights_rel"] = w_obj.numpy(), w_rel.numpy()
    for tag, cfg_name, seed in (("no_gt", "no_gt.json", 82), ("no_gt_image", "no_gt_image.json", 83)):
        cfg = json.load(open(os.path.join(REF, "scene_graph_prediction/scene_graph_helpers/configs", cfg_name)))
        torch.manual_seed(seed)
        model = ref_sgm.SGPNModelWrapper(cfg, 12, 15, w_obj, w_rel, names)
        out.update(sd_manifest(model.state_dict(), f"{tag}/"))
        out[f"{tag}/param_names"] = np.array([k for k, _ in model.named_parameters()])
        out[f"{tag}/requires_grad"] = np.array([p.requires_grad for _, p in model.named_parameters()])
        if tag == "no_gt_image":
            # the reference calls the CNN on batch['full_image'] (:97); the features it would return are the input here
            model.full_image_model.forward = lambda img: batch["full_image_features"]
            model.freeze_image_model_batchnorm = lambda: None
            b = dict(batch, full_image=torch.zeros(6, 3, 8, 8), take_idx=4)
        else:
            b = dict(batch, take_idx=4)
        model.eval()
        with torch.no_grad():
            obj, rel, of, rf, gof, grf, _ = model(b, return_meta_data=True)
        out[f"{tag}/eval/obj_cls"], out[f"{tag}/eval/rel_cls"] = obj.numpy(), rel.numpy()
        out[f"{tag}/eval/obj_feature"], out[f"{tag}/eval/rel_feature"] = of.numpy(), rf.numpy()
        out[f"{tag}/eval/gcn_obj_feature"], out[f"{tag}/eval/gcn_rel_feature"] = gof.numpy(), grf.numpy()
        model.train()
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.eval()
        loss = model.training_step(b, 1)
        loss.backward()
        out[f"{tag}/train/loss"] = np.array([float(loss)])
        named = [(k, p) for k, p in model.named_parameters() if p.grad is not None]
        out[f"{tag}/train/grad_names"] = np.array([k for k, _ in named])
        out[f"{tag}/train/grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in named])
    save("sgpn.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sa_msg", "fp", "msg_encoder", "gf3d_backbone", "gf3d_backbone_eval", "votes_pooling", "heads", "sample_uniformly", "triplet_gcn", "sgpn"]
    for name in which:
        globals()["fixture_" + name]()
