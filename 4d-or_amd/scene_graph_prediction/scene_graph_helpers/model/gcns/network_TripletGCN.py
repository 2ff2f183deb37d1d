"""TripletGCN message passing on MI355X.

API mirror of SGH/model/gcns/network_TripletGCN.py (SGH =
scene_graph_prediction/scene_graph_helpers): ``build_mlp`` (:11-27),
``TripletGCN(dim_node, dim_edge, dim_hidden, aggr='add', use_bn=True)`` (:30-58)
and ``TripletGCNModel(num_layers, **kwargs)`` (:61-80) with identical parameter
names (``gconvs.{l}.nn1.{0,1,3,4}`` / ``nn2.{0,1,3}``).

The reference inherits torch_geometric 2.0.2's ``MessagePassing`` and calls
torch_scatter 2.0.9's ``scatter`` — un-vendored pip dependencies (README.md:87).
Their semantics on this call path are restated here without either package:

* flow ``source_to_target``: ``x_j = x[edge_index[0]]`` (subject / source),
  ``x_i = x[edge_index[1]]`` (object / target)  (PyG ``__lift__``);
* ``message`` (:45-52): ``nn1(cat[x_i, e, x_j])`` -> split ``hidden | edge | hidden``;
  node message = first + last, new edge feature = middle;
* ``aggregate`` (:54-58): sum of node messages over ``edge_index[1]`` into ``N`` rows;
* ``update``: identity; then ``nn2`` on the aggregated rows (:42-43).

Gather and scatter run in libpn2_hip.so.  The first Linear of ``nn1`` is applied BEFORE the lift:
``W [x_i | e | x_j] = Wa x_i + Wb e + Wc x_j``, so the node part is one (N, dn) x (dn, 2H) GEMM instead of an
(E, 2 dn) one (E/N ~ 8 times fewer FLOPs) and ``pn2_gather2_add_rows`` lifts the PRODUCTS onto the edges — the
(E, 2 dn + de) concatenation is never built.  That form is taken from ``LIFT_MIN_EDGES`` edges on (below, the step is launch-bound and the
literal concat form has fewer launches).  In the lifted form the split + aggregate (:50-58) is one ``pn2_segment_sum2_rows`` (first +
last block of nn1's output summed while aggregating); the CSR sums add in edge order, i.e. like a sequential CPU
``scatter_add_``.  The literal concat path (``pn2_gather_rows`` into the concatenation buffer) also serves dimensions
that are not multiples of 4.
BatchNorm1d layers use batch statistics in train AND eval
(``track_running_stats=False``, :20), like the reference.

Batched scans (MI355X-first; the reference feeds one scan per step, main.py:54-56): several scans are concatenated
block-diagonally — node rows, edge rows and ``edge_index`` offset per scan — and described by a ``SceneBatch``.  The
edge gathers / scatters need nothing else (no edge crosses scans); the BatchNorm1d layers must keep normalising with
the statistics of EACH scan's rows, which ``pn2_segment_bn_rows`` does in one launch (+ReLU), so a batch of S scans gives
exactly the S single-scan results.
"""
import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from pointnet2_ops import _ext
from scene_graph_prediction.scene_graph_helpers.model.pointnets.networks_base import BaseNetwork


def build_mlp(dim_list, activation="relu", do_bn=False, dropout=0, on_last=False):
    layers = []
    last = len(dim_list) - 2
    for i, (d_in, d_out) in enumerate(zip(dim_list[:-1], dim_list[1:])):
        layers.append(torch.nn.Linear(d_in, d_out))
        if i != last or on_last:
            if do_bn:
                layers.append(torch.nn.BatchNorm1d(d_out, track_running_stats=False))
            if activation == "relu":
                layers.append(torch.nn.ReLU())
            elif activation == "leakyrelu":
                layers.append(torch.nn.LeakyReLU())
        if dropout > 0:
            layers.append(torch.nn.Dropout(p=dropout))
    return torch.nn.Sequential(*layers)


# Edge count from which nn1's first Linear is applied to the nodes and the products lifted (see the module docstring).
# Measured on MI355X, 2 layers 256/256/512, forward+backward (tools/gcn_time.py): the step is launch-bound (~2 ms) up to
# 64 scans (4 608 edges), where the lifted form's extra small launches cost +0.3 ms; from 256 scans (18 432 edges) it is
# 11-17 % faster (4.35 -> 3.87 ms; 1 024 scans 14.0 -> 11.9 ms; 4 096 scans 50.6 -> 42.1 ms).
LIFT_MIN_EDGES = 8192


class EdgeCSR:
    """Stable sort of the edge targets, computed once per graph and shared by all layers.

    Built without a device->host round trip: the counts come from a scatter-add (``torch.bincount``
    would read the maximum back) and the range check of a CUDA ``edge_index`` is an asynchronous
    device assertion, so a training step that only holds the edges on the GPU never drains the
    stream.  A CPU ``edge_index`` is validated eagerly (RuntimeError)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError("edge_index must have shape (2, E)")
        self.src = edge_index[0].contiguous()
        self.dst = edge_index[1].contiguous()
        self.num_nodes = n = int(num_nodes)
        if self.dst.numel():
            ok = (edge_index >= 0).all() & (edge_index < n).all()
            if edge_index.is_cuda:
                torch._assert_async(ok)
            elif not bool(ok):
                raise RuntimeError("edge_index out of range")
        self.order, self.rowptr = self._csr(self.dst, n)
        # CSR by source, for the backward of the x_j gather
        self.order_src, self.rowptr_src = self._csr(self.src, n)

    @staticmethod
    def _csr(key, n):
        order = torch.sort(key, stable=True).indices.contiguous()
        counts = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
        counts.scatter_add_(0, key + 1, torch.ones_like(key))
        return order, torch.cumsum(counts, 0)

    def to(self, device):
        for k in ("src", "dst", "order", "rowptr", "order_src", "rowptr_src"):
            setattr(self, k, getattr(self, k).to(device, non_blocking=True))
        return self


class SceneBatch:
    """Row ranges of the scans of a block-diagonal batch: ``node_ptr`` / ``edge_ptr`` (S+1) int64 offsets into the node
    and edge rows, plus the per-row scan ids the loss needs.  Built on the host by the collate step
    (dataset/synthetic.py::collate_scans); moving it to the device is asynchronous."""

    def __init__(self, node_ptr: torch.Tensor, edge_ptr: torch.Tensor):
        self.node_ptr = node_ptr.to(torch.int64).contiguous()
        self.edge_ptr = edge_ptr.to(torch.int64).contiguous()
        if self.node_ptr.numel() != self.edge_ptr.numel() or self.node_ptr.numel() < 2:
            raise RuntimeError("SceneBatch: node_ptr and edge_ptr must both have S + 1 entries")
        self.num_scenes = self.node_ptr.numel() - 1
        counts_n = self.node_ptr[1:] - self.node_ptr[:-1]
        counts_e = self.edge_ptr[1:] - self.edge_ptr[:-1]
        # host copies (the collate step builds the batch on the CPU: no device read-back later)
        self.nodes_per_scene, self.edges_per_scene = counts_n.tolist(), counts_e.tolist()
        # every BatchNorm of the model normalises over ONE scan's rows: torch's batch_norm refuses a single row in
        # training mode ("Expected more than 1 value per channel"), and a scan needs >= 2 objects to have an edge at all
        # (EXT or_dataset.py:130-133 builds all ordered pairs) — refuse here instead of normalising with var = 0
        if min(self.nodes_per_scene) < 2 or min(self.edges_per_scene) < 2:
            raise RuntimeError("SceneBatch: every scan needs at least 2 nodes and 2 edges (per-scan BatchNorm statistics)")
        ids = torch.arange(self.num_scenes, device=self.node_ptr.device)
        self.node_scene = torch.repeat_interleave(ids, counts_n)
        self.edge_scene = torch.repeat_interleave(ids, counts_e)

    # graph-input protocol of runtime.GraphedTrainStep: the tensors a captured step reads, the python state it branches on
    graph_tensor_fields = ("node_ptr", "edge_ptr", "node_scene", "edge_scene")

    def graph_static(self):
        return (tuple(self.nodes_per_scene), tuple(self.edges_per_scene))

    def map_tensors(self, fn):
        other = object.__new__(SceneBatch)
        other.num_scenes = self.num_scenes
        other.nodes_per_scene, other.edges_per_scene = self.nodes_per_scene, self.edges_per_scene
        for k in self.graph_tensor_fields:
            setattr(other, k, fn(getattr(self, k)))
        return other

    def to(self, device):
        return self.map_tensors(lambda t: t.to(device, non_blocking=True))


class OneScan:
    """The (node_ptr, edge_ptr) row offsets of ONE scan on the device: lets a single-scan step take the fused per-scan
    BatchNorm (+ReLU) kernel — one launch forward and one backward per BatchNorm1d instead of torch's three + ReLU resp. two
    + the ReLU mask — with the very arithmetic a batch of scans gets.  Cached per (device, rows): built from host numbers
    (a copy, not capturable), so the first, eager step of a signature creates it."""
    _CACHE = {}
    _MAX = 4096

    def __init__(self, device, n_nodes, n_edges):
        self.node_ptr = torch.tensor([0, int(n_nodes)], dtype=torch.int64, device=device)
        self.edge_ptr = torch.tensor([0, int(n_edges)], dtype=torch.int64, device=device)
        self.num_scenes = 1

    @classmethod
    def get(cls, device, n_nodes, n_edges):
        key = (device, int(n_nodes), int(n_edges))
        hit = cls._CACHE.get(key)
        if hit is None:
            if len(cls._CACHE) >= cls._MAX:
                cls._CACHE.clear()
            hit = cls._CACHE[key] = cls(device, n_nodes, n_edges)
        return hit


class _SegmentBNReLU(Function):
    """BatchNorm1d(track_running_stats=False) [+ ReLU] with per-scan statistics (one launch for all scans)."""

    @staticmethod
    def forward(ctx, x, ptr, gamma, beta, eps, relu):
        x = x.contiguous()
        y, mean, rstd = _ext.segment_bn_rows(x, ptr, gamma, beta, eps, relu)
        ctx.save_for_backward(x, ptr, gamma, beta, mean, rstd)
        ctx.relu, ctx.eps = relu, eps
        return y

    @staticmethod
    def backward(ctx, g):
        x, ptr, gamma, beta, mean, rstd = ctx.saved_tensors
        gx, dgamma, dbeta = _ext.segment_bn_rows_grad(g.contiguous(), x, ptr, gamma, beta, mean, rstd, ctx.relu, eps=ctx.eps)
        return gx, None, dgamma, dbeta, None, None


class SegmentBatchNorm(Function):
    """Per-scan BatchNorm1d returning the per-scan (mean, rstd) as well — for layers that keep running statistics (the
    classification heads, pointnets/network_PointNet.py::scan_batch_norm)."""

    @staticmethod
    def forward(ctx, x, ptr, gamma, beta, eps):
        x = x.contiguous()
        y, mean, rstd = _ext.segment_bn_rows(x, ptr, gamma, beta, eps, False)
        ctx.save_for_backward(x, ptr, gamma, beta, mean, rstd)
        ctx.eps = eps
        ctx.mark_non_differentiable(mean, rstd)
        return y, mean, rstd

    @staticmethod
    def backward(ctx, g, _gm, _gr):
        x, ptr, gamma, beta, mean, rstd = ctx.saved_tensors
        gx, dgamma, dbeta = _ext.segment_bn_rows_grad(g.contiguous(), x, ptr, gamma, beta, mean, rstd, False, eps=ctx.eps)
        return gx, None, dgamma, dbeta, None


def mlp_per_scene(mlp: torch.nn.Sequential, x: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    """Run a `build_mlp` stack on block-diagonally batched rows: Linear layers see all rows at once (rocBLAS), every
    BatchNorm1d (+ the ReLU behind it) normalises each scan's rows [ptr[s], ptr[s+1]) separately."""
    layers = list(mlp)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, torch.nn.BatchNorm1d):
            if layer.track_running_stats or not layer.affine:
                raise NotImplementedError("per-scene BatchNorm1d: build_mlp layers only (affine, no running statistics)")
            relu = i + 1 < len(layers) and isinstance(layers[i + 1], torch.nn.ReLU)
            x = _SegmentBNReLU.apply(x, ptr, layer.weight, layer.bias, layer.eps, relu)
            i += 2 if relu else 1
        else:
            x = layer(x)
            i += 1
    return x


class _TripletConcat(Function):
    """cat[x[dst], e, x[src]] -> (E, 2*dn + de), gathers written in place."""

    @staticmethod
    def forward(ctx, x, e, csr):
        E, dn, de = e.size(0), x.size(1), e.size(1)
        buf = torch.empty(E, 2 * dn + de, dtype=torch.float32, device=x.device)
        _ext.gather_rows(x.contiguous(), csr.dst, out=buf, col0=0, check=False)
        buf[:, dn:dn + de] = e
        _ext.gather_rows(x.contiguous(), csr.src, out=buf, col0=dn + de, check=False)
        ctx.csr, ctx.dn, ctx.de = csr, dn, de
        return buf

    @staticmethod
    def backward(ctx, g):
        csr, dn, de = ctx.csr, ctx.dn, ctx.de
        g = g.contiguous()
        gx = _ext.segment_sum_rows(g, csr.order, csr.rowptr, csr.num_nodes, h=dn, col0=0)
        gx = gx + _ext.segment_sum_rows(g, csr.order_src, csr.rowptr_src, csr.num_nodes, h=dn, col0=dn + de)
        return gx, g[:, dn:dn + de], None


class _TripletLinear(Function):
    """``Linear(cat[x[dst], e, x[src]])`` (nn1[0], :46-47) with the node part of the product computed per NODE."""

    @staticmethod
    def forward(ctx, x, e, weight, bias, csr):
        dn, de, H = x.size(1), e.size(1), weight.size(0)
        wn = torch.cat([weight[:, :dn], weight[:, dn + de:]], 0)          # (2H, dn): [Wa ; Wc]
        p = x @ wn.t()                                                     # (N, 2H) = [x Wa^T | x Wc^T]
        q = torch.addmm(bias, e, weight[:, dn:dn + de].t())                # (E, H)
        _ext.gather2_add_rows(q, p, csr.dst, csr.src, 0, H)
        ctx.save_for_backward(x, e, weight)
        ctx.csr = csr
        return q

    @staticmethod
    def backward(ctx, g):
        x, e, weight = ctx.saved_tensors
        csr = ctx.csr
        dn, de, H = x.size(1), e.size(1), weight.size(0)
        g = g.contiguous()
        gp = torch.cat([_ext.segment_sum_rows(g, csr.order, csr.rowptr, csr.num_nodes),
                        _ext.segment_sum_rows(g, csr.order_src, csr.rowptr_src, csr.num_nodes)], 1)   # (N, 2H)
        wn = torch.cat([weight[:, :dn], weight[:, dn + de:]], 0)
        gx = gp @ wn
        ge = g @ weight[:, dn:dn + de]
        gwn = gp.t() @ x                                                   # (2H, dn)
        gw = torch.cat([gwn[:H], g.t() @ e, gwn[H:]], 1)
        return gx, ge, gw, g.sum(0), None


class _SplitAggregate(Function):
    """h (E, 2 dh + de) -> (sum over edge_index[1] of h[:, :dh] + h[:, dh+de:], h[:, dh:dh+de])  (:50-58)."""

    @staticmethod
    def forward(ctx, h, csr, dh, de):
        h = h.contiguous()
        ctx.csr, ctx.dh, ctx.de = csr, dh, de
        node = _ext.segment_sum2_rows(h, csr.order, csr.rowptr, csr.num_nodes, dh, 0, dh + de)
        return node, h[:, dh:dh + de].contiguous()

    @staticmethod
    def backward(ctx, g_node, g_edge):
        csr, dh, de = ctx.csr, ctx.dh, ctx.de
        gh = torch.empty(csr.dst.numel(), 2 * dh + de, dtype=torch.float32, device=g_node.device)
        g_node = g_node.contiguous()
        _ext.gather_rows(g_node, csr.dst, out=gh, col0=0, check=False)
        gh[:, dh:dh + de] = g_edge
        _ext.gather_rows(g_node, csr.dst, out=gh, col0=dh + de, check=False)
        return gh, None, None, None


class _AggregateAdd(Function):
    """scatter(msg, index=edge_index[1], dim=-2, dim_size=N, reduce='add')."""

    @staticmethod
    def forward(ctx, msg, csr):
        ctx.csr = csr
        return _ext.segment_sum_rows(msg.contiguous(), csr.order, csr.rowptr, csr.num_nodes)

    @staticmethod
    def backward(ctx, g):
        return _ext.gather_rows(g.contiguous(), ctx.csr.dst, check=False), None


# The whole layer as 6 launches forward / 12 backward (csrc/gcn_fused.hip) instead of ~60 / ~180 through torch + the row
# kernels: scans of <= 128 edges and nodes (the dataset's scans have at most 110 / 11), dimensions multiples of 32.
# PN2_GCN_FUSED=0 restores the unfused path (A/B); results agree within fp32 summation order.
FUSED_LAYER = os.environ.get("PN2_GCN_FUSED") != "0"
# ... up to this many scans per batch.  Measured on MI355X (tools/gcn_time.py, 2 layers, forward + backward, fused vs the
# better unfused form, two boxes): 1 scan 0.55-0.78 vs 1.8-2.1 ms, 8 scans 0.49-0.51 vs 1.6-2.0, 16 scans 0.60 vs 1.5,
# 32 scans 0.96-1.0 vs 1.5, 64 scans 1.7-1.8 vs 1.6-2.2, 128 scans 3.4 vs 2.3 — the per-scan workgroups of the forward /
# input-gradient kernels re-read the weights once per scan and column tile (L2 traffic grows with the scan count), while the
# unfused path's library GEMMs see ONE tall matrix.
# Round 6: 64 (was 32) — BASELINE configs[4] names 64 scenes and the two routes measure the same there (1.7-1.8 vs 1.6-2.2 ms
# above, profiles/r06_gcn_time.jsonl), so the batch stays on the hand-written kernels instead of the library GEMMs; beyond
# 64 scans the unfused route is faster (128 scans: 2.3 vs 3.4 ms) and keeps being chosen.
FUSED_MAX_SCANS = 64
# The layer's launches issued from ONE C call each way (pn2_gcn_layer_forward / _backward) instead of block by block from
# python; PN2_GCN_LAYER_CALL=0 restores the block-by-block sequence (same kernels, same results; A/B of the host cost).
LAYER_CALL = os.environ.get("PN2_GCN_LAYER_CALL") != "0"


class _FusedTripletLayer(Function):
    """TripletGCN.forward (network_TripletGCN.py:40-58) as one autograd node over the fused per-scan kernels:
    nn1 on the virtual cat[x_i, e, x_j] -> split -> aggregate -> nn2, BatchNorm statistics per scan."""

    @staticmethod
    def forward(ctx, x, e, csr, node_ptr, edge_ptr, S, relu_out, eps, *params):
        params = tuple(p_.contiguous() for p_ in params)          # (the C side takes plain row-major pointers)
        W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, W4, b4 = params
        x, e = x.contiguous(), e.contiguous()
        if LAYER_CALL:
            # the whole layer from ONE C call each way (pn2_gcn_layer_forward / _backward): a scan-sized layer is bound by
            # the host thread's python -> C round trips, not by its kernels
            out, e_out, saved = _ext.gcn_layer_forward(x, e, csr.dst, csr.src, csr.order, csr.rowptr, node_ptr, edge_ptr, S,
                                                       relu_out, eps, params)
            ctx.csr, ctx.S, ctx.relu_out, ctx.layer_call = csr, S, relu_out, True
            ctx.save_for_backward(x, e, node_ptr, edge_ptr, saved, out, *params)
            return out, e_out
        ctx.layer_call = False
        trip = (x, e, csr.dst, csr.src)
        dh, de = W3.size(1), e.size(1)
        h1, h1p, m1, r1 = _ext.gcn_linear(W1, b1, edge_ptr, S, triplet=trip, bn=(g1, be1, eps[0]), relu=True)
        h2, h2p, m2, r2 = _ext.gcn_linear(W2, b2, edge_ptr, S, A=h1, bn=(g2, be2, eps[1]), relu=True)
        # node message = first + last block, summed over edge_index[1] in edge order (:50, 54-58); edge feature = the middle
        agg = _ext.segment_sum2_rows(h2, csr.order, csr.rowptr, csr.num_nodes, dh, 0, dh + de)
        e_out = _ext.gcn_edge_slice(h2, dh, de, relu_out)
        t, tp, m3, r3 = _ext.gcn_linear(W3, b3, node_ptr, S, A=agg, bn=(g3, be3, eps[2]), relu=True)
        out = _ext.gcn_linear(W4, b4, node_ptr, S, A=t, relu=relu_out)
        ctx.csr, ctx.S, ctx.relu_out = csr, S, relu_out
        ctx.save_for_backward(x, e, node_ptr, edge_ptr, h1, h1p, m1, r1, h2p, m2, r2, agg, t, tp, m3, r3, out, e_out, *params)
        return out, e_out

    @staticmethod
    def backward(ctx, g_out, g_e):
        if ctx.layer_call:
            x, e, node_ptr, edge_ptr, saved, out, *params = ctx.saved_tensors
            csr = ctx.csr
            gx, ge, grads = _ext.gcn_layer_backward(g_out.contiguous(), g_e.contiguous(), x, e, csr.dst, csr.src, csr.order,
                                                    csr.rowptr, node_ptr, edge_ptr, ctx.S, ctx.relu_out, params, saved, out)
            return (gx, ge, None, None, None, None, None, None, *grads)
        (x, e, node_ptr, edge_ptr, h1, h1p, m1, r1, h2p, m2, r2, agg, t, tp, m3, r3, out, e_out,
         W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, W4, b4) = ctx.saved_tensors
        csr, S = ctx.csr, ctx.S
        dn, de, dh = x.size(1), e.size(1), W3.size(1)
        g_out, g_e = g_out.contiguous(), g_e.contiguous()
        # (a ReLU on e_out needs no mask of its own: e_out is a slice of h2 = ReLU(..), block 2's mask zeroes the same entries)
        f32 = torch.float32
        shapes = [(tuple(p_.shape), f32) for p_ in (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, W4, b4)]
        (dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2, dW3, db3, dg3, dbe3, dW4, db4, gx) = _ext.zero_arena(
            x.device, shapes + [(tuple(x.shape), f32)])
        gz4 = _ext.gcn_linear_grad_w(W4.shape, node_ptr, S, dW4, db4, G=g_out, relu=ctx.relu_out, ypre=out, A=t)
        g_t = _ext.gcn_linear_grad_x(gz4, W4, node_ptr, S)
        gz3 = _ext.gcn_linear_grad_w(W3.shape, node_ptr, S, dW3, db3, G=g_t, bn=(tp, m3, r3, g3, be3), relu=True, A=agg,
                                     dgamma=dg3, dbeta=dbe3)
        g_agg = _ext.gcn_linear_grad_x(gz3, W3, node_ptr, S)
        # the adjoint of split + aggregate ([g_agg[dst] | g_e | g_agg[dst]]) is read in place by the kernel
        gz2 = _ext.gcn_linear_grad_w(W2.shape, edge_ptr, S, dW2, db2, adjoint=(g_agg, g_e, csr.dst, dh, de),
                                     bn=(h2p, m2, r2, g2, be2), relu=True, A=h1, dgamma=dg2, dbeta=dbe2)
        g_h1 = _ext.gcn_linear_grad_x(gz2, W2, edge_ptr, S)
        gz1 = _ext.gcn_linear_grad_w(W1.shape, edge_ptr, S, dW1, db1, G=g_h1, bn=(h1p, m1, r1, g1, be1), relu=True,
                                     triplet=(x, e, csr.dst, csr.src), dgamma=dg1, dbeta=dbe1)
        ge = torch.empty_like(e)
        _ext.gcn_linear_grad_x(gz1, W1, edge_ptr, S, scatter=(gx, ge, csr.dst, csr.src, dn, de))
        return (gx, ge, None, None, None, None, None, None,
                dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2, dW3, db3, dg3, dbe3, dW4, db4)


class TripletGCN(torch.nn.Module):
    def __init__(self, dim_node, dim_edge, dim_hidden, aggr="add", use_bn=True):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("the reference only ever aggregates with 'add' (GCN_AGGR is ignored)")
        self.aggr = aggr
        self.dim_node, self.dim_edge, self.dim_hidden = dim_node, dim_edge, dim_hidden
        self.nn1 = build_mlp([dim_node * 2 + dim_edge, dim_hidden, dim_hidden * 2 + dim_edge],
                             do_bn=use_bn, on_last=True)
        self.nn2 = build_mlp([dim_hidden, dim_hidden, dim_node], do_bn=use_bn)
        self.use_bn = use_bn

    def _fused_ok(self, x, edge_feature, scenes):
        if not (FUSED_LAYER and self.use_bn and x.is_cuda and x.dtype == torch.float32 and edge_feature.dtype == torch.float32
                and len(self.nn1) == 6 and len(self.nn2) == 4 and getattr(_ext, "gcn_linear", None)):
            return False
        if scenes is not None:
            if scenes.num_scenes > FUSED_MAX_SCANS:
                return False
            rows = max(max(scenes.edges_per_scene), max(scenes.nodes_per_scene))
        else:
            rows = max(edge_feature.size(0), x.size(0))
        return rows >= 2 and _ext.gcn_fused_supported(self.dim_node, self.dim_edge, self.dim_hidden, rows)

    def forward(self, x, edge_feature, edge_index, csr: Optional[EdgeCSR] = None, scenes: Optional[SceneBatch] = None,
                relu_out: bool = False):
        """`relu_out`: apply the ReLU TripletGCNModel puts on both results between layers (:76-78) inside the layer."""
        csr = csr if csr is not None else EdgeCSR(edge_index, x.size(0))
        if self._fused_ok(x, edge_feature, scenes):
            ptrs = scenes if scenes is not None else OneScan.get(x.device, x.size(0), edge_feature.size(0))
            n1, n2 = self.nn1, self.nn2
            params = (n1[0].weight, n1[0].bias, n1[1].weight, n1[1].bias, n1[3].weight, n1[3].bias, n1[4].weight, n1[4].bias,
                      n2[0].weight, n2[0].bias, n2[1].weight, n2[1].bias, n2[3].weight, n2[3].bias)
            return _FusedTripletLayer.apply(x, edge_feature, csr, ptrs.node_ptr, ptrs.edge_ptr, ptrs.num_scenes, bool(relu_out),
                                            (n1[1].eps, n1[4].eps, n2[1].eps), *params)
        gcn_x, gcn_e = self.propagate(csr, x=x, edge_feature=edge_feature, scenes=scenes)
        if scenes is not None:
            gcn_x = mlp_per_scene(self.nn2, gcn_x, scenes.node_ptr)
        else:
            gcn_x = self.nn2(gcn_x)
        if relu_out:
            gcn_x, gcn_e = torch.nn.functional.relu(gcn_x), torch.nn.functional.relu(gcn_e)
        return gcn_x, gcn_e

    def _lift_after_linear(self, n_edges):
        first = self.nn1[0]
        return (n_edges >= LIFT_MIN_EDGES and self.dim_node % 4 == 0 and self.dim_edge % 4 == 0 and self.dim_hidden % 4 == 0
                and isinstance(first, torch.nn.Linear) and first.bias is not None)

    def propagate(self, csr, x, edge_feature, scenes=None):
        if self._lift_after_linear(edge_feature.size(0)):
            # Linear on the nodes, lift the products, rest of nn1 on the edges, split + aggregate in one CSR sum
            h = _TripletLinear.apply(x.contiguous(), edge_feature.contiguous(), self.nn1[0].weight, self.nn1[0].bias, csr)
            rest = self.nn1[1:]
            h = mlp_per_scene(rest, h, scenes.edge_ptr) if scenes is not None else rest(h)
            return _SplitAggregate.apply(h, csr, self.dim_hidden, self.dim_edge)
        node_msg, new_e = self.message(csr, x, edge_feature, scenes)
        return self.aggregate(node_msg, csr), new_e

    def message(self, csr, x, edge_feature, scenes=None):
        cat = _TripletConcat.apply(x, edge_feature, csr)
        h = mlp_per_scene(self.nn1, cat, scenes.edge_ptr) if scenes is not None else self.nn1(cat)
        dh, de = self.dim_hidden, self.dim_edge
        return h[:, :dh] + h[:, dh + de:], h[:, dh:dh + de]

    def aggregate(self, node_msg, csr):
        return _AggregateAdd.apply(node_msg, csr)


class TripletGCNModel(BaseNetwork):
    """`num_layers` TripletGCN layers; ReLU on node and edge features between layers only."""

    def __init__(self, num_layers, **kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.gconvs = torch.nn.ModuleList(TripletGCN(**kwargs) for _ in range(num_layers))

    def forward(self, node_feature, edge_feature, edges_indices, csr: Optional[EdgeCSR] = None,
                scenes: Optional[SceneBatch] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """`csr` (optional) is an EdgeCSR of `edges_indices` prepared ahead of time (e.g. by the data
        loader, on the host); without it the CSR is built here, sync-free, once per call.
        `scenes` (optional): the rows are S scans batched block-diagonally; BatchNorm statistics stay per scan."""
        if csr is None:
            csr = EdgeCSR(edges_indices, node_feature.size(0))
        for i, gconv in enumerate(self.gconvs):
            # ReLU on both results between layers only (:76-78), applied inside the layer (fused into its last kernels)
            node_feature, edge_feature = gconv(node_feature, edge_feature, edges_indices, csr, scenes,
                                               relu_out=i < self.num_layers - 1)
        return node_feature, edge_feature
