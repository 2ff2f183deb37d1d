"""Whole training step as a replayed hipGraph.

The scene-graph step is ~450 kernel launches of which most are a few microseconds long
(72 pair clouds of 8000 points, 9 object clouds of 4000): enqueueing them costs the host more
than executing them costs the GPU.  ``GraphedTrainStep`` captures

    grads <- 0 ; loss, outputs = step_fn(batch) ; loss.backward() ; optimizer.step()

once per batch *signature* (the shapes / dtypes of the batch's tensors — nested dicts / lists of tensors
such as a prefetched ``batch["geometry"]`` included —, i.e. the number of objects and points of a scan) into a hipGraph on the capture stream and replays it afterwards;
a new signature is run eagerly once (that call is an ordinary training step and also warms
every lazily initialised handle) and captured on its next occurrence.

Gradients live in ONE flat fp32 buffer whose slices are the parameters' ``.grad`` views, so
  * the graph zeroes / accumulates them in place (static addresses across replays), and
  * data parallelism is a single all-reduce of that buffer between the backward graph and the
    optimizer graph (one process per GPU, RCCL over xGMI; the models on this path have a few MB
    of gradients, so one unbucketed collective after the backward costs less than a
    bucket-per-hook schedule would).  With world size 1 the two graphs are one.

``capture=False`` runs the same schedule eagerly (flat gradients, one all-reduce): the
debugging mode, and the only one available on a CPU/gloo job.

Reference behaviour replaced: pytorch_lightning's ``Trainer.fit`` inner loop over
``training_step`` / ``optimizer.step`` (scene_graph_prediction/main.py:58-66 of the reference),
which issues every kernel from Python each step.
"""
from typing import Any, Callable, Dict, Iterable, Optional, Tuple

import torch
import torch.distributed as dist


# Objects in a batch may carry device tensors too (e.g. the SceneBatch of a block-diagonal batch of scans).  A graph can only
# take them as inputs if it can see and re-point those tensors, so such a class declares them:
#     graph_tensor_fields : names of its tensor attributes (copied into the graph's static buffers on every call)
#     graph_static()      : hashable python state a step_fn may branch on (part of the graph's signature)
#     map_tensors(fn)     : a new object with every declared tensor replaced by fn(tensor)
def _is_graph_object(obj) -> bool:
    return hasattr(obj, "graph_tensor_fields") and hasattr(obj, "map_tensors") and hasattr(obj, "graph_static")


def _tensor_leaves(obj, path=()):
    """(path, tensor) for every tensor in a nest of dict / list / tuple (e.g. batch["geometry"]), in a fixed order."""
    if torch.is_tensor(obj):
        yield path, obj
    elif _is_graph_object(obj):
        for name in obj.graph_tensor_fields:
            yield path + (name,), getattr(obj, name)
    elif isinstance(obj, dict):
        for k in sorted(obj, key=str):
            yield from _tensor_leaves(obj[k], path + (k,))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _tensor_leaves(v, path + (i,))


def _map_tensors(obj, fn):
    """The same nest with every tensor replaced by fn(tensor)."""
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    if _is_graph_object(obj):
        return obj.map_tensors(fn)
    return obj


def _scalar_leaves(obj, path=()):
    """(path, value) for every int / float / bool below the top level of a nest (e.g. geometry["obj"][0]["n_src"])."""
    if isinstance(obj, dict):
        for k in sorted(obj, key=str):
            yield from _scalar_leaves(obj[k], path + (k,))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _scalar_leaves(v, path + (i,))
    elif isinstance(obj, (bool, int, float)) and len(path) > 1:
        yield path, obj
    elif _is_graph_object(obj):
        yield path + ("graph_static",), obj.graph_static()


def batch_signature(batch: Dict[str, Any], static_keys: Iterable[str] = ()) -> Tuple:
    """What a captured graph is keyed on: shape and dtype of every tensor of the nest, every NESTED python scalar
    (values a step_fn may branch on, e.g. the point count a prefetched geometry was computed for) and the top-level
    entries named in `static_keys`.  Other top-level non-tensor entries (scan id, take index, ...) are per-sample
    metadata: they ride along and must not influence the captured kernels."""
    sig = [(path, tuple(t.shape), str(t.dtype)) for path, t in _tensor_leaves(batch)]
    sig += [(path, type(v).__name__, v) for path, v in _scalar_leaves(batch)]
    sig += [((k,), "static", batch.get(k)) for k in sorted(static_keys)]
    return tuple(sig)


class FlatGrads:
    """One contiguous gradient buffer; every parameter's ``.grad`` is a view of it."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGrads needs all parameters on one device with one dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / world)


class _Captured:
    __slots__ = ("static", "fwd_bwd", "opt", "loss", "outputs")


class GraphedTrainStep:
    """step_fn(batch) -> (loss, outputs) where outputs is a tensor / tuple of tensors / None.

    __call__(batch) performs one optimisation step and returns (loss, outputs); with capture
    on, both are static tensors that the next call with the same signature overwrites.

    What a replay freezes, and what is done about it: (1) python values step_fn branches on — nested scalars and the
    `static_keys` entries are part of the signature (see batch_signature), so a different value captures a different
    graph; (2) the optimizer's python-float hyper-parameters (lr under a scheduler, weight_decay, betas, eps) are baked
    into the captured update — they are snapshotted at capture and every call compares them: a change drops the
    captured graphs and re-captures (pass lr as a device tensor to change it without re-capturing, which a capturable
    torch optimizer supports)."""

    def __init__(self, step_fn: Callable[[Dict[str, Any]], Tuple[torch.Tensor, Any]],
                 params: Iterable[torch.nn.Parameter], optimizer: torch.optim.Optimizer,
                 capture: bool = True, process_group=None, max_graphs: int = 32, static_keys: Iterable[str] = ()):
        self.step_fn, self.optimizer = step_fn, optimizer
        self.static_keys = tuple(static_keys)
        self._hyper = None
        self.grads = FlatGrads(params)
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.capture = bool(capture)
        self.max_graphs = int(max_graphs)
        self._seen: Dict[Tuple, int] = {}
        self._graphs: Dict[Tuple, _Captured] = {}
        if self.capture:
            if not self.grads.flat.is_cuda:
                raise RuntimeError("GraphedTrainStep(capture=True) needs the model on a GPU; use capture=False on CPU")
            for g in optimizer.param_groups:
                if "capturable" in g and not g["capturable"]:
                    raise ValueError("the optimizer must be built with capturable=True to be replayed in a graph")
            self._stream = torch.cuda.Stream(device=self.grads.flat.device)

    # ------------------------------------------------------------------ eager schedule
    def _eager(self, batch):
        self.grads.zero_()
        loss, outputs = self.step_fn(batch)
        loss.backward()
        if self.distributed:
            self.grads.all_reduce_mean(self.group)
        self.optimizer.step()
        return loss.detach(), outputs

    # ------------------------------------------------------------------ capture
    def _capture(self, batch, sig) -> _Captured:
        for k, v in batch.items():
            if not (torch.is_tensor(v) or v is None or isinstance(v, (str, int, float, bool, dict, list, tuple))
                    or _is_graph_object(v)):
                raise ValueError(f"batch[{k!r}] ({type(v).__name__}) may hold device tensors whose addresses a graph "
                                 "would freeze; pass plain tensors (or dicts / lists of them) and let step_fn derive such objects")
        c = _Captured()
        c.static = _map_tensors(batch, lambda t: t.clone())       # nested tensors too (e.g. prefetched geometry)
        c.fwd_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(c.fwd_bwd, stream=self._stream):
            self.grads.zero_()
            loss, outputs = self.step_fn(c.static)
            loss.backward()
            if not self.distributed:
                self.optimizer.step()
        c.loss = loss.detach()
        c.outputs = outputs
        c.opt = None
        if self.distributed:
            c.opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(c.opt, stream=self._stream):
                self.optimizer.step()
        return c

    def __call__(self, batch: Dict[str, Any]):
        if not self.capture:
            return self._eager(batch)
        hyper = self._hyper_params()
        if self._graphs and hyper != self._hyper:
            self._graphs.clear()                                  # a scheduler / the user changed a baked-in hyper-parameter
        sig = batch_signature(batch, self.static_keys)
        c = self._graphs.get(sig)
        if c is None:
            n = self._seen.get(sig, 0)
            self._seen[sig] = n + 1
            if n == 0 or len(self._graphs) >= self.max_graphs:
                # first occurrence: an ordinary step on the capture stream (lazy handles, optimizer state)
                self._stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._stream):
                    out = self._eager(batch)
                torch.cuda.current_stream().wait_stream(self._stream)
                return out
            c = self._graphs[sig] = self._capture(batch, sig)
            self._hyper = hyper
        for (_, dst), (_, src) in zip(_tensor_leaves(c.static), _tensor_leaves(batch)):
            dst.copy_(src, non_blocking=True)
        for k, v in batch.items():
            if not isinstance(v, (torch.Tensor, dict, list, tuple)) and not _is_graph_object(v):
                c.static[k] = v                                   # plain metadata (scan id, ...) just rides along
        c.fwd_bwd.replay()
        if c.opt is not None:
            self.grads.all_reduce_mean(self.group)
            c.opt.replay()
        return c.loss, c.outputs

    def _hyper_params(self):
        """Python-number hyper-parameters of every param group (device tensors are read by the graph at replay time)."""
        out = []
        for g in self.optimizer.param_groups:
            for k in sorted(g):
                v = g[k]
                if k == "params" or torch.is_tensor(v):
                    continue
                if isinstance(v, (bool, int, float)) or (isinstance(v, tuple) and all(isinstance(x, (int, float)) for x in v)):
                    out.append((k, v))
        return tuple(out)

    @property
    def num_graphs(self) -> int:
        return len(self._graphs)
