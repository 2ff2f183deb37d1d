"""Next-batch geometry prefetch for a scan-at-a-time training loop.

The sampling / grouping geometry of a batch (FPS chains, ball queries, 3-NN weights) involves no parameters, so the
loop can enqueue it for batch i+1 on a side HIP stream before it enqueues the optimisation step of batch i; the
latency-bound sampling kernels then co-run with the MFMA kernels of the step.  The result is
handed to the model as ``batch["geometry"]`` — identical numbers, the work is only moved.

Reference behaviour replaced: the DataLoader worker of scene_graph_prediction/main.py:54-56 prepares the next scan on
the host while the GPU trains; here the device-side part of that preparation overlaps as well.
"""
from typing import Any, Callable, Dict, Iterable, Iterator

import torch


def _record_stream(obj, stream):
    if torch.is_tensor(obj):
        obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class GeometryPrefetcher:
    """Iterates `batches` (already on the device) and yields each with ``batch["geometry"]`` filled in, while the
    geometry of the FOLLOWING batch is being computed on a side stream.

        for batch in GeometryPrefetcher(model.precompute_geometry, device_batches):
            loss = model.training_step(batch, ...)
    """

    def __init__(self, precompute: Callable[[Dict[str, Any]], Any], batches: Iterable[Dict[str, Any]], device=None):
        self.precompute = precompute
        self.batches = batches
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # high priority = a hardware queue of its own: two default-priority streams can be mapped onto ONE queue (HIP
        # assigns queues round-robin) and then never overlap (profiles/r02_stream_queue_aliasing.md)
        self.side = torch.cuda.Stream(device=self.device, priority=-1)

    def _launch(self, batch):
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)                    # the batch's tensors are ready for the side stream
        with torch.cuda.stream(self.side):
            return self.precompute(batch)

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        geo = self._launch(cur)
        while cur is not None:
            nxt = next(it, None)
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self.side)
            _record_stream(geo, main)
            out = dict(cur, geometry=geo)
            if nxt is not None:
                geo = self._launch(nxt)                # enqueued BEFORE the consumer enqueues this batch's step
            yield out
            cur = nxt
