"""Garbage collection on the training loop's schedule instead of the allocator's.

A training step here is a few hundred asynchronous launches that the host thread enqueues in 4-8 ms while the GPU needs
10-15 ms: the GPU stays busy as long as the host never stops for longer than its lead.  CPython's cyclic collector does
exactly that — a generation-2 pass over this process's heap (modules, autograd nodes, cached tables) takes 30-90 ms and
was measured as an idle GPU in about half of the 20-step bench runs (bench.py, `_NoCollectorPauses`).  `ScheduledGC`
freezes what exists after set-up (it will never be garbage), disables the automatic collector and collects every
`every` steps at a step boundary, where a pause costs the host's lead at most once per interval.  Reference counting
still frees every tensor immediately; only reference CYCLES wait for the scheduled pass.  Every `full_every`-th scheduled
pass is a FULL collection: objects that survive a young pass are promoted to the oldest generation, which a young pass never
visits — cycles that die later (an autograd context -> saved output -> grad_fn ring of a forward that was never followed by
a backward holds GPU tensors) would otherwise never be freed for the rest of training.  The frozen set-up heap is not
scanned by either, so the full pass only walks what training itself allocated."""
import gc


class ScheduledGC:
    def __init__(self, every: int = 200, full_every: int = 10):
        self.every = int(every)
        self.full_every = int(full_every)
        self.steps = 0
        self.passes = 0
        self.active = False
        self._was_enabled = True

    def __enter__(self):
        self._was_enabled = gc.isenabled()
        gc.collect()
        gc.freeze()
        gc.disable()
        self.active = True
        return self

    def step(self):
        """Call once per training step, after the step's work has been enqueued."""
        self.steps += 1
        if self.active and self.every > 0 and self.steps % self.every == 0:
            self.collect()

    def collect(self):
        """One scheduled pass (also callable at an epoch boundary): young generations, every `full_every`-th time all."""
        self.passes += 1
        if self.full_every > 0 and self.passes % self.full_every == 0:
            gc.collect()           # oldest generation too: cycles promoted by earlier young passes
        else:
            gc.collect(1)          # young generations: cycles created by the last `every` steps

    def __exit__(self, *exc):
        self.active = False
        gc.unfreeze()
        gc.collect()               # whatever the loop left behind, before the automatic collector takes over again
        if self._was_enabled:      # a caller that had the collector off keeps it off
            gc.enable()
        return False
