"""Garbage collection on the training loop's schedule instead of the allocator's.

A training step here is a few hundred asynchronous launches that the host thread enqueues in 4-8 ms while the GPU needs
10-15 ms: the GPU stays busy as long as the host never stops for longer than its lead.  CPython's cyclic collector does
exactly that — a generation-2 pass over this process's heap (modules, autograd nodes, cached tables) takes 30-90 ms and
was measured as an idle GPU in about half of the 20-step bench runs (bench.py, `_NoCollectorPauses`).  `ScheduledGC`
freezes what exists after set-up (it will never be garbage), disables the automatic collector and collects every
`every` steps at a step boundary, where a pause costs the host's lead at most once per interval.  Reference counting
still frees every tensor immediately; only reference CYCLES wait for the scheduled pass."""
import gc


class ScheduledGC:
    def __init__(self, every: int = 200):
        self.every = int(every)
        self.steps = 0
        self.active = False

    def __enter__(self):
        gc.collect()
        gc.freeze()
        gc.disable()
        self.active = True
        return self

    def step(self):
        """Call once per training step, after the step's work has been enqueued."""
        self.steps += 1
        if self.active and self.every > 0 and self.steps % self.every == 0:
            gc.collect(1)          # young generations: cycles created by the last `every` steps

    def __exit__(self, *exc):
        self.active = False
        gc.enable()
        gc.unfreeze()
        return False
