"""Numpy model of the several-samples-per-hand-off rule of the cluster FPS (csrc/fps.hip, fps_multi_kernel; profiles/HISTORY.md 4c): how
many samples can be accepted from ONE exchange of per-sub-blob candidates, with exact FPS as the referee.  CPU only.

Two rules on a spatially binned 50k-point unit-ball cloud (16^3 Morton cells like fps_bucket_kernel), 2048 samples:
  run  : "reach" rule — a sub-blob is dirty as soon as an accepted sample can reach its bounding box (bound = its old maximum)
  run2 : "candidate" rule (the kernel's) — every sub-blob's cached candidate is updated exactly with each accepted sample and
         stays valid while it is strictly above `second`, the largest distance among the sub-blob's other points
Prints exchanges, samples per exchange, the histogram of samples per exchange, and whether the accepted sequence equals exact FPS.
    python tools/fps_multi_model.py        (about two minutes)"""
import numpy as np, sys
rng = np.random.default_rng(0)
N, m = 50000, 2048
# unit-ball cloud, zero mean / unit sphere like the bench's synthetic scenes
p = rng.normal(size=(N, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True); p *= rng.random((N, 1)) ** (1 / 3)
p -= p.mean(0); p /= np.linalg.norm(p, axis=1).max()
p = p.astype(np.float32)
# Morton order over 16^3 cells of the bounding box
lo, hi = p.min(0), p.max(0)
c = np.minimum(((p - lo) / (hi - lo) * 16).astype(int), 15)
def spread(v):
    r = np.zeros_like(v)
    for b in range(4): r |= ((v >> b) & 1) << (3 * b)
    return r
mort = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
order = np.argsort(mort, kind="stable")
p = p[order]
def run(blob, KMAX, verbose=False):
    nb = (N + blob - 1) // blob
    bid = np.arange(N) // blob
    bb0 = np.array([p[bid == b].min(0) for b in range(nb)]); bb1 = np.array([p[bid == b].max(0) for b in range(nb)])
    td = np.full(N, 1e10, np.float32)
    cur = [p[0]]
    out = [0]
    exchanges = 0
    hist = np.zeros(KMAX + 1, int)
    starts = np.arange(0, N, blob)
    while len(out) < m:
        for s in cur:
            d = ((p - s) ** 2).sum(1).astype(np.float32)
            td = np.minimum(td, d)
        exchanges += 1
        # per-blob best / second best
        best = np.maximum.reduceat(td, starts)
        arg = np.array([starts[b] + np.argmax(td[starts[b]:starts[b] + blob]) for b in range(nb)])
        t2 = td.copy(); t2[arg] = -1
        second = np.maximum.reduceat(t2, starts)
        dirty = np.zeros(nb, bool); bound = np.zeros(nb, np.float32)
        cur = []
        for k in range(KMAX):
            if len(out) + len(cur) >= m: break
            cand = np.where(dirty, -1, best)
            w = int(np.argmax(cand))
            if k > 0 and not (cand[w] > bound[dirty].max()): break
            s = p[arg[w]]
            cur.append(s); out.append(arg[w])
            e = np.maximum(np.maximum(bb0 - s, s - bb1), 0)
            lb = (e ** 2).sum(1) * 0.999998
            newd = (lb < best) & ~dirty
            bound[newd] = best[newd]
            dirty |= newd
            bound[w] = second[w]; dirty[w] = True
        hist[len(cur)] += 1
    return exchanges, hist, out
ref = None
for blob, K in [(1664, 1), (1664, 4), (1664, 8), (832, 8), (416, 8), (208,8)]:
    ex, hist, out = run(blob, K)
    if ref is None: ref = out
    print(blob, K, "exchanges", ex, "samples/exchange %.2f" % (m / ex), hist, "exact" if out == ref else "MISMATCH")

def run2(blob, KMAX):
    """candidate rule: a sub-blob's cached candidate c stays exact while its own running distance (updated exactly with
    every accepted sample) stays strictly above `second` (the largest old distance among the blob's other points)"""
    nb = (N + blob - 1) // blob
    td = np.full(N, 1e10, np.float32)
    cur = [p[0]]; out = [0]; exchanges = 0
    hist = np.zeros(KMAX + 1, int)
    starts = np.arange(0, N, blob)
    while len(out) < m:
        for s in cur:
            d = ((p - s) ** 2).sum(1).astype(np.float32)
            td = np.minimum(td, d)
        exchanges += 1
        best = np.maximum.reduceat(td, starts)
        arg = np.array([starts[b] + np.argmax(td[starts[b]:starts[b] + blob]) for b in range(nb)])
        t2 = td.copy(); t2[arg] = -1
        second = np.maximum.reduceat(t2, starts)
        cd = best.copy()           # candidate's running distance
        dirty = np.zeros(nb, bool)
        cpts = p[arg]
        cur = []
        for k in range(KMAX):
            if len(out) + len(cur) >= m: break
            cand = np.where(dirty, -1, cd)
            w = int(np.argmax(cand))
            if k > 0 and dirty.any() and not (cand[w] > second[dirty].max()): break
            if k > 0 and not (cand[w] > 0): break
            s = cpts[w]
            cur.append(s); out.append(arg[w])
            d = ((cpts - s) ** 2).sum(1).astype(np.float32)
            cd = np.minimum(cd, d)
            dirty |= ~(cd > second)
        hist[len(cur)] += 1
    return exchanges, hist, out
for blob, K in [(1664, 8), (832, 8), (1664, 16)]:
    ex, hist, out = run2(blob, K)
    print("cand rule", blob, K, "exchanges", ex, "samples/exchange %.2f" % (m / ex), hist, "exact" if out == ref else "MISMATCH")
