"""Independent numpy restatements of the op semantics (second opinion for the
oracle).  Written from the behavioural description in SURVEY.md §2a, using
explicit total orders instead of simulating threads, so an error in the C
oracle's lane simulation and an error here are unlikely to coincide."""
import math

import numpy as np

f32 = np.float32


def _fma(a, b, c):
    # exact fused multiply-add in fp32 via float64 (53 bits hold the 48-bit product
    # exactly; one extra rounding of the sum to f64 before f32 can double-round only
    # in astronomically rare ties; tests use inputs where it does not)
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def sq3(dx, dy, dz):
    t = f32(dy * dy)
    t = _fma(dx, dx, t)
    return _fma(dz, dz, t)


def ref_block(n):
    p = int(math.log(float(n)) / math.log(2.0))
    return max(min(1 << p, 512), 1)


def bitrev(v, bits):
    r = 0
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def fps(xyz, m):
    """xyz (N,3) f32 -> list of m indices.  Winner = max running distance; ties ->
    smaller bit-reversed (k mod bs), then smaller k; skipped points never win."""
    N = xyz.shape[0]
    bs = ref_block(N)
    L = bs.bit_length() - 1
    mag = np.array([sq3(*p) for p in xyz], dtype=f32)
    valid = ~(mag.astype(np.float64) <= 1e-3)
    temp = np.full(N, 1e10, dtype=f32)
    rank = np.array([bitrev(k % bs, L) * (N // bs + 1) + k // bs for k in range(N)])
    out = [0]
    old = 0
    for _ in range(1, m):
        d = np.array([sq3(*(xyz[k] - xyz[old])) for k in range(N)], dtype=f32)
        temp = np.where(valid, np.fmin(d, temp), temp)
        if not valid.any():
            old = 0
        else:
            cand = np.where(valid)[0]
            best = temp[cand].max()
            tied = cand[temp[cand] == best]
            old = int(tied[np.argmin(rank[tied])])
        out.append(old)
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    r2 = f32(f32(radius) * f32(radius))
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    for j in range(m):
        hits = [k for k in range(xyz.shape[0]) if sq3(*(new_xyz[j] - xyz[k])) < r2][:nsample]
        if hits:
            idx[j, :] = hits[0]
            idx[j, :len(hits)] = hits
    return idx


def three_nn(unknown, known):
    n = unknown.shape[0]
    d2 = np.full((n, 3), np.inf, dtype=f32)
    idx = np.zeros((n, 3), dtype=np.int32)
    for j in range(n):
        d = np.array([sq3(*(unknown[j] - known[k])) for k in range(known.shape[0])], dtype=f32)
        order = np.argsort(d, kind="stable")[:3]      # stable: earliest index wins ties
        d2[j, :len(order)] = d[order]
        idx[j, :len(order)] = order
    return d2, idx
