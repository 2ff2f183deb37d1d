"""Second opinion: the C oracle against the independent numpy restatement
(tests/naive_ref.py) on seeded random and adversarial (duplicate-heavy, zero-tail)
clouds.  Index outputs must be identical."""
import numpy as np
import pytest
import torch

import naive_ref
from oracle.oracle import OracleExt


def cloud(rng, n, kind):
    p = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    if kind == "dup":                       # up-sampling with replacement (data_preparation_utils.py:38-39)
        base = p[: max(n // 4, 1)]
        p = base[rng.integers(0, len(base), size=n)]
    elif kind == "zero_tail":               # augmentation zeroing (augmentation_utils.py:54)
        p[n - n // 3:] = 0.0
    elif kind == "grid":                    # many exactly equal distances
        p = (rng.integers(0, 4, size=(n, 3)) * 0.25).astype(np.float32)
    return p


@pytest.mark.parametrize("n,m", [(5, 5), (37, 11), (64, 20), (130, 40), (513, 33), (1030, 17)])
@pytest.mark.parametrize("kind", ["uniform", "dup", "zero_tail", "grid"])
def test_fps(n, m, kind):
    rng = np.random.default_rng(n * 1000 + m)
    p = cloud(rng, n, kind)
    want = naive_ref.fps(p, m)
    got = OracleExt.furthest_point_sampling(torch.from_numpy(p)[None], m)[0].tolist()
    assert got == want


@pytest.mark.parametrize("n,m,ns,r", [(50, 7, 4, 0.5), (200, 16, 8, 0.3), (300, 5, 64, 0.9), (64, 3, 2, 0.01)])
@pytest.mark.parametrize("kind", ["uniform", "dup", "grid"])
def test_ball_query(n, m, ns, r, kind):
    rng = np.random.default_rng(n + m)
    p = cloud(rng, n, kind)
    q = p[rng.integers(0, n, size=m)] if kind != "uniform" else cloud(rng, m, "uniform")
    want = naive_ref.ball_query(q, p, r, ns)
    got = OracleExt.ball_query(torch.from_numpy(q)[None].contiguous(), torch.from_numpy(p)[None], r, ns)[0].numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n,m", [(9, 1), (20, 2), (33, 3), (40, 100)])
@pytest.mark.parametrize("kind", ["uniform", "grid"])
def test_three_nn(n, m, kind):
    rng = np.random.default_rng(n * 7 + m)
    u, k = cloud(rng, n, kind), cloud(rng, m, kind)
    d2w, iw = naive_ref.three_nn(u, k)
    d2, idx = OracleExt.three_nn(torch.from_numpy(u)[None], torch.from_numpy(k)[None])
    np.testing.assert_array_equal(idx[0].numpy()[:, :min(m, 3)], iw[:, :min(m, 3)])
    np.testing.assert_array_equal(d2[0].numpy(), d2w)
