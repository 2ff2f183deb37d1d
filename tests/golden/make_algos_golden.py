"""Generates tests/golden/algos.npz from the REFERENCE's own Cython module (role_prediction/graphormer/algos.pyx compiled
into oracle/_ref by `make -C oracle ref`; only possible in the build container where /root/reference exists).  The
fixture holds inputs and the reference's outputs (hop distances, intermediate vertices, edge features along the paths) for
the nine graphs of tests/test_gpu_round2.py plus their transposes and the empty graph — data, no reference text — so the
GPU box checks pn2_floyd_warshall / pn2_gen_edge_input without importing anything built from the reference.

    make -C oracle ref && python tests/golden/make_algos_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.abspath(os.path.join(HERE, "..", "..", "oracle", "_ref"))
CASES = [(1, 0.5, 0), (2, 1.0, 1), (5, 0.3, 2), (13, 0.15, 3), (14, 0.12, 4), (20, 0.1, 5), (31, 0.05, 6), (40, 0.04, 7),
         (9, 0.0, 8)]


def main():
    assert os.path.isdir("/root/reference"), "the fixture is generated from the reference's module: build container only"
    sys.path.insert(0, REF_DIR)
    ref = importlib.import_module("algos")
    assert ref.__file__.endswith(".so") and os.path.dirname(os.path.abspath(ref.__file__)) == REF_DIR
    out = {}
    for n, density, seed in CASES:
        rng = np.random.default_rng(seed)
        adj = rng.random((n, n)) < density
        np.fill_diagonal(adj, False)
        feat = rng.integers(1, 50, size=(n, n, 3))
        key = f"n{n}_s{seed}"
        out[key + "_adj"], out[key + "_feat"] = adj, feat
        for tag, a in (("", adj), ("_T", adj.T.copy()), ("_Z", np.zeros_like(adj))):
            M, P = ref.floyd_warshall(a)
            out[key + tag + "_M"], out[key + tag + "_P"] = M, P
        M, P = out[key + "_M"], out[key + "_P"]
        md = int(np.amax(M)) if n else 0
        if md > 0:
            out[key + "_E"] = ref.gen_edge_input(md, P, feat)
    np.savez_compressed(os.path.join(HERE, "algos.npz"), **out)
    print("wrote", os.path.join(HERE, "algos.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
