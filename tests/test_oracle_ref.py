"""oracle/_ref — real reference builds.  The only reference source that compiles in this image is the Cython
Graphormer pre-processing module (role_prediction/graphormer/algos.pyx -> oracle/_ref/algos*.so, `make -C oracle ref`).
Known answers derived by hand from the .pyx pin that the build is the reference's code and behaves as read:
the MAX_DIST = 12 marker, intermediate-vertex matrix, and the `path == 0 means direct edge` rule of get_all_edges."""
import os
import sys

import numpy as np
import pytest

REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


@pytest.fixture(scope="module")
def ref_algos():
    import importlib
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    saved = sys.modules.get("role_prediction.graphormer.algos")
    try:
        mod = importlib.import_module("algos")
    except ImportError:
        pytest.skip("oracle/_ref not built: needs /root/reference (build container) — run `make -C oracle ref`")
    if saved is not None:
        sys.modules["role_prediction.graphormer.algos"] = saved
    else:
        sys.modules.pop("role_prediction.graphormer.algos", None)
    return mod


def test_chain_graph_known_answer(ref_algos):
    adj = np.zeros((4, 4), dtype=bool)
    adj[0, 1] = adj[1, 2] = adj[2, 3] = True                     # 0 -> 1 -> 2 -> 3
    M, path = ref_algos.floyd_warshall(adj)
    assert M.tolist() == [[0, 1, 2, 3], [12, 0, 1, 2], [12, 12, 0, 1], [12, 12, 12, 0]]
    assert path.tolist() == [[0, 0, 1, 2], [12, 0, 0, 2], [12, 12, 0, 0], [12, 12, 12, 0]]
    feat = np.arange(16).reshape(4, 4, 1)
    e = ref_algos.gen_edge_input(3, path, feat)
    assert e[0, 3].ravel().tolist() == [1, 6, 11]               # edges (0,1), (1,2), (2,3)
    assert e[0, 2].ravel().tolist() == [1, 6, -1]
    assert (e[3] == -1).all() and (e[1, 0] == -1).all()


def test_vertex_zero_is_never_expanded(ref_algos):
    """2 -> 0 -> 1: the shortest path from 2 to 1 runs through vertex 0, which `path` stores as 0 = "direct edge", so
    gen_edge_input emits the (2, 1) feature although that edge does not exist — a reference quirk the kernels keep."""
    adj = np.zeros((3, 3), dtype=bool)
    adj[2, 0] = adj[0, 1] = True
    M, path = ref_algos.floyd_warshall(adj)
    assert M[2, 1] == 2 and path[2, 1] == 0
    feat = (np.arange(9).reshape(3, 3, 1) + 1) * 10
    e = ref_algos.gen_edge_input(2, path, feat)
    assert e[2, 1].ravel().tolist() == [int(feat[2, 1, 0]), -1]
