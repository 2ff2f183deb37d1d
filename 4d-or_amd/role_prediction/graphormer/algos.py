"""``algos`` of the Graphormer role-prediction pre-processing on MI355X.

API mirror of role_prediction/graphormer/algos.pyx (``floyd_warshall`` :11-54, ``gen_edge_input`` :62-89; the reference
compiles it with Cython and calls it once per graph from wrapper.py:39-41): same names, same argument order, same
int64 results, numpy in -> numpy out.  Torch tensors on the GPU are accepted too and may carry a leading batch
dimension — that is the MI355X-first use: all graphs of a data-loader batch in ONE launch (csrc/graph_algos.hip)
instead of a Python loop of tiny host calls.  There is no host fallback.
"""
import numpy as np
import torch

from pointnet2_ops import _ext

MAX_DIST = 12


def _to_device(a):
    t = torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)) if isinstance(a, np.ndarray) else a.to(torch.int64)
    return t.contiguous().cuda() if not t.is_cuda else t.contiguous()


def floyd_warshall(adjacency_matrix):
    """(n, n) [or (B, n, n)] adjacency (bool / int) -> (M, path): hop distances with unreachable = 12, intermediate vertices."""
    as_numpy = isinstance(adjacency_matrix, np.ndarray)
    adj = _to_device(adjacency_matrix)
    single = adj.dim() == 2
    assert adj.size(-1) == adj.size(-2)
    M, path = _ext.floyd_warshall(adj.unsqueeze(0) if single else adj)
    if single:
        M, path = M[0], path[0]
    return (M.cpu().numpy(), path.cpu().numpy()) if as_numpy else (M, path)


def gen_edge_input(max_dist, path, edge_feat):
    """(n, n) path, (n, n, F) edge_feat [or batched] -> (n, n, max_dist, F) int64, -1 where no edge."""
    as_numpy = isinstance(path, np.ndarray)
    p, f = _to_device(path), _to_device(edge_feat)
    single = p.dim() == 2
    out = _ext.gen_edge_input(int(max_dist), p.unsqueeze(0) if single else p, f.unsqueeze(0) if single else f)
    if single:
        out = out[0]
    return out.cpu().numpy() if as_numpy else out
