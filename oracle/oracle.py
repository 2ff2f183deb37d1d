"""ctypes binding of the CPU oracle + a ``pointnet2_ops._ext``-shaped facade.

TEST INFRASTRUCTURE ONLY.  ``OracleExt`` exposes the nine functions of the
reference's pybind11 module (EXT/src/bindings.cpp:6-19) on CPU torch tensors so
that (a) the reference's own python layer can be imported on top of it inside
the build container (tests/golden/make_golden.py) and (b) the CPU test-suite can
exercise the build's python layer without a GPU.  It is never reachable from the
product path.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpn2_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpn2_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_opt_n_threads.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed rc={rc}")


def _f32(t, name):
    assert t.device.type == "cpu", f"oracle is CPU-only ({name})"
    assert t.dtype == torch.float32, f"{name} must be a float tensor"
    assert t.is_contiguous(), f"{name} must be a contiguous tensor"
    return t


def _i32(t, name):
    assert t.device.type == "cpu", f"oracle is CPU-only ({name})"
    assert t.dtype == torch.int32, f"{name} must be an int tensor"
    assert t.is_contiguous(), f"{name} must be a contiguous tensor"
    return t


def opt_n_threads(n):
    return int(lib().orc_opt_n_threads(ctypes.c_int(int(n))))


def num_threads():
    return int(lib().orc_num_threads())


class OracleExt:
    """Same nine entry points, argument order and allocation behaviour as the
    reference's ``pointnet2_ops._ext`` (zero-initialised outputs, 1e10 FPS
    scratch), computing on the CPU oracle."""

    @staticmethod
    def furthest_point_sampling(points, nsamples):
        _f32(points, "points")
        B, N, _ = points.shape
        out = torch.zeros(B, nsamples, dtype=torch.int32)
        tmp = torch.full((B, N), 1e10, dtype=torch.float32)
        _chk(lib().orc_furthest_point_sampling(B, N, int(nsamples), _p(points), _p(tmp), _p(out)), "fps")
        return out

    @staticmethod
    def gather_points(points, idx):
        _f32(points, "points"); _i32(idx, "idx")
        B, C, N = points.shape
        m = idx.shape[1]
        out = torch.zeros(B, C, m, dtype=torch.float32)
        _chk(lib().orc_gather_points(B, C, N, m, _p(points), _p(idx), _p(out)), "gather_points")
        return out

    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        _f32(grad_out, "grad_out"); _i32(idx, "idx")
        B, C, m = grad_out.shape
        out = torch.zeros(B, C, int(n), dtype=torch.float32)
        _chk(lib().orc_gather_points_grad(B, C, int(n), m, _p(grad_out), _p(idx), _p(out)), "gather_points_grad")
        return out

    @staticmethod
    def ball_query(new_xyz, xyz, radius, nsample):
        _f32(new_xyz, "new_xyz"); _f32(xyz, "xyz")
        B, m, _ = new_xyz.shape
        N = xyz.shape[1]
        idx = torch.zeros(B, m, int(nsample), dtype=torch.int32)
        _chk(lib().orc_ball_query(B, N, m, ctypes.c_float(float(radius)), int(nsample),
                                  _p(new_xyz), _p(xyz), _p(idx)), "ball_query")
        return idx

    @staticmethod
    def group_points(points, idx):
        _f32(points, "points"); _i32(idx, "idx")
        B, C, N = points.shape
        _, npoints, nsample = idx.shape
        out = torch.zeros(B, C, npoints, nsample, dtype=torch.float32)
        _chk(lib().orc_group_points(B, C, N, npoints, nsample, _p(points), _p(idx), _p(out)), "group_points")
        return out

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        _f32(grad_out, "grad_out"); _i32(idx, "idx")
        B, C, npoints, nsample = grad_out.shape
        out = torch.zeros(B, C, int(n), dtype=torch.float32)
        _chk(lib().orc_group_points_grad(B, C, int(n), npoints, nsample, _p(grad_out), _p(idx), _p(out)),
             "group_points_grad")
        return out

    @staticmethod
    def three_nn(unknowns, knows):
        _f32(unknowns, "unknowns"); _f32(knows, "knows")
        B, n, _ = unknowns.shape
        m = knows.shape[1]
        idx = torch.zeros(B, n, 3, dtype=torch.int32)
        dist2 = torch.zeros(B, n, 3, dtype=torch.float32)
        _chk(lib().orc_three_nn(B, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx)), "three_nn")
        return [dist2, idx]

    @staticmethod
    def three_interpolate(points, idx, weight):
        _f32(points, "points"); _i32(idx, "idx"); _f32(weight, "weight")
        B, C, m = points.shape
        n = idx.shape[1]
        out = torch.zeros(B, C, n, dtype=torch.float32)
        _chk(lib().orc_three_interpolate(B, C, m, n, _p(points), _p(idx), _p(weight), _p(out)), "three_interpolate")
        return out

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):
        _f32(grad_out, "grad_out"); _i32(idx, "idx"); _f32(weight, "weight")
        B, C, n = grad_out.shape
        out = torch.zeros(B, C, int(m), dtype=torch.float32)
        _chk(lib().orc_three_interpolate_grad(B, C, n, int(m), _p(grad_out), _p(idx), _p(weight), _p(out)),
             "three_interpolate_grad")
        return out

    # --- TripletGCN primitives (torch_geometric / torch_scatter restatement) ---
    @staticmethod
    def gather_rows(x, index):
        _f32(x, "x")
        assert index.dtype == torch.int64 and index.is_contiguous()
        N, H = x.shape
        E = index.numel()
        out = torch.zeros(E, H, dtype=torch.float32)
        _chk(lib().orc_gather_rows(ctypes.c_int64(E), ctypes.c_int64(H), ctypes.c_int64(N),
                                   _p(x), _p(index), _p(out)), "gather_rows")
        return out

    @staticmethod
    def scatter_add_rows(src, index, dim_size):
        _f32(src, "src")
        assert index.dtype == torch.int64 and index.is_contiguous()
        E, H = src.shape
        out = torch.zeros(int(dim_size), H, dtype=torch.float32)
        _chk(lib().orc_scatter_add_rows(ctypes.c_int64(E), ctypes.c_int64(H), ctypes.c_int64(int(dim_size)),
                                        _p(src), _p(index), _p(out)), "scatter_add_rows")
        return out
