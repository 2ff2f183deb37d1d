"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/pn2_hip.h declares (no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "pn2_hip.h")
LIB = os.path.join(REPO, "4d-or_amd", "libpn2_hip.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_nine_reference_ops():
    syms = declared_symbols()
    for op in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
               "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
               "group_points_grad"):                      # EXT/src/bindings.cpp:6-19
        assert f"pn2_{op}" in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.pn2_abi_version.restype = ctypes.c_int
    assert lib.pn2_abi_version() == 3
    lib.pn2_strerror.restype = ctypes.c_char_p
    lib.pn2_strerror.argtypes = [ctypes.c_int]
    assert lib.pn2_strerror(-1) and lib.pn2_strerror(12345)


def test_python_binding_covers_every_symbol_and_refuses_cpu_tensors():
    import torch
    from pointnet2_ops import _ext
    assert sorted(_ext.EXPORTED_SYMBOLS) == declared_symbols()
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(torch.zeros(1, 8, 3), 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.ball_query(torch.zeros(1, 3, 2).transpose(1, 2), torch.zeros(1, 4, 3), 0.1, 2)
    with pytest.raises(RuntimeError, match="int tensor"):
        _ext.gather_points(torch.zeros(1, 3, 4), torch.zeros(1, 2, dtype=torch.int64))


def test_ctypes_signatures_have_the_header_arity():
    """Every ctypes argtypes list (incl. the trailing stream) has as many entries as the C prototype has parameters: a
    missing entry makes ctypes pass the 64-bit stream handle as a 32-bit int (hipErrorInvalidValue at launch)."""
    from pointnet2_ops import _ext
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(pn2_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(1), m.group(2)
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        if name in _ext._SIGNATURES:
            assert len(_ext._SIGNATURES[name]) == n, name
            seen += 1
    assert seen == len(_ext._SIGNATURES)
