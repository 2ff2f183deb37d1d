"""Test configuration.

* registers the ``gpu`` marker (tests that need a real MI355X);
* puts ``4d-or_amd/`` (the product) and the repo root (``oracle`` package) on sys.path;
* builds the CPU oracle (gcc) on first use.

CPU-only tests never compute through libpn2_hip.so; they exercise the python layer
on top of the oracle facade (tests/oracle_ext.py), which is test infrastructure.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(REPO, "4d-or_amd")
for p in (PRODUCT, REPO, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    from oracle import oracle as _o
    _o.build()
    return _o


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Route the product's python layer to the CPU oracle for the duration of a test."""
    import oracle_ext
    from pointnet2_ops import pointnet2_utils
    monkeypatch.setattr(pointnet2_utils, "_ext", oracle_ext.OracleRowsExt)
    try:
        from scene_graph_prediction.scene_graph_helpers.model.gcns import network_TripletGCN as g
        monkeypatch.setattr(g, "_ext", oracle_ext.OracleRowsExt)
    except ImportError:
        pass
    return oracle_ext.OracleRowsExt


def has_gpu():
    import torch
    return torch.cuda.is_available()


def assert_same_product(a, b):
    """Two kernel routes that form the SAME fp32 products in the same order are bit-identical — under the default (exact fp32
    MFMA) arithmetic.  With the f32x3 route forced on for the whole suite (PN2_X3=1 PN2_X3_DGRAD=1 PN2_X3_MIN_ROWS=0) one side of
    such a comparison runs on the split-bf16 product and the other on the exact kernel it specialises: they then agree at fp32
    rounding level (1e-5 of the largest value) instead."""
    import torch
    from pointnet2_ops import _ext
    if getattr(_ext, "X3_GEMM", False):
        torch.testing.assert_close(a, b, atol=1e-5 * max(1.0, float(b.abs().max())), rtol=1e-5)
    else:
        assert torch.equal(a, b), float((a - b).abs().max())
