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
