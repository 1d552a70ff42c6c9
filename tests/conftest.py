"""pytest configuration: the ``gpu`` marker, import paths, and shared scene builders.

``-m "not gpu"`` runs here (CPU only): oracle vs closed forms / autograd / golden vectors, host logic,
C-ABI export check, world_size-2 gloo data-parallel test.  ``-m gpu`` runs on the MI355X: the HIP path
(through the C ABI) against the oracle and the golden fixtures.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as _orc

    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def dns():
    import dn_splatter_amd as _dns

    return _dns
