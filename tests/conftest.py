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


@pytest.fixture
def hip_deterministic(dns):
    """The HIP path in its deterministic gradient mode for the duration of a test (dnsplat_det_reduce: every Gaussian's rows added in
    list order in double).  For tests that compare TWO GPU runs of the same frame — batched vs sequential, exchange on vs off, tight
    vs gsplat tile boxes: in the default mode the two differ by the arrival order of fp32 atomics, up to ~1e-4 of the tensor's scale
    on ill-conditioned entries (an anisotropic scene's quaternion gradient drew 0.01 ... 0.9 of its allowance over the round's
    runs); in this mode they must agree to the last bit or two, whatever the tolerance written in the test."""
    from dn_splatter_amd import _ops

    prev = _ops.DETERMINISTIC["on"]
    dns.set_deterministic(True)
    yield
    dns.set_deterministic(prev)
