import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "probes: runs against libgigapose_hip_probe.so (the -DGP_PROBES build: A/B switches, test-only "
                                       "epilogues, error-word readers; include/gigapose_hip_probe.h) instead of the product library")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` here."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _suite_numerics(monkeypatch):
    """The product default is "split" (gigapose_amd/_lib.py: default_numerics; tests/test_numerics_default.py).  The suite's
    baseline is the verification mode "chain" -- the kernels that are bit-exact against the CPU oracle -- so every test that
    does not name a mode itself (monkeypatch.setenv / set_numerics / a `numerics` parameter) runs chain, whatever the caller's
    environment holds; the split tests all set it explicitly."""
    monkeypatch.setenv("GIGAPOSE_NUMERICS", "chain")


@pytest.fixture(autouse=True)
def _probe_library(request):
    """Tests marked `probes` need a hook the product library does not carry (an A/B switch, a plain-f32 epilogue of the plane GEMM, the
    stream-K scratch's error word): their calls go to libgigapose_hip_probe.so -- the same sources with -DGP_PROBES -- for the duration
    of the test.  Every other test runs against the product library."""
    if "probes" not in request.keywords:
        yield
        return
    from gigapose_amd import _lib

    with _lib.probe_library():
        yield
