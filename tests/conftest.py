import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a machine without a GPU: they are simply deselected by
    # '-m "not gpu"'; if someone runs them without a device they fail loudly in the fixture below.
    pass


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a HIP device; run it on the GPU box")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _no_pipeline_wait_gave_up():
    """the point-to-point mEVP pipeline (csrc/mevp_fused4.hip) bounds every wait and counts the waits that gave up: after a GPU
    session the counter must be zero (a non-zero count means a dependency was waited for that never came: wrong results)"""
    yield
    import ctypes

    import torch

    from nextsimdg_amd import abi

    if abi._lib is None or not torch.cuda.is_available() or not hasattr(abi._lib, "nsdg_debug_p2p_timeouts"):
        return
    n = ctypes.c_uint(0)
    rc = abi._lib.nsdg_debug_p2p_timeouts(ctypes.byref(n))
    assert rc == 0 and n.value == 0, "the mEVP pipeline gave up %d wait(s)" % n.value
