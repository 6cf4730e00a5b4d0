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
    """the point-to-point mEVP pipeline (csrc/mevp_fused4.hip) bounds every wait and counts the waits that gave up
    (nsdg_mevp_pipeline_health): after a GPU session the counter must be zero (a non-zero count means a dependency was waited
    for that never came: wrong results)"""
    yield
    import torch

    from nextsimdg_amd import abi

    if abi._lib is None or not torch.cuda.is_available():
        return
    ctx = abi.Context(torch.device("cuda:0"))
    n = ctx.pipeline_waits_given_up()
    ctx.close()
    assert n == 0, "the mEVP pipeline gave up %d wait(s)" % n
