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
    """the point-to-point mEVP pipelines (csrc/mevp_p2p.h) bound every wait and report the waits that gave up per context: an error
    status from nsdg_ctx_synchronize / nsdg_mevp_subcycle / nsdg_rb_mevp_run, a count from nsdg_mevp_pipeline_health.  Tests that
    synchronise through torch never see the status, so every Context adds what nobody took to abi.GIVEN_UP_UNSEEN when it is closed:
    after a GPU session the tally must be zero (a non-zero count means a dependency was waited for that never came: wrong results)"""
    yield
    from nextsimdg_amd import abi

    import gc

    gc.collect()  # contexts that were dropped without close()
    assert abi.GIVEN_UP_UNSEEN == 0, "the mEVP pipeline gave up %d wait(s)" % abi.GIVEN_UP_UNSEEN
