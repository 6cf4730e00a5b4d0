"""The report channel of a pipeline wait that gives up (csrc/mevp_p2p.h), seen end to end.  The waits of the stage-per-wave mEVP
pipelines are bounded; in a correct program none ever hits its bound, so the give-up path -- the workgroup's sticky flag, the context's
device counter, the flag in host memory, the error status of the next calls -- never runs in the product build.  The diagnostic build
`giveup` (nextsimdg_amd/build.py: NSDG_P2P_SPIN_LIMIT = 1, built by __graft_entry__.build()) makes every wait give up after one
poll; a process of its own loads it instead of the product library (NSDG_LIB) and reports what a host sees."""
import json
import os
import subprocess
import sys

import pytest

from nextsimdg_amd import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", [4, 3])
def test_a_wait_that_gives_up_is_an_error_status_not_a_hang(gpu, variant):
    lib = build.diag_lib_path("giveup")
    if not os.path.exists(lib):
        lib = build.build_diag("giveup", verbose=False)
    env = dict(os.environ, NSDG_LIB=lib, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "giveup_probe.py"), str(variant)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["launch_returned"]  # a wrong result, never a hung GPU
    assert "gave up" in out["synchronize"], out  # nsdg_ctx_synchronize: NSDG_ERR_HIP once the launch has completed
    assert "gave up" in out["next_subcycle"], out  # sticky: nsdg_mevp_subcycle refuses to start from wrong fields
    assert out["given_up"] > 0 and out["fields_differ"], out
    assert out["synchronize_after_health"] == "ok" and out["given_up_again"] == 0, out  # nsdg_mevp_pipeline_health takes the events
