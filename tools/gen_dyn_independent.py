"""Writes tests/golden/dyn_independent_v3.npz: inputs and outputs of tests/dyn_independent.py (the independent dense numpy
restatement of DESIGN.md section 3) on its 6 x 5 case -- per-step preparation, ONE mEVP sub-iteration, ONE DG2 transport
stage, and the closure of the transport (cap + scaling limiter, round 5) on the case's H and A; v2: with the ice-free-node rule; v3: the sub-iteration also with local, solution-adaptive alpha and beta (round 6).  NOT reference parity: the reference snapshot has no dynamics code (/root/reference/CMakeLists.txt:43-46); the file
pins the oracle AND the HIP path to a second, independently written statement of the scheme.
usage: python tools/gen_dyn_independent.py [--check]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import dyn_independent as D  # noqa: E402

path = os.path.join(root, "tests", "golden", "dyn_independent_v3.npz")
inp = D.case_inputs()
out = D.case_outputs(inp)
arrays = {"in_" + k: (np.stack(v) if isinstance(v, list) else v) for k, v in inp.items()}
arrays.update({"out_" + k: v for k, v in out.items()})
if "--check" in sys.argv:
    old = np.load(path)
    assert sorted(old.files) == sorted(arrays), "array names differ"
    worst = max(float(np.max(np.abs(old[k] - v)) / max(np.max(np.abs(v)), 1e-300)) for k, v in arrays.items())
    print("dyn_independent_v3.npz: %d arrays, largest relative difference from a fresh evaluation %.2e" % (len(arrays), worst))
    sys.exit(0 if worst < 1e-13 else 1)
np.savez(path, **arrays)
print("wrote", path, "%d arrays, %d bytes" % (len(arrays), os.path.getsize(path)))
