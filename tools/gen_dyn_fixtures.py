"""Writes the dynamics SELF-FIXTURE tests/golden/dyn_selfcheck_v1.{json,f64} from oracle/dyn_oracle.c.

    python tools/gen_dyn_fixtures.py            # regenerate (only after a DELIBERATE change of the specification)
    python tools/gen_dyn_fixtures.py --check    # recompute and compare with the committed files, exit 1 on any difference

"self-fixture -- not reference parity": the reference snapshot has no DG / mEVP code; the oracle
is the specification of that path and these files freeze it.  Cases and input recipes: tests/dyn_fixture_cases.py.

Format: the .f64 file is the concatenation of the little-endian float64 arrays (exact; ~1.5 MB, too large as text);
the .json index names each array (case, name, shape, offset in doubles), its sha256 and, for a human reader, its first
four values as C99 hex floats, plus the compiler and libm the oracle was built with (ice strength uses exp: the bytes
of the mEVP cases depend on glibc's exp to the last bit; the transport cases use + - * / only).
"""
import argparse
import hashlib
import json
import os
import platform
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dyn_fixture_cases as cases  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
JSON_PATH = os.path.join(GOLDEN, "dyn_selfcheck_v1.json")
DATA_PATH = os.path.join(GOLDEN, "dyn_selfcheck_v1.f64")


def compute():
    """-> (index entries, bytes)"""
    entries, blobs, off = [], [], 0
    for case in sorted(cases.CASES):
        out = cases.CASES[case]()
        for name in sorted(out):
            a = np.ascontiguousarray(out[name], dtype="<f8")
            raw = a.tobytes()
            entries.append(dict(case=case, name=name, shape=list(a.shape), offset=off, sha256=hashlib.sha256(raw).hexdigest(),
                                first=[float(x).hex() for x in a.reshape(-1)[:4]], absmax=float(np.max(np.abs(a))).hex()))
            blobs.append(raw)
            off += a.size
    return entries, b"".join(blobs)


def toolchain():
    gcc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    return dict(gcc=gcc, libc=" ".join(platform.libc_ver()), numpy=np.__version__, machine=platform.machine())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    entries, data = compute()
    if args.check:
        idx = json.load(open(JSON_PATH))
        old = {(e["case"], e["name"]): e["sha256"] for e in idx["arrays"]}
        new = {(e["case"], e["name"]): e["sha256"] for e in entries}
        bad = sorted(k for k in set(old) | set(new) if old.get(k) != new.get(k))
        same_bytes = open(DATA_PATH, "rb").read() == data
        print("dyn_selfcheck_v1: %d arrays, %d differ, data file %s" % (len(new), len(bad), "identical" if same_bytes else "DIFFERS"))
        for k in bad:
            print("  differs:", k)
        sys.exit(1 if bad or not same_bytes else 0)
    index = dict(title="dynamics self-fixture v1 -- self-fixture, NOT reference parity (SURVEY.md section 0: the reference has no DG / mEVP code)",
                 source="oracle/dyn_oracle.c (+ oracle/column_oracle.c for coupled_step) through tests/dyn_fixture_cases.py",
                 generator="tools/gen_dyn_fixtures.py", dtype="<f8", data_file=os.path.basename(DATA_PATH),
                 data_sha256=hashlib.sha256(data).hexdigest(), toolchain=toolchain(), arrays=entries)
    with open(DATA_PATH, "wb") as f:
        f.write(data)
    with open(JSON_PATH, "w") as f:
        json.dump(index, f, indent=1)
        f.write("\n")
    print("wrote %s (%d arrays) and %s (%d bytes)" % (JSON_PATH, len(entries), DATA_PATH, len(data)))


if __name__ == "__main__":
    main()
