"""Diagnostic for the zero-denominator parity case that failed in round 2 (gpurun_out/r02_t3.log): the ORIGINAL inputs
of that test (cice == min_conc on O(1) cell means, qlw = Inf) through the column kernel and the CPU oracle, with every
input and diagnostic of the elements that disagree written as hex floats.  Test infrastructure (imports the oracle).

    python tools/diag_column_cutoff.py > gpurun_out/r03_cutoff_diag.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from nextsimdg_amd import abi, synthetic  # noqa: E402


def original_inputs(n=4096):
    state, forcing, newice = synthetic.column_fields(n, seed=99)
    rng = np.random.default_rng(5)
    forcing["mld"][rng.random(n) < 0.25] = 0.0
    state["cice"][rng.random(n) < 0.1] = 1e-12
    state["cice"][rng.random(n) < 0.05] = 1e-13
    state["hice"][rng.random(n) < 0.05] = 5e-324
    forcing["qlw"][:7] = np.inf
    forcing["wind"][rng.random(n) < 0.3] = 0.0
    return state, forcing, newice


def main():
    dt = 600.0
    state, forcing, newice = original_inputs()
    s0 = {k: v.copy() for k, v in state.items()}
    n0 = newice.copy()
    ctx = abi.Context(torch.device("cuda:0"))
    ctx.set_column_params(ctx.column_default_params())
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ds, df, dn = {k: dev(v) for k, v in state.items()}, {k: dev(v) for k, v in forcing.items()}, dev(newice)
    diag = torch.zeros(abi.NDIAG, len(newice), dtype=torch.float64, device="cuda")
    with np.errstate(all="ignore"):
        want = O.column_step(O.column_params(), dt, state, forcing, newice, want_diag=True)
    ctx.column_step(dt, ds, df, dn, diag)
    got = {k: ds[k].cpu().numpy() for k in abi.STATE}
    gd = diag.cpu().numpy()
    bad = np.zeros(len(newice), bool)
    for k in abi.STATE:
        g, w = got[k], state[k]
        with np.errstate(all="ignore"):
            differ = ~((g == w) | (np.isnan(g) & np.isnan(w)) | (np.abs(g - w) <= 1e-13 + 1e-11 * np.abs(w)))
        bad |= differ
    idx = np.where(bad)[0]
    print("elements that disagree: %d of %d" % (len(idx), len(newice)))
    hx = lambda x: float(x).hex()
    for i in idx:
        print("--- element", i)
        print("  in :", " ".join("%s=%s(%.6g)" % (k, hx(s0[k][i]), s0[k][i]) for k in abi.STATE), "newice=%s" % hx(n0[i]))
        print("  frc:", " ".join("%s=%s(%.6g)" % (k, hx(forcing[k][i]), forcing[k][i]) for k in forcing))
        print("  want:", " ".join("%s=%.17g" % (k, state[k][i]) for k in abi.STATE), "newice=%.17g" % newice[i])
        print("  got :", " ".join("%s=%.17g" % (k, got[k][i]) for k in abi.STATE), "newice=%.17g" % dn[i].item())
        for j, k in enumerate(abi.DIAG):
            w, g = want[k][i], gd[j][i]
            flag = "" if (w == g or (np.isnan(w) and np.isnan(g))) else "   <-- differs"
            print("    %-8s want %-24s got %-24s%s" % (k, hx(w), hx(g), flag))


if __name__ == "__main__":
    main()
