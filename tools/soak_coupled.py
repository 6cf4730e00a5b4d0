"""BASELINE config 5 as a run: dynamics + column thermodynamics for `steps` model steps of 120 s (720 = one day) on an
n x n box, with the device-side forcing providers -- cyclone wind moving with model time (nsdg_boxtest_forcing), winter
thermodynamic forcing with a diurnal short-wave cycle (nsdg_column_forcing) and the column wind speed taken from the
dynamics' wind (nsdg_column_wind) -- on the native row-block driver.  Prints ranges every `every` steps and the wall time.

Concentration overshoot: neither process that changes A has a cap in the reference or here -- the column step's Hibler
freeze (HiblerConcentration.cpp:32-38: del_c = newice / h0, no limit at 1) and the unlimited DG2 transport (convergent flow
piles concentration up; no limiter exists in the scheme).  The run therefore books, per step and on the device, the
EXCESS E = sum over elements of max(A - 1, 0) (cell means) before the column step, after it and after the transport, and
reports which of the two produced it (cumulative, and over the last reporting interval), with the maxima of A after each.
usage: python tools/soak_coupled.py [steps=720] [n=512] [forcing=winter|dummy|host]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nextsimdg_amd import abi, rowblock, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 720
nx = ny = int(sys.argv[2]) if len(sys.argv) > 2 else 512
forcing = sys.argv[3] if len(sys.argv) > 3 else "winter"
L, dt, nsub = 512e3, 120.0, 120
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
bt = synthetic.BoxTest(nx, ny, L)
alpha = bt.stable_alpha(dt) * float(os.environ.get("NSDG_ALPHA_SCALE", "1"))  # NSDG_ALPHA_SCALE: a wider stability margin than the default 2.4 x the bound
ctx.set_mevp_params(ctx.mevp_default_params(alpha=alpha, beta=alpha))
blk = rowblock.RowBlock(nx, ny, 0, 1)
core = rowblock.CoupledCore(ctx, blk, L / nx, L / ny, dt, nsub, dev, native=True, forcing=None if forcing == "host" else forcing)
cs, cf = synthetic.column_fields_smooth(nx, ny, L)  # initial snow / ice temperature; sst at the freezing point (see the docstring there)
H, A = bt.dg_fields()
if os.environ.get("NSDG_SOAK_INIT") == "const":  # the constant initial state of the C++ host's [init] keys (tools/r03_config5_day.sh)
    import numpy as np

    H[:], A[:] = 0.0, 0.0
    H[0], A[0] = 0.3, 0.9
    cs = {"hsnow": np.full((ny, nx), 0.05), "tice0": np.full((ny, nx), -8.0)}
    cf["sst"], cf["sss"] = np.full((ny, nx), -1.76), np.full((ny, nx), 32.0)
core.load_column({**cs, **cf})
uo, vo = bt.ocean()
ua, va = bt.wind(0.0)
core.load_global(H, A, uo, vo, ua, va)
del H, A, uo, vo, ua, va, cs, cf
m0 = float(core.H[0].sum())
every = int(os.environ.get("NSDG_SOAK_EVERY", max(1, steps // 12)))
torch.cuda.synchronize()
t0 = time.perf_counter()
excess = lambda: (core.A[0] - 1.0).clamp_min(0.0).sum()
zero = lambda: torch.zeros((), dtype=torch.float64, device=dev)
made = {"column": zero(), "transport": zero()}  # excess produced by each process (device scalars: no host sync per step)
amax = {"column": zero(), "transport": zero()}  # largest A seen right after each process
last = {"column": 0.0, "transport": 0.0}
for step in range(steps):
    core.device_wind(L, step * dt)  # the cyclone moves
    # core.step(), with the excess booked between its parts
    core._set_grid()
    core.external_forcing()
    e0 = excess()
    core.thermodynamics()
    e1 = excess()
    amax["column"] = torch.maximum(amax["column"], core.A[0].max())
    core.momentum()
    core.transport()
    core.time += core.dt
    e2 = excess()
    amax["transport"] = torch.maximum(amax["transport"], core.A[0].max())
    made["column"] += e1 - e0
    made["transport"] += e2 - e1
    if step % every == 0 or step == steps - 1:
        fin = all(bool(torch.isfinite(f).all()) for f in (core.u, core.v, core.H, core.A, core.col["hsnow"], core.col["tice0"]))
        if not fin:
            for name, f in (("u", core.u), ("v", core.v), ("H", core.H), ("A", core.A), ("hsnow", core.col["hsnow"]), ("tice0", core.col["tice0"])):
                bad = ~torch.isfinite(f)
                if bool(bad.any()):
                    idx = bad.nonzero()[:5].tolist()
                    print("           non-finite %s: %d entries, first at %s" % (name, int(bad.sum()), idx), flush=True)
        print("step %4d  t = %5.2f h  finite %s  umax %.3g  H [%.4f, %.4f]  A [%.4f, %.4f]  tice [%.2f, %.2f]  hsnow [%.3f, %.3f]  wind max %.1f  qsw max %.0f  newice max %.2e"
              % (step, (step + 1) * dt / 3600.0, fin, float(core.u.abs().max()), float(core.H[0].min()), float(core.H[0].max()),
                 float(core.A[0].min()), float(core.A[0].max()), float(core.col["tice0"].min()), float(core.col["tice0"].max()),
                 float(core.col["hsnow"].min()), float(core.col["hsnow"].max()), float(core.col["wind"].max()), float(core.col["qsw"].max()),
                 float(core.newice.max())), flush=True)
        tot = {k: float(v) for k, v in made.items()}
        print("           A > 1: excess sum(max(A - 1, 0)) = %.4e; produced so far by the column step (uncapped Hibler freeze) %.4e, by the DG2 "
              "transport (no limiter) %.4e; since the last report %.3e / %.3e; max A right after the column step %.6f, after the transport %.6f"
              % (float(excess()), tot["column"], tot["transport"], tot["column"] - last["column"], tot["transport"] - last["transport"],
                 float(amax["column"]), float(amax["transport"])), flush=True)
        last = tot
        if not fin or float(core.u.abs().max()) > 5.0:
            raise SystemExit("the coupled run left the physical range")
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("%d steps of %d x %d (%.1f model hours) in %.1f s wall: %.1f ms per step, %.3g element-steps/s; ice volume change %.3e (thermodynamic growth)"
      % (steps, nx, ny, steps * dt / 3600.0, wall, 1e3 * wall / steps, nx * ny * steps / wall, float(core.H[0].sum()) / m0 - 1.0))
