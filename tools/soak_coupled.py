"""BASELINE config 5 as a run: dynamics + column thermodynamics for `steps` model steps of 120 s (720 = one day) on an
n x n box, with the device-side forcing providers -- cyclone wind moving with model time (nsdg_boxtest_forcing), winter
thermodynamic forcing with a diurnal short-wave cycle (nsdg_column_forcing) and the column wind speed taken from the
dynamics' wind (nsdg_column_wind) -- on the native row-block driver.  Prints ranges every `every` steps and the wall time.

Concentration overshoot: neither process that changes A has a cap in the reference or here -- the column step's Hibler
freeze (HiblerConcentration.cpp:32-38: del_c = newice / h0, no limit at 1) and the unlimited DG2 transport (convergent flow
piles concentration up; no limiter exists in the scheme).  The run therefore books, per step and on the device, the
EXCESS E = sum over elements of max(A - 1, 0) (cell means) before the column step, after it and after the transport, and
reports which of the two produced it (cumulative, and over the last reporting interval), with the maxima of A after each.
usage: python tools/soak_coupled.py [steps=720] [n=512] [forcing=winter|dummy|host]

NSDG_SOAK_DIAG=first,last[,every]: per-step device-side dump for the steps first..last (round 4: where and why a run leaves the
physical range): the node of the largest speed, and around it the cell means and Gauss-point extrema of H and A, the ice
strength, the nodal thickness that enters the mass, the strain rate; plus domain-wide extrema and the change of the velocity
over the last kernel pass of the sub-cycle (how far from converged it is)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nextsimdg_amd import abi, rowblock, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 720
nx = ny = int(sys.argv[2]) if len(sys.argv) > 2 else 512
forcing = sys.argv[3] if len(sys.argv) > 3 else "winter"
L, dt, nsub = 512e3, float(os.environ.get("NSDG_SOAK_DT", "120")), int(os.environ.get("NSDG_SOAK_NSUB", "120"))  # NSDG_SOAK_DT: model time step (experiments on the strength / concentration coupling); NSDG_SOAK_NSUB: sub-iterations per step (how far the sub-cycle is from converged)
# NSDG_SOAK_SUBCYCLE: how the sub-cycle satisfies its stability bound (synthetic.BoxTest.subcycle_parameters = nsdg_mevp_stable_params):
# "adaptive" (default, the hosts' policy since round 6: local alpha / beta, Delta_min 2e-9), "keep_alpha" (round 5: alpha = beta = 1500
# and the Delta_min the mesh needs for it), "keep_delta_min" (rounds 1-4: Delta_min 2e-9 and the alpha of the bound).
# NSDG_SOAK_DELTA_MIN: another regularisation than 2e-9.  NSDG_AEVP_C / NSDG_AEVP_ALPHA_MIN: the adaptive form's constants.
delta_min = float(os.environ["NSDG_SOAK_DELTA_MIN"]) if os.environ.get("NSDG_SOAK_DELTA_MIN") else None
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
bt = synthetic.BoxTest(nx, ny, L)
sub = bt.subcycle_parameters(dt, mode=os.environ.get("NSDG_SOAK_SUBCYCLE", "adaptive"), delta_min=delta_min)
if sub["aevp_c"] > 0:
    sub["aevp_c"] = float(os.environ.get("NSDG_AEVP_C", sub["aevp_c"]))
    sub["aevp_alpha_min"] = float(os.environ.get("NSDG_AEVP_ALPHA_MIN", sub["aevp_alpha_min"]))
delta_min = sub["delta_min"]
alpha = sub["alpha"] * float(os.environ.get("NSDG_ALPHA_SCALE", "1"))  # NSDG_ALPHA_SCALE: a wider stability margin than the default 2.4 x the bound
# NSDG_SOAK_CLOSURE: 1 (default) the closure of the product -- ridging cap + scaling limiter in the transport, free drift at ice-free
# nodes; 0 the bare scheme of rounds 1-4; "transport" / "nodes": only one of the two halves (which one a run needs)
mode = os.environ.get("NSDG_SOAK_CLOSURE", "1")
rule = {} if mode in ("1", "nodes") else dict(min_conc=0.0, min_thick=0.0)
pm = ctx.mevp_default_params(alpha=alpha, beta=alpha, delta_min=delta_min, aevp_c=sub["aevp_c"], aevp_alpha_min=sub["aevp_alpha_min"], **rule)
ctx.set_mevp_params(pm)
if sub["aevp_c"] > 0:
    print("nsub %d, adaptive alpha / beta (c = %.2f, alpha_min = %.0f), Delta_min %.1e (creep below %.3g %% per day)" % (
        nsub, sub["aevp_c"], sub["aevp_alpha_min"], delta_min, abi.creep_percent_per_day(pm)), flush=True)
else:
    print("nsub %d, alpha = beta = %.0f, Delta_min %.1e (creep below %.3g %% per day): the sub-cycle relaxes %.1f %% of the way per model step" % (
        nsub, alpha, delta_min, abi.creep_percent_per_day(pm), 100.0 * min(1.0, nsub / alpha)), flush=True)
blk = rowblock.RowBlock(nx, ny, 0, 1)
core = rowblock.CoupledCore(ctx, blk, L / nx, L / ny, dt, nsub, dev, native=True, forcing=None if forcing == "host" else forcing,
                            closure=mode in ("1", "transport"))
print("closure: %s (transport cap + limiter %s, ice-free-node rule %s)" % (mode, core.closure, not rule), flush=True)
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
diag = [int(x) for x in os.environ.get("NSDG_SOAK_DIAG", "").split(",") if x]
if diag:
    import numpy as np

    g = 0.5 * np.sqrt(0.6)
    gp = [(x, y) for y in (-g, 0.0, g) for x in (-g, 0.0, g)]
    psi = lambda x, y: (1.0, x, y, x * x - 1.0 / 12.0, y * y - 1.0 / 12.0, x * y)
    GP = torch.tensor([psi(x, y) for (x, y) in gp], dtype=torch.float64, device=dev)  # [9, 6]: DG2 basis at the 3 x 3 Gauss points


def dump(step):
    """where the speed is largest, and what the scheme sees there"""
    hx, hy = L / nx, L / ny
    sp = core.u * core.u + core.v * core.v
    k = int(torch.argmax(sp))
    gy, gx = divmod(k, 2 * nx + 1)
    iy, ix = min(gy // 2, ny - 1), min(gx // 2, nx - 1)
    y0, y1, x0, x1 = max(iy - 2, 0), min(iy + 3, ny), max(ix - 2, 0), min(ix + 3, nx)
    Hw, Aw = core.H[:, y0:y1, x0:x1], core.A[:, y0:y1, x0:x1]
    Hg, Ag = torch.einsum("qc,cyx->qyx", GP, Hw), torch.einsum("qc,cyx->qyx", GP, Aw)
    hnode = core.packed[:2 * (2 * ny + 1) * (2 * nx + 1)].view(2 * ny + 1, 2 * nx + 1, 2)[:, :, 0]  # h' = max(cgH, h_min) of the last packing: first entry of pair plane 0 (csrc/mevp_common.h, NSDG_NODAL_LAYOUT 1)
    hnode = torch.where(hnode > 1e20, hnode * 2.0 ** -100, hnode)  # ice-free nodes store it scaled by 2^100 (csrc/mevp.hip: pack_node)
    pgw = abi.untile(core.pg, nx)[:, y0:y1, x0:x1]
    # strain rate at the element centres of the window from the nodal velocities (central differences over the element)
    U, V = core.u[2 * y0:2 * y1 + 1, 2 * x0:2 * x1 + 1], core.v[2 * y0:2 * y1 + 1, 2 * x0:2 * x1 + 1]
    e11 = (U[1::2, 2::2] - U[1::2, :-2:2]) / hx
    e22 = (V[2::2, 1::2] - V[:-2:2, 1::2]) / hy
    e12 = 0.5 * ((U[2::2, 1::2] - U[:-2:2, 1::2]) / hy + (V[1::2, 2::2] - V[1::2, :-2:2]) / hx)
    delta = torch.sqrt(4e-18 + 1.25 * (e11 * e11 + e22 * e22) + 1.5 * e11 * e22 + e12 * e12)
    allH, allA = torch.einsum("qc,cyx->qyx", GP, core.H), torch.einsum("qc,cyx->qyx", GP, core.A)
    print("DIAG step %4d  umax %.4g at node (%d, %d) = element (%d, %d);  |u - u four sub-iterations earlier| max %.3g;  domain: H at Gauss points [%.4g, %.4g], "
          "A at Gauss points [%.4g, %.4g], nodal thickness h' min %.4g, ice strength max %.4g"
          % (step, float(sp.flatten()[k].sqrt()), gy, gx, iy, ix, float(torch.maximum((core.u - core.ub).abs().max(), (core.v - core.vb).abs().max())),
             float(allH.min()), float(allH.max()), float(allA.min()), float(allA.max()), float(hnode.min()), float(core.pg.max())), flush=True)
    f = lambda t: np.array2string(t.cpu().numpy(), precision=4, max_line_width=200, suppress_small=False)
    print("  window rows %d..%d, columns %d..%d (top row = highest y)" % (y0, y1 - 1, x0, x1 - 1))
    for name, t in (("H mean", Hw[0]), ("H min over Gauss points", Hg.min(0).values), ("A mean", Aw[0]), ("A min over Gauss points", Ag.min(0).values),
                    ("A max over Gauss points", Ag.max(0).values), ("ice strength max over Gauss points", pgw.max(0).values),
                    ("ice strength min over Gauss points", pgw.min(0).values), ("Delta at the element centre", delta),
                    ("divergence e11 + e22", e11 + e22), ("nodal thickness h' at the centre nodes", hnode[2 * y0 + 1:2 * y1:2, 2 * x0 + 1:2 * x1:2]),
                    ("speed at the centre nodes", sp[2 * y0 + 1:2 * y1:2, 2 * x0 + 1:2 * x1:2].sqrt())):
        print("  %s\n%s" % (name, f(t.flip(0))), flush=True)
    del allH, allA


def limit(F, lo, hi):
    """EXPERIMENT (NSDG_SOAK_LIMITER=1; torch, outside the product path): Zhang-Shu scaling limiter after the transport -- the
    higher coefficients of a DG2 field are scaled towards the cell mean until its values at the 3 x 3 Gauss points lie in [lo, hi]
    (hi = None: no upper bound); the cell mean -- the conserved quantity -- is untouched"""
    g = torch.einsum("qc,cyx->qyx", GP, F)
    mean = F[0]
    gmin, gmax = g.min(0).values, g.max(0).values
    theta = torch.ones_like(mean)
    theta = torch.minimum(theta, ((mean - lo).clamp_min(0.0) / (mean - gmin).clamp_min(1e-300)))
    if hi is not None:
        theta = torch.minimum(theta, ((hi - mean).clamp_min(0.0) / (gmax - mean).clamp_min(1e-300)))
    F[1:] *= theta.clamp(0.0, 1.0)
    return float((theta < 1.0).sum())


limiter = os.environ.get("NSDG_SOAK_LIMITER") == "1"
if limiter and not diag:
    import numpy as np

    g = 0.5 * np.sqrt(0.6)
    psi = lambda x, y: (1.0, x, y, x * x - 1.0 / 12.0, y * y - 1.0 / 12.0, x * y)
    GP = torch.tensor([psi(x, y) for y in (-g, 0.0, g) for x in (-g, 0.0, g)], dtype=torch.float64, device=dev)
limited = 0.0
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
    if limiter:
        limited = limit(core.H, 0.0, None) + limit(core.A, 0.0, 1.0)
    core.time += core.dt
    e2 = excess()
    amax["transport"] = torch.maximum(amax["transport"], core.A[0].max())
    made["column"] += e1 - e0
    made["transport"] += e2 - e1
    if diag and diag[0] <= step <= diag[1] and (step - diag[0]) % (diag[2] if len(diag) > 2 else 1) == 0:
        dump(step)
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
        nfree = int((core.packed[:2 * (2 * ny + 1) * (2 * nx + 1)].view(-1, 2)[:, 0] > 1e20).sum())
        print("           ice-free nodes (free drift) in the last packing: %d of %d" % (nfree, (2 * ny + 1) * (2 * nx + 1)), flush=True)
        tot = {k: float(v) for k, v in made.items()}
        print("           A > 1: excess sum(max(A - 1, 0)) = %.4e; produced so far by the column step (uncapped Hibler freeze) %.4e, by the DG2 "
              "transport (no limiter) %.4e; since the last report %.3e / %.3e; max A right after the column step %.6f, after the transport %.6f"
              % (float(excess()), tot["column"], tot["transport"], tot["column"] - last["column"], tot["transport"] - last["transport"],
                 float(amax["column"]), float(amax["transport"])), flush=True)
        last = tot
        if limiter:
            print("           limiter (experiment): %d elements scaled in this step" % limited, flush=True)
        if not fin or float(core.u.abs().max()) > 5.0:
            raise SystemExit("the coupled run left the physical range")
torch.cuda.synchronize()
ctx.synchronize()  # also the status of the pipeline's bounded waits: raises if one gave up (the fields would be wrong)
wall = time.perf_counter() - t0
print("%d steps of %d x %d (%.1f model hours) in %.1f s wall: %.1f ms per step, %.3g element-steps/s; ice volume change %.3e (thermodynamic growth)"
      % (steps, nx, ny, steps * dt / 3600.0, wall, 1e3 * wall / steps, nx * ny * steps / wall, float(core.H[0].sum()) / m0 - 1.0))
