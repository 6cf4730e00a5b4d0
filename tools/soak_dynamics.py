"""Diagnostic: one model day (720 steps of 120 s) of the dynamics core on a 512 x 512 box test, wind of the moving
cyclone re-evaluated every step; prints the ranges of the fields every 60 steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nextsimdg_amd import abi, rowblock, synthetic

nx = ny = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L, dt, nsub = 512e3, 120.0, 120
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
bt = synthetic.BoxTest(nx, ny, L)
sub = bt.subcycle_parameters(dt, mode=os.environ.get("NSDG_SOAK_SUBCYCLE", "adaptive"),  # adaptive (the hosts' policy) / keep_alpha (round 5) / keep_delta_min (rounds 1-4)
                             delta_min=float(os.environ["NSDG_SOAK_DELTA_MIN"]) if os.environ.get("NSDG_SOAK_DELTA_MIN") else None)
alpha = sub["alpha"] * float(os.environ.get("NSDG_ALPHA_SCALE", "1"))  # experiment: margin over the stability bound
pm = ctx.mevp_default_params(**dict(sub, alpha=alpha, beta=alpha))
ctx.set_mevp_params(pm)
print("sub-cycle: %s, Delta_min %.1e (creep below %.3g %% per day)" % ("adaptive alpha / beta (c = %.2f, alpha_min = %.0f)" % (sub["aevp_c"], sub["aevp_alpha_min"])
                                                                       if sub["aevp_c"] > 0 else "alpha = beta = %.0f" % alpha, sub["delta_min"], abi.creep_percent_per_day(pm)), flush=True)
core = rowblock.DynamicsCore(ctx, rowblock.RowBlock(nx, ny, 0, 1), L / nx, L / ny, dt, nsub, dev, native=True)
H, A = bt.dg_fields()
uo, vo = bt.ocean()
ua, va = bt.wind(0.0)
core.load_global(H, A, uo, vo, ua, va)
m0 = float(core.H[0].sum())
for step in range(720):
    ctx.set_grid(nx, ny, L / nx, L / ny)
    ctx.boxtest_forcing(L, step * dt, wind=(core.ua, core.va))
    core.step()
    if step % 60 == 59:
        ok = all(bool(torch.isfinite(f).all()) for f in (core.u, core.v, core.H, core.A))
        print("hour %2d finite %s umax %.3f m/s  H %.3f..%.3f  A %.3f..%.3f  mass drift %.1e" % (
            (step + 1) // 30, ok, float(core.u.abs().max()), float(core.H[0].min()), float(core.H[0].max()), float(core.A[0].min()),
            float(core.A[0].max()), float(core.H[0].sum()) / m0 - 1.0), flush=True)
        if not ok:
            break
ctx.synchronize()  # also the status of the pipeline's bounded waits: raises if one gave up (the fields would be wrong)
