"""How far 120 sub-iterations bring the sub-cycle towards the implicit viscous-plastic step it iterates for, under the three ways of satisfying
its stability bound (nsdg_mevp_stable_params: adaptive alpha / beta; uniform alpha = 1500 with the mesh's Delta_min; uniform alpha of the
bound at Delta_min = 2e-9) -- on the device, the momentum equation of the box test on an n x n mesh of the 512 km domain (A0 = 0.9, H0 = 0.3 held
fixed: no transport), spun up from rest for `NSDG_CONV_WARM` (default 300) model steps of 120 sub-iterations each under the policy, then ONE more
step measured: the velocity after its 120 sub-iterations against the velocity after `NSDG_CONV_NSUB` (default 60 000) more sub-iterations of the
SAME implicit step, relative to the latter's maximum.  (From rest nothing deforms yet, every policy then runs the rigid limit's alpha: the first
step says nothing; NSDG_CONV_WARM=0 shows it.)
usage (on the GPU box): python tools/subcycle_convergence.py [n=1024]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch

from nextsimdg_amd import abi, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
long_nsub = int(os.environ.get("NSDG_CONV_NSUB", "60000"))
warm = int(os.environ.get("NSDG_CONV_WARM", "300"))
dev = torch.device("cuda:0")
L, dt = 512e3, 120.0
bt = synthetic.BoxTest(n, n, L)
H, A = np.zeros((6, n, n)), np.zeros((6, n, n))
H[0], A[0] = 0.3, 0.9
put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dH, dA = put(H), put(A)
uo, vo = [put(a) for a in bt.ocean()]
ua, va = [put(a) for a in bt.wind(0.0)]
shape = (2 * n + 1, 2 * n + 1)
z = lambda: torch.zeros(shape, dtype=torch.float64, device=dev)
converged = {}
print("%d x %d (h = %.0f m), dt = %.0f s, cyclone wind of t = 0, %d model steps of spin-up from rest, then the measured step; long run: %d sub-iterations" % (
    n, n, L / n, dt, warm, long_nsub), flush=True)
for mode in ("adaptive", "adaptive_converged", "keep_alpha", "keep_delta_min"):
    ctx = abi.Context(dev)
    sub = bt.subcycle_parameters(dt, mode=mode)
    pm = ctx.mevp_default_params(**sub)
    ctx.set_mevp_params(pm)
    ctx.set_grid(n, n, bt.hx, bt.hy)
    cgh, cga, tax, tay = z(), z(), z(), z()
    ctx.dg_to_cg(dH, cgh)
    ctx.dg_to_cg(dA, cga)
    ctx.wind_stress(ua, va, tax, tay)
    pg = ctx.private_zeros(9, n, n, dev)
    ctx.ice_strength(dH, dA, pg)
    u, v = z(), z()
    s = [ctx.private_zeros(8, n, n, dev) for _ in range(3)]
    scratch = torch.zeros(10 * u.numel() + 3 * s[0].numel(), dtype=torch.float64, device=dev)
    u0, v0 = z(), z()
    for _ in range(warm):  # spin-up: the velocity of a step's end is the next step's start, the stress is carried over
        ctx.mevp_subcycle(dt, 120, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
        u0.copy_(u)
        v0.copy_(v)
    ctx.synchronize()
    ctx.mevp_subcycle(dt, 120, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
    step_change = float(torch.maximum((u - u0).abs().max(), (v - v0).abs().max()))
    u120, v120 = u.clone(), v.clone()
    s120 = s[0].clone()
    # more sub-iterations of the SAME implicit step (u0 = the velocity at the start of the step = rest): chunks keep the watchdog fed
    done, last = 120, None
    while done < 120 + long_nsub:
        k = min(12000, 120 + long_nsub - done)
        prev = u.clone()
        ctx.mevp_subcycle(dt, k, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
        done += k
        last = float((u - prev).abs().max())
    ctx.synchronize()
    scale = float(torch.maximum(u.abs().max(), v.abs().max()))
    d120 = float(torch.maximum((u120 - u).abs().max(), (v120 - v).abs().max()))
    l2 = float(torch.sqrt(((u120 - u) ** 2 + (v120 - v) ** 2).sum() / (u ** 2 + v ** 2).sum()))
    sscale = float(s[0].abs().max())
    ds = float((s120 - s[0]).abs().max())
    what = ("adaptive alpha / beta (c = %.2f, alpha_min = %.0f)" % (sub["aevp_c"], sub["aevp_alpha_min"])) if sub["aevp_c"] > 0 else "uniform alpha = beta = %.0f" % sub["alpha"]
    print("%-15s %s, Delta_min %.2e (creep below %.3g %% per day)" % (mode, what, sub["delta_min"], abi.creep_percent_per_day(pm)))
    print("    after 120 sub-iterations: |u - u_long| max %.3e = %.1f %% of max |u_long| = %.4g m/s; relative L2 %.3f; s11: %.1f %% of its maximum away" % (
        d120, 100 * d120 / scale, scale, l2, 100 * ds / sscale))
    print("    the long run still moved %.2e m/s in its last %d sub-iterations (%.2e of the maximum); the measured step itself changed the velocity by %.2e m/s" % (
        last, min(12000, long_nsub), last / scale, step_change), flush=True)
    converged[mode] = (u.clone(), v.clone(), scale)
    ctx.close()
a, b = converged["adaptive"], converged["keep_delta_min"]
d = float(torch.maximum((a[0] - b[0]).abs().max(), (a[1] - b[1]).abs().max()))
print("same Delta_min, same limit: |u_long(adaptive) - u_long(keep_delta_min)| max %.3e = %.2f %% of the maximum (what is left is the uniform run's distance from ITS limit)" % (d, 100 * d / a[2]))
