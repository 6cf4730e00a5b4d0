"""Diagnostic (round 6): where does the adaptive sub-cycle keep moving?  Momentum equation of the box test with H, A fixed, spun up from
rest under adaptive alpha / beta; then single sub-iterations: where the velocity changes most, and the alphas (oracle, on a band of rows)
of the elements around that node in consecutive sub-iterations.  usage: python tools/diag_adaptive_noise.py [n=1024] [warm=300]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch

import oracle_lib as O
from nextsimdg_amd import abi, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
L, dt = 512e3, 120.0
bt = synthetic.BoxTest(n, n, L)
H, A = np.zeros((6, n, n)), np.zeros((6, n, n))
H[0], A[0] = 0.3, float(os.environ.get("NSDG_DIAG_A0", "0.9"))
put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dH, dA = put(H), put(A)
uo, vo = [put(a) for a in bt.ocean()]
ua, va = [put(a) for a in bt.wind(0.0)]
shape = (2 * n + 1, 2 * n + 1)
z = lambda: torch.zeros(shape, dtype=torch.float64, device=dev)
ctx = abi.Context(dev)
sub = bt.subcycle_parameters(dt, mode=os.environ.get("NSDG_DIAG_SUBCYCLE", "adaptive"))
for k in ("aevp_c", "aevp_alpha_min"):
    if os.environ.get("NSDG_" + k.upper()):
        sub[k] = float(os.environ["NSDG_" + k.upper()])
print("parameters", sub, flush=True)
ctx.set_mevp_params(ctx.mevp_default_params(**sub))
ctx.set_grid(n, n, bt.hx, bt.hy)
cgh, cga, tax, tay = z(), z(), z(), z()
ctx.dg_to_cg(dH, cgh)
ctx.dg_to_cg(dA, cga)
ctx.wind_stress(ua, va, tax, tay)
pg = ctx.private_zeros(9, n, n, dev)
ctx.ice_strength(dH, dA, pg)
u, v, u0, v0 = z(), z(), z(), z()
s = [ctx.private_zeros(8, n, n, dev) for _ in range(3)]
scratch = torch.zeros(10 * u.numel() + 3 * s[0].numel(), dtype=torch.float64, device=dev)
for k in range(warm):
    ctx.mevp_subcycle(dt, 120, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
    if k % 50 == 49 or k == warm - 1:
        d = torch.maximum((u - u0).abs(), (v - v0).abs())
        idx = int(torch.argmax(d))
        print("step %4d: umax %.4f, largest change of the step %.3e at node (gy %d, gx %d); nodes that changed by > 1e-3: %d, > 1e-4: %d" % (
            k, float(torch.maximum(u.abs().max(), v.abs().max())), float(d.flatten()[idx]), idx // shape[1], idx % shape[1], int((d > 1e-3).sum()), int((d > 1e-4).sum())), flush=True)
    u0.copy_(u)
    v0.copy_(v)
if os.environ.get("NSDG_DIAG_BRIEF"):
    ctx.synchronize()
    sys.exit(0)
# single sub-iterations of the next step
po = O.mevp_params(**sub)
pg_h = abi.untile(pg, n).cpu().numpy()
cgh_h, cga_h = cgh.cpu().numpy(), cga.cpu().numpy()
prev_u, prev_v = u.clone(), v.clone()
ctx.mevp_subcycle(dt, 1, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
d = torch.maximum((u - prev_u).abs(), (v - prev_v).abs())
idx = int(torch.argmax(d))
gy, gx = idx // shape[1], idx % shape[1]
iy, ix = min(gy // 2, n - 1), min(gx // 2, n - 1)
print("one sub-iteration: largest change %.3e at node (%d, %d) = element (iy %d, ix %d)" % (float(d.flatten()[idx]), gy, gx, iy, ix))
y0, y1, x0, x1 = max(iy - 3, 0), min(iy + 4, n), max(ix - 3, 0), min(ix + 4, n)
for it in range(6):
    uh, vh = u.cpu().numpy(), v.cpu().numpy()
    sh = [abi.untile(x, n).cpu().numpy().copy() for x in s]
    al = np.zeros((n, n))
    O.mevp_stress(n, n, y0, y1, bt.hx, bt.hy, po, uh, vh, pg_h, *sh, dt=dt, cgh=cgh_h, cga=cga_h, alpha_e=al)
    print("sub-iteration +%d: alpha_e of the elements rows %d..%d (top first), columns %d..%d; u at the node %.5f v %.5f" % (it, y0, y1 - 1, x0, x1 - 1, uh[gy, gx], vh[gy, gx]))
    print(np.array2string(al[y0:y1, x0:x1][::-1], precision=0, max_line_width=200, suppress_small=True))
    ctx.mevp_subcycle(dt, 1, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
ctx.synchronize()
