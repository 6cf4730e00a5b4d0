import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nextsimdg_amd import abi, rowblock, synthetic
nx = ny = 512
L, dt, nsub = 512e3, 120.0, 120
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
bt = synthetic.BoxTest(nx, ny, L)
alpha = bt.stable_alpha(dt)
ctx.set_mevp_params(ctx.mevp_default_params(alpha=alpha, beta=alpha))
blk = rowblock.RowBlock(nx, ny, 0, 1)
core = rowblock.CoupledCore(ctx, blk, L / nx, L / ny, dt, nsub, dev)
if len(sys.argv) > 2 and sys.argv[2] == "random":
    cs, cf, _ = synthetic.column_fields(nx * ny)
    col = {k: v.reshape(ny, nx) for k, v in {**cs, **cf}.items() if k not in ("hice", "cice")}
else:
    cs, cf = synthetic.column_fields_smooth(nx, ny, L)
    col = {**cs, **cf}
core.load_column(col)
H, A = bt.dg_fields(); uo, vo = bt.ocean(); ua, va = bt.wind(0.0)
core.load_global(H, A, uo, vo, ua, va)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    core.step()
    fin = {k: bool(torch.isfinite(getattr(core, k)).all()) for k in ("u", "v", "H", "A")}
    fin.update({k: bool(torch.isfinite(core.col[k]).all()) for k in ("hsnow", "tice0")})
    if step % 10 == 0 or not all(fin.values()) or float(core.u.abs().max()) > 5.0:
        print(step, fin, "umax %.3g Hmin %.3g Hmax %.3g Amin %.3g Amax %.3g" % (float(core.u.abs().max()), float(core.H[0].min()), float(core.H[0].max()), float(core.A[0].min()), float(core.A[0].max())), flush=True)
    if not all(fin.values()) or float(core.u.abs().max()) > 5.0:
        break
