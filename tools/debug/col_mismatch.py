import sys, os
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle_lib as O
from nextsimdg_amd import abi, synthetic
n = 4096
state, forcing, newice = synthetic.column_fields(n, seed=99)
rng = np.random.default_rng(5)
forcing["mld"][rng.random(n) < 0.25] = 0.0
m=rng.random(n) < 0.1; state["hice"][m] *= 1e-12 / np.maximum(state["cice"][m], 0.1); state["hsnow"][m] = 0.1*state["hice"][m]; state["cice"][m] = 1e-12
m=rng.random(n) < 0.05; state["hice"][m] *= 1e-13 / np.maximum(state["cice"][m], 0.1); state["hsnow"][m] = 0.1*state["hice"][m]; state["cice"][m] = 1e-13
state["hice"][rng.random(n) < 0.05] = 5e-324
pass
forcing["wind"][rng.random(n) < 0.3] = 0.0
inp = {k: v.copy() for k, v in {**state, **forcing}.items()}
ctx = abi.Context(torch.device("cuda:0"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ds, df, dn = {k: dev(v) for k, v in state.items()}, {k: dev(v) for k, v in forcing.items()}, dev(newice)
diag = torch.zeros(abi.NDIAG, n, dtype=torch.float64, device="cuda")
with np.errstate(all="ignore"):
    want = O.column_step(O.column_params(), 600.0, state, forcing, newice, want_diag=True)
ctx.column_step(600.0, ds, df, dn, diag)
g = ds["hice"].cpu().numpy()
bad = np.where(np.isfinite(state["hice"]) & (np.abs(g - state["hice"]) > 1e-9))[0]
print(len(bad), "mismatches")
d = diag.cpu().numpy()
for i in bad[:6]:
    print(i, {k: inp[k][i] for k in inp})
    print("   want", {k: want[k][i] for k in ("hi", "hs", "cnew", "qia", "qio", "qow", "dqdt")}, state["hice"][i], newice[i])
    print("   got ", {k: d[j][i] for j, k in enumerate(abi.DIAG) if k in ("hi", "hs", "cnew", "qia", "qio", "qow", "dqdt")}, g[i], float(dn[i]))
