"""The cases of the dynamics SELF-FIXTURE tests/golden/dyn_selfcheck_v1.{json,f64}.  TEST INFRASTRUCTURE ONLY.

"self-fixture -- not reference parity": the reference snapshot holds no DG / mEVP code, test or vector
(SURVEY.md section 0), so oracle/dyn_oracle.c IS the specification of that path.  These cases freeze its outputs on
fixed seeded inputs, so that a change made consistently to the oracle and to the kernels (a coefficient, a summation
order) can no longer pass unnoticed: tests/test_oracle_dynamics.py::test_oracle_reproduces_its_frozen_outputs holds
the oracle to the committed bytes exactly, tests/test_gpu_parity.py::test_hip_path_matches_the_frozen_oracle_outputs
holds the HIP path to them at the tolerances of the live comparisons.  tools/gen_dyn_fixtures.py writes the files.

Every case is a function: inputs from a fixed recipe (numpy default_rng seeds, analytic box-test fields) -> the
oracle's outputs as a dict of float64 arrays.  The GPU test rebuilds the same inputs through `inputs_*`.
"""
import numpy as np

import oracle_lib as O
from nextsimdg_amd import basis, synthetic

ORDERS = (0, 1, 2)


# ----------------------------------------------------------------------------------------------- DG transport
def inputs_transport(order):
    """70 x 37 (ragged in both directions), smooth velocity with in- and outflow, random coefficients, 3 SSP-RK steps"""
    nx, ny = 70, 37
    rng = np.random.default_rng(300 + order)
    hx, hy = 1.0 / nx, 0.8 / ny
    X, Y = basis.node_coords(nx, ny, 1.0, 0.8)
    u = np.ascontiguousarray(np.sin(3 * X) * np.cos(2 * Y) + 0.3)
    v = np.ascontiguousarray(np.cos(2 * X + 1) * np.sin(4 * Y) - 0.2)
    phi = rng.uniform(-1, 1, (basis.NCOEF[order], ny, nx))
    dt = 0.1 * min(hx, hy) / 1.5 / (2 * order + 1)
    return dict(nx=nx, ny=ny, hx=hx, hy=hy, u=u, v=v, phi=phi, dt=dt, nsteps=3)


def case_transport(order):
    i = inputs_transport(order)
    adv = O.prepare_advection(i["nx"], i["ny"], order, i["u"], i["v"])
    phi = i["phi"].copy()
    for _ in range(i["nsteps"]):
        O.transport_step(i["nx"], i["ny"], i["hx"], i["hy"], order, i["dt"], phi, adv)
    return dict(vx=adv[0], vy=adv[1], unx=adv[2], uny=adv[3], phi=phi)


# ----------------------------------------------------------------------------------------------- mEVP
class BoxInputs:
    """box-test fields on nx x ny with a perturbed concentration and thickness slopes (tests/test_gpu_parity.py Box)"""

    def __init__(self, nx, ny, seed, **params):
        self.nx, self.ny = nx, ny
        self.bt = bt = synthetic.BoxTest(nx, ny)
        self.params = params
        rng = np.random.default_rng(seed)
        H, A = bt.dg_fields()
        A[0] -= 0.2 * rng.random((ny, nx))
        H[1:] += 0.01 * rng.standard_normal(H[1:].shape)
        self.H, self.A = H, A
        self.uo, self.vo = [np.ascontiguousarray(a) for a in bt.ocean()]
        self.ua, self.va = [np.ascontiguousarray(a) for a in bt.wind(0.0)]
        self.rng = rng

    def derived(self):
        po = O.mevp_params(**self.params)
        pg = O.ice_strength(self.nx, self.ny, po, self.H, self.A)
        cgh, cga = O.dg_to_cg(self.nx, self.ny, self.H), O.dg_to_cg(self.nx, self.ny, self.A)
        tax, tay = O.wind_stress(po, self.ua, self.va)
        return po, pg, cgh, cga, tax, tay


def inputs_mevp_single():
    """one sub-iteration on 67 x 21 from a random velocity and stress"""
    b = BoxInputs(67, 21, 505)
    shape = (2 * b.ny + 1, 2 * b.nx + 1)
    u, v = 0.05 * b.rng.standard_normal(shape), 0.05 * b.rng.standard_normal(shape)
    for a in (u, v):
        a[0] = a[-1] = 0
        a[:, 0] = a[:, -1] = 0
    s = [1e3 * b.rng.standard_normal((8, b.ny, b.nx)) for _ in range(3)]
    return b, u, v, s


def case_mevp_single():
    b, u, v, s = inputs_mevp_single()
    po, pg, cgh, cga, tax, tay = b.derived()
    s = [x.copy() for x in s]
    O.mevp_stress(b.nx, b.ny, 0, b.ny, b.bt.hx, b.bt.hy, po, u, v, pg, *s)
    un, vn = np.zeros_like(u), np.zeros_like(v)
    O.mevp_velocity(b.nx, b.ny, 0, b.ny, b.bt.hx, b.bt.hy, 120.0, po, s, (u, v), (un, vn), (0.9 * u, 0.9 * v), (tax, tay), (b.uo, b.vo), cgh, cga)
    return dict(pg=pg, cgh=cgh, cga=cga, tax=tax, tay=tay, s11=s[0], s12=s[1], s22=s[2], u=un, v=vn)


def inputs_mevp_cycle():
    """the 25-sub-iteration cycle on 48 x 40 from rest (alpha = beta = 300)"""
    return BoxInputs(48, 40, 606, alpha=300.0, beta=300.0), 25


def case_mevp_cycle():
    b, nsub = inputs_mevp_cycle()
    po, pg, cgh, cga, tax, tay = b.derived()
    shape = (2 * b.ny + 1, 2 * b.nx + 1)
    u, v = np.zeros(shape), np.zeros(shape)
    s = [np.zeros((8, b.ny, b.nx)) for _ in range(3)]
    O.mevp_subcycle(b.nx, b.ny, b.bt.hx, b.bt.hy, 120.0, nsub, po, s, u, v, u.copy(), v.copy(), tax, tay, b.uo, b.vo, cgh, cga, pg)
    return dict(s11=s[0], s12=s[1], s22=s[2], u=u, v=v)


# ----------------------------------------------------------------------------------------------- one coupled step
COUPLED = dict(nx=40, ny=32, nsub=9, dt=120.0, alpha=200.0, beta=200.0, wind_factor=3.0)


def inputs_coupled():
    """column thermodynamics + dynamics (mEVP sub-cycle + DG2 transport of H and A), one model step on 40 x 32"""
    c = COUPLED
    nx, ny = c["nx"], c["ny"]
    bt = synthetic.BoxTest(nx, ny)
    rng = np.random.default_rng(707)
    H, A = bt.dg_fields()
    A[0] -= 0.3 * rng.random((ny, nx))
    H[1:3] += 0.02 * rng.standard_normal((2, ny, nx))
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    cs, cf = synthetic.column_fields_smooth(nx, ny)
    return bt, H, A, uo, vo, c["wind_factor"] * ua, c["wind_factor"] * va, {**cs, **cf}


def run_coupled(ops, device, native=False):
    """the driver's own step sequence (nextsimdg_amd/rowblock.py CoupledCore) on `ops`: the oracle stand-in or the C ABI"""
    import torch

    from nextsimdg_amd import rowblock

    c = COUPLED
    bt, H, A, uo, vo, ua, va, col = inputs_coupled()
    blk = rowblock.RowBlock(c["nx"], c["ny"], 0, 1)
    # closure=False: the frozen fixture v1 holds the bare scheme of rounds 1-4 (no cap, no limiter); the closure has its own tests
    core = rowblock.CoupledCore(ops, blk, bt.hx, bt.hy, c["dt"], c["nsub"], device, native=native, closure=False)
    core.load_global(H, A, uo, vo, ua, va)
    core.load_column(col)
    core.step()
    return core


def case_coupled():
    import torch

    from oracle_ops import OracleOps

    c = COUPLED
    core = run_coupled(OracleOps(mevp_variant=1, alpha=c["alpha"], beta=c["beta"]), torch.device("cpu"))
    out = {k: getattr(core, k).numpy().copy() for k in ("H", "A", "u", "v")}
    out.update(s11=core.s[0].numpy().copy(), s12=core.s[1].numpy().copy(), s22=core.s[2].numpy().copy())
    out.update(tice0=core.col["tice0"].numpy().copy(), hsnow=core.col["hsnow"].numpy().copy())
    return out


CASES = {"transport_dg%d" % o: (lambda o=o: case_transport(o)) for o in ORDERS}
CASES.update(mevp_single=case_mevp_single, mevp_cycle=case_mevp_cycle, coupled_step=case_coupled)
