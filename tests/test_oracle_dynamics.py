"""Analytic / self-consistency checks of the dynamics oracle (oracle/dyn_oracle.c).

PARITY UNPINNED: the reference snapshot contains no DG transport or mEVP implementation, test or
fixture, so these tests pin the oracle to the mathematics it restates rather than
to reference outputs."""
import os

import numpy as np
import pytest

import oracle_lib as O
from nextsimdg_amd import basis, synthetic


def const_velocity(nx, ny, ux, vy):
    u = np.full((2 * ny + 1, 2 * nx + 1), ux)
    v = np.full((2 * ny + 1, 2 * nx + 1), vy)
    return u, v


@pytest.mark.parametrize("order", [0, 1, 2])
def test_transport_constant_state_in_uniform_flow(order):
    nx, ny = 12, 10
    u, v = const_velocity(nx, ny, 1.0, 0.5)
    adv = O.prepare_advection(nx, ny, order, u, v)
    nc = O.ncoef(order)
    phi = np.zeros((nc, ny, nx))
    phi[0] = 1.0
    O.transport_step(nx, ny, 0.1, 0.1, order, 0.005, phi, adv)
    # away from the inflow boundaries (left, bottom; one cell per RK stage) the constant is
    # reproduced to round-off
    k = order + 1
    assert np.max(np.abs(phi[0, k:, k:] - 1.0)) < 1e-14
    if nc > 1:
        assert np.max(np.abs(phi[1:, k:, k:])) < 1e-13
    # zero-inflow boundary: the first column/row loses mass
    assert np.all(phi[0, :, 0] < 1.0) and np.all(phi[0, 0, :] < 1.0)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_transport_mass_conservation_closed_domain(order):
    nx, ny = 24, 20
    phi, u, v, _ = synthetic.rotating_patch(nx, ny, order)
    adv = O.prepare_advection(nx, ny, order, u, v)
    m0 = basis.mass_total(phi, 1 / nx, 1 / ny)
    dt = 0.1 / (2 * order + 1) * (1 / nx) / np.pi
    for _ in range(20):
        O.transport_step(nx, ny, 1 / nx, 1 / ny, order, dt, phi, adv)
    m1 = basis.mass_total(phi, 1 / nx, 1 / ny)
    assert abs(m1 - m0) <= 1e-13 * abs(m0)


def test_transport_dg0_upwind_monotone():
    nx, ny = 20, 20
    phi, u, v, _ = synthetic.rotating_patch(nx, ny, 0, kind="cosbell")
    adv = O.prepare_advection(nx, ny, 0, u, v)
    lo, hi = phi.min(), phi.max()
    dt = 0.2 * (1 / nx) / np.pi
    for _ in range(30):
        O.transport_step(nx, ny, 1 / nx, 1 / ny, 0, dt, phi, adv)
    assert phi.min() >= lo - 1e-12 and phi.max() <= hi + 1e-12


def rotate_error(n, order, frac=0.25):
    phi, u, v, phi0 = synthetic.rotating_patch(n, n, order)
    adv = O.prepare_advection(n, n, order, u, v)
    T = frac
    dt0 = 0.15 / (2 * order + 1) * (1 / n) / np.pi
    steps = int(np.ceil(T / dt0))
    dt = T / steps
    for _ in range(steps):
        O.transport_step(n, n, 1 / n, 1 / n, order, dt, phi, adv)
    th = 2 * np.pi * T

    def exact(x, y):  # rigid rotation by th about the centre (the patch stays inside the uncut core)
        xr = 0.5 + np.cos(th) * (x - 0.5) + np.sin(th) * (y - 0.5)
        yr = 0.5 - np.sin(th) * (x - 0.5) + np.cos(th) * (y - 0.5)
        return phi0(xr, yr)

    return basis.l2_error(phi, exact, 1.0, 1.0)


def test_transport_convergence_orders():
    # quarter revolution of the smooth bump; expected L2 orders ~ p+1 (a bit less on coarse grids)
    e1 = [rotate_error(n, 1) for n in (16, 32)]
    e2 = [rotate_error(n, 2) for n in (16, 32)]
    r1 = np.log2(e1[0] / e1[1])
    r2 = np.log2(e2[0] / e2[1])
    assert r1 > 1.6, (e1, r1)
    assert r2 > 2.5, (e2, r2)
    assert e2[1] < e1[1]


def box(nx=16, ny=12):
    bt = synthetic.BoxTest(nx, ny)
    p = O.mevp_params()
    H, A = bt.dg_fields()
    return bt, p, H, A


def test_dg_to_cg_reproduces_linear_field():
    nx, ny = 7, 5
    f = basis.project_dg(lambda x, y: 2 + 3 * x - y, nx, ny, 1.0, 1.0, 6)
    g = O.dg_to_cg(nx, ny, f)
    X, Y = basis.node_coords(nx, ny, 1.0, 1.0)
    assert np.max(np.abs(g - (2 + 3 * X - Y))) < 1e-13


def test_strain_of_linear_velocity_and_rigid_motion():
    nx, ny = 6, 5
    hx, hy = 2.0, 3.0
    X, Y = basis.node_coords(nx, ny, nx * hx, ny * hy)
    p = O.mevp_params(alpha=1.0)  # alpha = 1: the new stress is exactly the projected sigma(v)
    pg = np.ones((9, ny, nx))
    a, b, c, d = 1e-6, -2e-6, 3e-6, 0.5e-6
    u = np.ascontiguousarray(a * X + b * Y + 0.3)
    v = np.ascontiguousarray(c * X + d * Y - 0.1)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    O.mevp_stress(nx, ny, 0, ny, hx, hy, p, u, v, pg, *s)
    e11, e22, e12 = a, d, 0.5 * (b + c)
    delta = np.sqrt(p.delta_min ** 2 + 1.25 * (e11 ** 2 + e22 ** 2) + 1.5 * e11 * e22 + e12 ** 2)
    want11 = (0.625 * e11 + 0.375 * e22) / delta - 0.5
    want22 = (0.625 * e22 + 0.375 * e11) / delta - 0.5
    want12 = 0.25 * e12 / delta
    for S, w in zip(s, (want11, want12, want22)):
        assert np.max(np.abs(S[0] - w)) < 1e-10
        assert np.max(np.abs(S[1:])) < 1e-9
    # rigid translation: zero strain => sigma = -P/2 exactly
    u[:] = 0.7
    v[:] = -0.2
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    O.mevp_stress(nx, ny, 0, ny, hx, hy, p, u, v, pg, *s)
    # (round-off strain ~1e-17 divided by delta_min = 2e-9)
    assert np.max(np.abs(s[0][0] + 0.5)) < 1e-7 and np.max(np.abs(s[2][0] + 0.5)) < 1e-7
    assert np.max(np.abs(s[1])) < 1e-7


def test_stress_relaxation_rate_at_rest():
    # v = 0 => sigma(v) = -P/2; S^p = (1-1/alpha) S^{p-1} + (1/alpha)(-P/2)
    bt, p, H, A = box()
    nx, ny = bt.nx, bt.ny
    pg = O.ice_strength(nx, ny, p, H, A)
    u = np.zeros((2 * ny + 1, 2 * nx + 1))
    v = np.zeros_like(u)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    O.mevp_stress(nx, ny, 0, ny, bt.hx, bt.hy, p, u, v, pg, *s)
    first = s[0].copy()
    O.mevp_stress(nx, ny, 0, ny, bt.hx, bt.hy, p, u, v, pg, *s)
    np.testing.assert_allclose(s[0], first * (1 + (1 - 1 / p.alpha)), rtol=1e-12, atol=1e-9)
    assert np.max(np.abs(s[1])) == 0.0


def test_uniform_stress_has_no_divergence_and_ice_stays_at_rest():
    nx, ny = 8, 6
    bt = synthetic.BoxTest(nx, ny)
    p = O.mevp_params()
    N = (2 * ny + 1, 2 * nx + 1)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    s[0][0] = -5.0
    s[2][0] = -7.0
    s[1][0] = 2.0
    z = np.zeros(N)
    un, vn = np.ones(N), np.ones(N)
    O.mevp_velocity(nx, ny, 0, ny, bt.hx, bt.hy, 120.0, p, s, (z, z), (un, vn), (z, z), (z, z), (z, z),
                    np.full(N, 0.3), np.ones(N))
    assert np.max(np.abs(un)) < 1e-18 and np.max(np.abs(vn)) < 1e-18


def run_box(nx, ny, nsub, wind_t=0.0, flip=False, alpha=1500.0):
    bt = synthetic.BoxTest(nx, ny)
    p = O.mevp_params(alpha=alpha, beta=alpha)
    H, A = bt.dg_fields()
    if flip:
        H = np.ascontiguousarray(H[:, :, :])
    pg = O.ice_strength(nx, ny, p, H, A)
    cgh, cga = O.dg_to_cg(nx, ny, H), O.dg_to_cg(nx, ny, A)
    uo, vo = [np.ascontiguousarray(a) for a in bt.ocean()]
    ua, va = [np.ascontiguousarray(a) for a in bt.wind(wind_t)]
    tax, tay = O.wind_stress(p, ua, va)
    u, v = np.zeros_like(uo), np.zeros_like(uo)
    u0, v0 = u.copy(), v.copy()
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    hist = []
    for _ in range(nsub):
        up = u.copy()
        O.mevp_subcycle(nx, ny, bt.hx, bt.hy, 120.0, 1, p, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg)
        hist.append(np.max(np.abs(u - up)))
    return u, v, s, hist


def test_mevp_subcycle_converges_and_respects_dirichlet():
    u, v, s, hist = run_box(16, 16, 250, alpha=300.0)
    assert np.all(np.isfinite(u)) and np.all(np.isfinite(v))
    for a in (u, v):
        assert np.all(a[0] == 0) and np.all(a[-1] == 0) and np.all(a[:, 0] == 0) and np.all(a[:, -1] == 0)
    assert np.max(np.abs(u)) > 1e-4  # the wind moves the ice
    assert np.max(np.abs(u)) < 1.0
    # the pseudo-time iteration contracts: the update size decays (slowly: mEVP needs O(alpha) sweeps)
    assert hist[-1] < 0.35 * hist[0]


def test_mevp_point_symmetry():
    # with a uniform ice cover the box set-up (cyclone centred in the box, circular current) is
    # invariant under the point reflection x -> L - x, y -> L - y with v -> -v
    nx = ny = 10
    bt = synthetic.BoxTest(nx, ny)
    p = O.mevp_params()
    H = np.zeros((6, ny, nx)); H[0] = 0.3
    A = np.zeros((6, ny, nx)); A[0] = 0.9
    pg = O.ice_strength(nx, ny, p, H, A)
    cgh, cga = O.dg_to_cg(nx, ny, H), O.dg_to_cg(nx, ny, A)
    uo, vo = [np.ascontiguousarray(a) for a in bt.ocean()]
    ua, va = [np.ascontiguousarray(a) for a in bt.wind(0.0)]
    tax, tay = O.wind_stress(p, ua, va)
    u, v = np.zeros_like(uo), np.zeros_like(uo)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    O.mevp_subcycle(nx, ny, bt.hx, bt.hy, 120.0, 25, p, s, u, v, u.copy(), v.copy(), tax, tay, uo, vo, cgh, cga, pg)
    assert np.max(np.abs(u)) > 1e-5
    np.testing.assert_allclose(u, -u[::-1, ::-1], rtol=0, atol=1e-12 * np.max(np.abs(u)))
    np.testing.assert_allclose(v, -v[::-1, ::-1], rtol=0, atol=1e-12 * np.max(np.abs(v)))


@pytest.mark.parametrize("strain", [(2e-6, -1e-6, 0.5e-6), (-3e-6, -1e-6, 0.0), (1e-6, 1e-6, 2e-6), (0.0, 0.0, 1.5e-6)])
def test_converged_stress_lies_on_hiblers_elliptic_yield_curve(strain):
    """A uniform strain rate far above Delta_min on a uniform cover: the relaxed stress is uniform and must be the viscous-plastic
    stress of Hibler (1979) -- checked through what the LITERATURE says about it, not through the formula either restatement uses:
    the principal stresses lie on the ellipse ((s1 + s2) / P + 1)^2 + e^2 ((s1 - s2) / P)^2 = 1 with e = 2, the stress is coaxial
    with the strain rate and obeys the normal flow rule (strain rate parallel to the gradient of the yield function).  Oracle and
    independent restatement both."""
    import dyn_independent as I

    e11, e22, e12 = strain
    nx, ny, hx, hy = 4, 3, 500.0, 400.0
    p = O.mevp_params(alpha=2.0, delta_min=2e-9)
    H = np.zeros((6, ny, nx)); H[0] = 0.8
    A = np.zeros((6, ny, nx)); A[0] = 0.93
    pg = O.ice_strength(nx, ny, p, H, A)
    P = float(pg[0, 0, 0])
    assert np.ptp(pg) < 1e-12 * P and abs(P - p.pstar * 0.8 * np.exp(-p.compaction * 0.07)) < 1e-9 * P
    X, Y = basis.node_coords(nx, ny, nx * hx, ny * hy)
    u = np.ascontiguousarray(e11 * X + e12 * Y)  # du/dx = e11, (du/dy + dv/dx) / 2 = e12, dv/dy = e22
    v = np.ascontiguousarray(e12 * X + e22 * Y)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    for _ in range(80):  # alpha = 2: the relaxation halves the distance to sigma(u, v) per sweep
        O.mevp_stress(nx, ny, 0, ny, hx, hy, p, u, v, pg, *s)
    par = dict(alpha=2.0, delta_min=2e-9)
    si = [np.zeros((8, ny, nx)) for _ in range(3)]
    for _ in range(60):
        si = I.mevp_stress(par, hx, hy, u, v, pg, si)
    for name, S in (("oracle", s), ("independent restatement", si)):
        for c in range(3):  # uniform: only the cell mean (coefficient 0) is non-zero, the same in every element
            assert np.max(np.abs(S[c][1:])) < 1e-9 * P, name
            assert np.ptp(S[c][0]) < 1e-9 * P, name
        s11, s12, s22 = (float(S[c][0, 1, 1]) for c in range(3))
        # principal stresses and the ellipse
        mean, dev = 0.5 * (s11 + s22), np.hypot(0.5 * (s11 - s22), s12)
        s1, s2 = mean + dev, mean - dev
        assert abs(((s1 + s2) / P + 1.0) ** 2 + 4.0 * ((s1 - s2) / P) ** 2 - 1.0) < 1e-5, name  # Delta_min / Delta ~ 1e-3: squared
        # coaxial with the strain rate: the deviators are parallel
        d_eps = np.array([0.5 * (e11 - e22), e12])
        d_sig = np.array([0.5 * (s11 - s22), s12])
        assert abs(d_eps[0] * d_sig[1] - d_eps[1] * d_sig[0]) < 1e-9 * np.linalg.norm(d_eps) * P, name
        assert d_eps @ d_sig >= 0.0, name
        # normal flow rule: with F = (sI / P + 1)^2 + e^2 (sII / P)^2 - 1 in the invariants sI = s1 + s2, sII = s1 - s2, the strain
        # rate invariants (eI, eII) = (e11 + e22, |principal difference|) are parallel to (dF/dsI, dF/dsII)
        eI, eII = e11 + e22, 2.0 * np.linalg.norm(d_eps)
        gI, gII = 2.0 * ((s1 + s2) / P + 1.0), 8.0 * (s1 - s2) / P
        assert abs(eI * gII - eII * gI) < 2e-5 * np.hypot(eI, eII) * np.hypot(gI, gII), name


def free_drift_case(nx=6, ny=5):
    """uniform wind, uniform ocean current, uniform cover WITHOUT strength: every interior node is on its own (shared with the GPU twin)"""
    hx, hy = 500.0, 400.0
    shape = (2 * ny + 1, 2 * nx + 1)
    ua, va = np.full(shape, 9.0), np.full(shape, -4.0)
    uo, vo = np.full(shape, 0.05), np.full(shape, 0.02)
    cgh, cga = np.full(shape, 0.7), np.full(shape, 0.85)
    return hx, hy, ua, va, uo, vo, cgh, cga


def free_drift_solution(p, ua, va, uo, vo, h, a):
    """the steady momentum balance of the literature (Hibler 1979; Mehlmann & Richter 2017, eq. 1, with the sea-surface tilt written
    through the geostrophic current), solved by scipy -- no code of either restatement:
        0 = A tau_a + A C_w rho_w |u_o - u| (u_o - u) - rho_i h f k x (u - u_o)"""
    from scipy.optimize import fsolve

    wind = np.hypot(ua, va)
    tax, tay = p.c_atm * p.rho_atm * wind * ua, p.c_atm * p.rho_atm * wind * va

    def balance(w):
        du, dv = uo - w[0], vo - w[1]
        drag = a * p.c_ocean * p.rho_ocean * np.hypot(du, dv)
        m = p.rho_ice * h
        return [a * tax + drag * du + m * p.fc * (w[1] - vo), a * tay + drag * dv - m * p.fc * (w[0] - uo)]

    sol = fsolve(balance, [uo, vo], xtol=1e-14)
    assert np.max(np.abs(balance(sol))) < 1e-12
    return sol


def test_strengthless_cover_reaches_the_free_drift_of_the_literature():
    """P* = 0: no stress, every interior node integrates its own momentum balance; after enough model steps (implicit Euler in u0, time
    scale rho h / drag ~ 16 min) the velocity must be the STEADY free drift: ~2 % of the wind speed, turned to the right of the wind
    on the northern hemisphere (Nansen's rule), and equal to scipy's solution of the published balance.  (The steady state does not
    depend on how far the sub-cycle converges inside a step: u = u0 = fixed point of the sweep is the balance itself.)"""
    nx, ny = 6, 5
    hx, hy, ua, va, uo, vo, cgh, cga = free_drift_case(nx, ny)
    p = O.mevp_params(pstar=0.0, alpha=5.0, beta=5.0)
    tax, tay = O.wind_stress(p, ua, va)
    pg = np.zeros((9, ny, nx))
    u, v = np.zeros_like(ua), np.zeros_like(ua)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    for _ in range(70):
        O.mevp_subcycle(nx, ny, hx, hy, 600.0, 30, p, s, u, v, u.copy(), v.copy(), tax, tay, uo, vo, cgh, cga, pg)
    assert all(np.max(np.abs(x)) < 1e-12 for x in s)  # -P/2 = 0: no stress at all
    want = free_drift_solution(p, 9.0, -4.0, 0.05, 0.02, 0.7, 0.85)
    ui, vi = u[1:-1, 1:-1], v[1:-1, 1:-1]
    assert np.max(np.abs(ui - want[0])) < 1e-10 and np.max(np.abs(vi - want[1])) < 1e-10
    rel = np.array([want[0] - 0.05, want[1] - 0.02])  # drift relative to the current
    ratio = np.linalg.norm(rel) / np.hypot(9.0, -4.0)
    assert 0.01 < ratio < 0.03
    cross = 9.0 * rel[1] - (-4.0) * rel[0]  # z component of wind x drift: negative = drift to the RIGHT of the wind (f > 0)
    assert p.fc > 0 and cross < 0


def implicit_vp_case(adaptive=False):
    """a small box whose cover deforms under the wind: inputs of the fixed-point test (shared with the GPU twin); adaptive: local,
    solution-adaptive alpha and beta with the stability bound's own constant (no uniform alpha is stated at all)"""
    nx = ny = 8
    bt = synthetic.BoxTest(nx, ny, 40e3)  # 5 km elements: the stress divergence matters
    pk = dict(alpha=80.0, beta=80.0, delta_min=2e-7)
    if adaptive:
        pk = dict(delta_min=2e-7, aevp_c=(2.4 * np.pi) ** 2, aevp_alpha_min=10.0)
    p = O.mevp_params(**pk)
    H, A = bt.dg_fields()
    A[0] -= 0.15  # a cover that deforms under this wind
    c = dict(nx=nx, ny=ny, hx=bt.hx, hy=bt.hy, dt=120.0, pk=pk, p=p, H=H, A=A)
    c["pg"] = O.ice_strength(nx, ny, p, H, A)
    c["cgh"], c["cga"] = O.dg_to_cg(nx, ny, H), O.dg_to_cg(nx, ny, A)
    c["uo"], c["vo"] = [np.ascontiguousarray(a) for a in bt.ocean()]
    ua, va = [np.ascontiguousarray(a) for a in bt.wind(0.0)]
    c["tax"], c["tay"] = O.wind_stress(p, ua, va)
    rng = np.random.default_rng(3)
    c["u0"], c["v0"] = 0.02 * rng.standard_normal(c["uo"].shape), 0.02 * rng.standard_normal(c["uo"].shape)
    for a in (c["u0"], c["v0"]):
        a[0] = a[-1] = 0
        a[:, 0] = a[:, -1] = 0
    return c


def check_implicit_vp_fixed_point(c, u, v, s):
    """(u, v, s) -- a converged sub-cycle -- against ONE Picard sweep of the implicit equation written by the independent restatement"""
    import dyn_independent as I

    scale = np.max(np.hypot(u, v))
    smax = max(np.max(np.abs(x)) for x in s)
    assert scale > 1e-4 and smax > 1.0
    par = {k: getattr(c["p"], k) for k in I.PARAMS}
    par.update(alpha=1.0, beta=0.0)  # alpha = 1: S = Proj sigma(u); beta = 0: no pseudo-time term
    S = I.mevp_stress(par, c["hx"], c["hy"], u, v, c["pg"], [np.zeros_like(x) for x in s])
    for a, b in zip(S, s):
        assert np.max(np.abs(a - b)) < 1e-7 * smax
    args = (c["u0"], c["v0"], c["tax"], c["tay"], c["uo"], c["vo"], c["cgh"], c["cga"])
    un, vn = I.mevp_velocity(par, c["hx"], c["hy"], c["dt"], S, u, v, *args)
    assert np.max(np.abs(un - u)) < 1e-7 * scale and np.max(np.abs(vn - v)) < 1e-7 * scale
    # and the step is not trivial: the stress divergence is a leading term of the balance
    uf, vf = I.mevp_velocity(par, c["hx"], c["hy"], c["dt"], [np.zeros_like(x) for x in s], u, v, *args)
    assert np.max(np.hypot(uf - u, vf - v)) > 1e-2 * scale


@pytest.mark.parametrize("adaptive", [False, True])
def test_converged_subcycle_solves_the_implicit_vp_step(adaptive):
    """What defines mEVP (Bouillon et al. 2013, Kimmritz et al. 2015): the pseudo-time iteration's fixed point is the solution of the
    implicit viscous-plastic step  m (u - u0) / dt = F(u, sigma(u)),  whatever alpha and beta are.  The oracle iterates to convergence;
    the INDEPENDENT restatement then evaluates one Picard sweep of the implicit equation itself (alpha = 1, beta = 0) at that state --
    it must return the same stress and the same velocity.  adaptive (round 6): the same with local, solution-adaptive alpha and beta
    (Kimmritz et al. 2016) -- the limit does not depend on them, so the SAME check must pass unchanged; the alphas in use at the fixed
    point differ from element to element."""
    c = implicit_vp_case(adaptive)
    nx, ny = c["nx"], c["ny"]
    u, v = c["u0"].copy(), c["v0"].copy()
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    args = (c["u0"], c["v0"], c["tax"], c["tay"], c["uo"], c["vo"], c["cgh"], c["cga"], c["pg"])
    O.mevp_subcycle(nx, ny, c["hx"], c["hy"], c["dt"], 2500, c["p"], s, u, v, *args)
    up, vp = u.copy(), v.copy()
    O.mevp_subcycle(nx, ny, c["hx"], c["hy"], c["dt"], 1, c["p"], [x.copy() for x in s], up, vp, *args)
    assert np.max(np.abs(up - u)) < 1e-10 * np.max(np.hypot(u, v))  # converged
    check_implicit_vp_fixed_point(c, u, v, s)
    if adaptive:
        al = np.zeros((ny, nx))
        O.mevp_stress(nx, ny, 0, ny, c["hx"], c["hy"], c["p"], u, v, c["pg"], *[x.copy() for x in s], dt=c["dt"], cgh=c["cgh"], cga=c["cga"], alpha_e=al)
        assert float(al.min()) >= 10.0 and float(al.max()) > float(al.min()) + 3.0 and float(al.max()) < 80.0, (al.min(), al.max())  # local values, all below the uniform run's 80


# ------------------------------------------------------------------------------------ frozen outputs (self-fixture)
def test_oracle_reproduces_its_frozen_outputs():
    """SELF-FIXTURE -- NOT reference parity (the reference has no DG / mEVP code).  The oracle is the
    specification of the dynamics; tests/golden/dyn_selfcheck_v1.{json,f64} (tools/gen_dyn_fixtures.py) freeze its
    outputs for DG0/1/2 transport (3 steps, 70 x 37), one mEVP sub-iteration (67 x 21), the 25-sub-iteration cycle
    (48 x 40) and one coupled step (40 x 32).  Every array must come out bit for bit: a change of a coefficient or of
    a summation order in oracle/dyn_oracle.c fails here even if the kernels were changed the same way."""
    import hashlib
    import json
    import os

    import dyn_fixture_cases as cases

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    idx = json.load(open(os.path.join(golden, "dyn_selfcheck_v1.json")))
    data = np.fromfile(os.path.join(golden, idx["data_file"]), dtype="<f8")
    assert hashlib.sha256(data.tobytes()).hexdigest() == idx["data_sha256"]
    assert "NOT reference parity" in idx["title"]
    seen = set()
    outs = {}
    for e in idx["arrays"]:
        if e["case"] not in outs:
            outs[e["case"]] = cases.CASES[e["case"]]()
        got = np.ascontiguousarray(outs[e["case"]][e["name"]], dtype="<f8")
        n = int(np.prod(e["shape"]))
        want = data[e["offset"]:e["offset"] + n].reshape(e["shape"])
        assert list(got.shape) == e["shape"], (e["case"], e["name"])
        assert [float(x).hex() for x in want.reshape(-1)[:4]] == e["first"]  # the index and the data file belong together
        if hashlib.sha256(got.tobytes()).hexdigest() != e["sha256"]:
            diff = np.abs(got - want)
            raise AssertionError("%s/%s is not the frozen output: %d of %d values differ, largest difference %.3e (%.3e relative to the "
                                 "largest value)" % (e["case"], e["name"], int((got != want).sum()), n, diff.max(), diff.max() / max(np.abs(want).max(), 1e-300)))
        seen.add(e["case"])
    assert seen == set(cases.CASES)


# ------------------------------------------------------------------------------------ operators against independent quadrature
def _lagrange(s):
    """quadratic Lagrange functions and derivatives on [0, 1] (nodes 0, 1/2, 1) at the points s"""
    s = np.asarray(s, dtype=float)
    L = np.stack([2 * (s - 0.5) * (s - 1), -4 * s * (s - 1), 2 * s * (s - 0.5)])
    dL = np.stack([4 * s - 3, -8 * s + 4, 4 * s - 1])
    return L, dL


def test_strain_projection_is_pointwise_exact_for_biquadratic_velocities():
    """The DG8 strain coefficients are an L2 projection that loses nothing: the derivatives of a biquadratic CG2 function lie in
    the 8-space.  Checked against an independent numpy evaluation of grad w at random points of every element, through the only
    door the oracle has (the stress update): with alpha = 1, Delta_min = D huge and P = 2 D everywhere the new stress is
    Proj(1.25 e11 + 0.75 e22) - D, Proj(e12 / 2), Proj(1.25 e22 + 0.75 e11) - D."""
    nx, ny, hx, hy = 5, 4, 2.0, 3.0
    rng = np.random.default_rng(8)
    D = 1e8
    p = O.mevp_params(alpha=1.0, delta_min=D)
    u, v = rng.standard_normal((2 * ny + 1, 2 * nx + 1)), rng.standard_normal((2 * ny + 1, 2 * nx + 1))
    s = [rng.standard_normal((8, ny, nx)) for _ in range(3)]  # alpha = 1 forgets the old stress
    O.mevp_stress(nx, ny, 0, ny, hx, hy, p, u, v, np.full((9, ny, nx), 2 * D), *s)
    pts = rng.random((7, 2))
    Lx, dLx = _lagrange(pts[:, 0])
    Ly, dLy = _lagrange(pts[:, 1])
    worst12 = worst11 = 0.0
    for iy in range(ny):
        for ix in range(nx):
            ul, vl = u[2 * iy:2 * iy + 3, 2 * ix:2 * ix + 3], v[2 * iy:2 * iy + 3, 2 * ix:2 * ix + 3]  # [ay, ax]
            ux = np.einsum("yx,xp,yp->p", ul, dLx, Ly) / hx
            uy = np.einsum("yx,xp,yp->p", ul, Lx, dLy) / hy
            vx = np.einsum("yx,xp,yp->p", vl, dLx, Ly) / hx
            vy = np.einsum("yx,xp,yp->p", vl, Lx, dLy) / hy
            psi = np.array([basis.psi(i, pts[:, 0] - 0.5, pts[:, 1] - 0.5) for i in range(8)])  # [i, p]
            got12 = s[1][:, iy, ix] @ psi
            got11 = s[0][:, iy, ix] @ psi + D
            got22 = s[2][:, iy, ix] @ psi + D
            worst12 = max(worst12, np.max(np.abs(got12 - 0.25 * (uy + vx))))
            worst11 = max(worst11, np.max(np.abs(got11 - (1.25 * ux + 0.75 * vy))), np.max(np.abs(got22 - (1.25 * vy + 0.75 * ux))))
    assert worst12 < 1e-13  # values O(1): round-off only
    assert worst11 < 1e-6  # the constant D = 1e8 that was subtracted costs 8 digits


def test_nodal_divergence_is_the_weak_divergence_of_the_dg_stress():
    """-(sigma, grad phi_n) at every interior CG2 node, for a random DG8 stress, against numpy quadrature with explicitly written
    Lagrange functions (4 x 4 Gauss points: exact for the degree-5 integrands) -- and the lumped mass against int phi_n.  The
    oracle's divergence is read off the velocity update with everything else switched off: u_new = (div_x / M_n) / (rho h (1 + beta) / dt)."""
    nx, ny, hx, hy, dt = 5, 4, 2.0, 3.0, 120.0
    rng = np.random.default_rng(9)
    p = O.mevp_params(c_ocean=0.0, fc=0.0, beta=3.0)
    s = [rng.standard_normal((8, ny, nx)) for _ in range(3)]
    N = (2 * ny + 1, 2 * nx + 1)
    z, h = np.zeros(N), 0.4
    un, vn = np.zeros(N), np.zeros(N)
    O.mevp_velocity(nx, ny, 0, ny, hx, hy, dt, p, s, (z, z), (un, vn), (z, z), (z, z), (z, z), np.full(N, h), np.ones(N))
    g, w = basis.gauss(4)  # on [-1/2, 1/2]
    L, dL = _lagrange(g + 0.5)
    divx, divy, lump = np.zeros(N), np.zeros(N), np.zeros(N)
    psi = np.array([[[basis.psi(i, g[qx], g[qy]) for qx in range(4)] for qy in range(4)] for i in range(8)])  # [i, qy, qx]
    W = np.outer(w, w)  # [qy, qx]
    for iy in range(ny):
        for ix in range(nx):
            S11, S12, S22 = (np.tensordot(c[:, iy, ix], psi, axes=1) for c in s)  # values at the Gauss points [qy, qx]
            for ay in range(3):
                for ax in range(3):
                    phix = np.outer(L[ay], dL[ax]) / hx  # d phi / dx at [qy, qx]
                    phiy = np.outer(dL[ay], L[ax]) / hy
                    n = (2 * iy + ay, 2 * ix + ax)
                    divx[n] -= hx * hy * np.sum(W * (S11 * phix + S12 * phiy))
                    divy[n] -= hx * hy * np.sum(W * (S12 * phix + S22 * phiy))
                    lump[n] += hx * hy * np.sum(W * np.outer(L[ay], L[ax]))
    scale = p.rho_ice * h * (1.0 + p.beta) / dt
    inner = (slice(1, -1), slice(1, -1))
    assert np.max(np.abs(divx[inner])) > 0.1
    np.testing.assert_allclose(un[inner] * scale * lump[inner], divx[inner], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vn[inner] * scale * lump[inner], divy[inner], rtol=1e-12, atol=1e-12)
    for a in (un, vn):  # Dirichlet walls
        assert np.all(a[0] == 0) and np.all(a[-1] == 0) and np.all(a[:, 0] == 0) and np.all(a[:, -1] == 0)


def test_oracle_agrees_with_the_independent_restatement():
    """tests/dyn_independent.py restates DESIGN.md section 3 a second time, in dense numpy, from the formulas only and by
    different routes (zeta / eta form of the VP law, strain from the derivative of the biquadratic instead of the projected
    coefficients, full-mass-matrix projections, weak divergence by quadrature).  Its outputs on a 6 x 5 case are committed
    (tests/golden/dyn_independent_v3.npz, tools/gen_dyn_independent.py); the oracle must reproduce every one of them --
    ice strength, nodal means, wind stress, ONE mEVP sub-iteration (stress and velocity, with the ice-free-node rule), advection
    velocity, ONE DG2 transport stage, the closure of the transport (cap + scaling limiter) on H and A, and the same sub-iteration with
    local, solution-adaptive alpha and beta (round 6: every element's alpha, the stress, the velocity) -- to 1e-12.  NOT reference parity: the snapshot has no dynamics code (/root/reference/CMakeLists.txt:43-46).
    A fresh evaluation of the restatement must equal the committed file (the file is not an opaque blob)."""
    import dyn_independent as D

    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dyn_independent_v3.npz"))
    inp = D.case_inputs()
    for k, v in inp.items():
        v = np.stack(v) if isinstance(v, list) else v
        assert np.array_equal(fix["in_" + k], v), k
    fresh = D.case_outputs(inp)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    for k, v in fresh.items():
        assert rel(v, fix["out_" + k]) < 1e-13, ("fresh evaluation", k)
    c, nx, ny = D.CASE, D.CASE["nx"], D.CASE["ny"]
    po = O.mevp_params(**D.PARAMS)
    got = {"pg": O.ice_strength(nx, ny, po, inp["H"], inp["A"]), "cgh": O.dg_to_cg(nx, ny, inp["H"]), "cga": O.dg_to_cg(nx, ny, inp["A"])}
    got["tax"], got["tay"] = O.wind_stress(po, inp["ua"], inp["va"])
    s = [x.copy() for x in inp["S"]]
    O.mevp_stress(nx, ny, 0, ny, c["hx"], c["hy"], po, inp["u"], inp["v"], got["pg"], *s)
    got["s11"], got["s12"], got["s22"] = s
    un, vn = np.zeros_like(inp["u"]), np.zeros_like(inp["u"])
    O.mevp_velocity(nx, ny, 0, ny, c["hx"], c["hy"], c["dt"], po, s, (inp["u"], inp["v"]), (un, vn), (inp["u0"], inp["v0"]),
                    (got["tax"], got["tay"]), (inp["uo"], inp["vo"]), got["cgh"], got["cga"])
    got["u_new"], got["v_new"] = un, vn
    adv = O.prepare_advection(nx, ny, 2, inp["u"], inp["v"])
    got["vx_dg"], got["vy_dg"], got["un_x"], got["un_y"] = adv
    out = np.zeros_like(inp["phi"])
    O.transport_stage(nx, ny, 0, ny, c["hx"], c["hy"], 2, c["dt"], c["rk_a"], c["rk_b"], inp["phi0"], inp["phi"], out, adv)
    got["phi_stage"] = out
    got["H_limited"], got["A_limited"] = inp["H"].copy(), inp["A"].copy()
    O.transport_limit(nx, ny, 2, got["H_limited"], 0.0, np.inf, False)
    O.transport_limit(nx, ny, 2, got["A_limited"], 0.0, 1.0, True)
    # the sub-iteration in its adaptive form
    pa = O.mevp_params(**dict(D.PARAMS, **D.ADAPTIVE))
    sa = [x.copy() for x in inp["S"]]
    alpha_e = np.zeros((ny, nx))
    O.mevp_stress(nx, ny, 0, ny, c["hx"], c["hy"], pa, inp["u"], inp["v"], got["pg"], *sa, dt=c["dt"], cgh=got["cgh"], cga=got["cga"], alpha_e=alpha_e)
    got["ad_s11"], got["ad_s12"], got["ad_s22"], got["ad_alpha"] = sa[0], sa[1], sa[2], alpha_e
    una, vna = np.zeros_like(inp["u"]), np.zeros_like(inp["u"])
    O.mevp_velocity(nx, ny, 0, ny, c["hx"], c["hy"], c["dt"], pa, sa, (inp["u"], inp["v"]), (una, vna), (inp["u0"], inp["v0"]),
                    (got["tax"], got["tay"]), (inp["uo"], inp["vo"]), got["cgh"], got["cga"], alpha_e=alpha_e)
    got["ad_u_new"], got["ad_v_new"] = una, vna
    assert sorted(got) == sorted(fresh)
    for k, v in got.items():
        assert rel(v, fix["out_" + k]) < 1e-12, (k, rel(v, fix["out_" + k]))
    # the case exercises the clamps and the floor: concentration above 1 and thickness below 0 at Gauss points, a node thinner than h_min
    assert float(fix["out_cga"].max()) > 1.0 and float(fix["out_cgh"].min()) < D.PARAMS["h_min"] and float(np.abs(fix["out_u_new"]).max()) > 0.05
    assert float((fix["out_pg"] == 0.0).sum()) > 0  # max(h, 0) was active
    # the adaptive case has elements at the lower bound (among them the one with the ice-free centre node) and elements well above it
    al = fix["out_ad_alpha"]
    assert al[2, 3] == D.ADAPTIVE["aevp_alpha_min"] and float(al.max()) > 3 * D.ADAPTIVE["aevp_alpha_min"] and int((al == al.min()).sum()) >= 5
    # ... and the round-5 closure: ice-free nodes exist (and others do not), the cap and both sides of the limiter were active
    free = np.array([[D.ice_free(D.PARAMS, fix["out_cgh"][gy, gx], fix["out_cga"][gy, gx]) for gx in range(2 * nx + 1)] for gy in range(2 * ny + 1)])
    assert 0 < int(free[1:-1, 1:-1].sum()) < free[1:-1, 1:-1].size // 2
    assert int((inp["A"][0] > 1.0).sum()) > 0 and float(fix["out_A_limited"][0].max()) == 1.0
    assert np.array_equal(fix["out_H_limited"][0], inp["H"][0])  # the limiter never touches a cell mean
    scaled = lambda a, b: int((np.abs(a[1:] - b[1:]).max(axis=0) > 0).sum())
    assert scaled(fix["out_H_limited"], inp["H"]) > 0 and scaled(fix["out_A_limited"], inp["A"]) > 0
    assert scaled(fix["out_H_limited"], inp["H"]) < nx * ny  # ... and most elements were left alone


@pytest.mark.parametrize("order", [1, 2])
def test_oracle_limiter_properties(order):
    """oracle_transport_limit: cell means untouched (the cap aside), point values inside the bounds afterwards, elements inside the
    bounds untouched bit for bit, theta = the LARGEST admissible scaling (the binding point lands on the bound)"""
    nx, ny = 23, 17
    nc = O.ncoef(order)
    rng = np.random.default_rng(order)
    f = np.zeros((nc, ny, nx))
    f[0] = 0.5 + 0.3 * rng.standard_normal((ny, nx))
    f[1:] = 0.2 * rng.standard_normal((nc - 1, ny, nx)) * (rng.random((ny, nx)) < 0.7)
    g = [-0.5 / np.sqrt(3.0), 0.5 / np.sqrt(3.0)] if order == 1 else [-0.5 * np.sqrt(0.6), 0.0, 0.5 * np.sqrt(0.6)]
    pts = [(x, y) for y in g for x in g] + [(0.5, s) for s in g] + [(-0.5, s) for s in g] + [(s, 0.5) for s in g] + [(s, -0.5) for s in g]
    pts += [(x, y) for y in (-0.5, 0.5) for x in (-0.5, 0.5)]  # ... and the corners
    psi = lambda x, y: (1.0, x, y, x * x - 1.0 / 12.0, y * y - 1.0 / 12.0, x * y)[:nc]
    P = np.array([psi(x, y) for (x, y) in pts])
    vals = lambda a: np.einsum("pc,cyx->pyx", P, a)
    out = f.copy()
    O.transport_limit(nx, ny, order, out, 0.0, 1.0, True)
    assert np.array_equal(out[0], np.minimum(f[0], 1.0))
    v0, v1 = vals(f), vals(out)
    ok = (f[0] >= 0.0)
    assert v1[:, ok].min() >= -1e-15 and v1.max() <= 1.0 + 1e-15
    inside = (v0.min(axis=0) >= 0.0) & (v0.max(axis=0) <= 1.0)
    assert 10 < int(inside.sum()) < nx * ny - 10 and np.array_equal(out[:, inside], f[:, inside])
    scaled = ~inside & ok & (f[0] < 1.0) & (np.abs(f[1:]).max(axis=0) > 0)
    touch = np.minimum(np.abs(v1[:, scaled].min(axis=0) - 0.0), np.abs(v1[:, scaled].max(axis=0) - 1.0))
    assert int(scaled.sum()) > 10 and touch.max() <= 1e-14  # a scaled element touches one of its bounds
    # no upper bound, no cap: only the lower side acts
    out2 = f.copy()
    O.transport_limit(nx, ny, order, out2, 0.0, np.inf, False)
    assert np.array_equal(out2[0], f[0]) and vals(out2)[:, ok].min() >= -1e-15 and vals(out2).max() > 1.0
    # a row range
    out3 = f.copy()
    O.transport_limit(nx, ny, order, out3, 0.0, 1.0, True, 4, 9)
    assert np.array_equal(out3[:, 4:9], out[:, 4:9]) and np.array_equal(out3[:, :4], f[:, :4]) and np.array_equal(out3[:, 9:], f[:, 9:])
