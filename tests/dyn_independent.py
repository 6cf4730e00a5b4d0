"""INDEPENDENT dense restatement of the nonlinear parts of the dynamics scheme, TEST INFRASTRUCTURE.

Why it exists.  The reference snapshot holds no DG / mEVP code (/root/reference/CMakeLists.txt:43-46 comments the `dynamics`
component out; there is no dynamics/ directory, no test, no fixture), so nothing of the reference can pin
`oracle/dyn_oracle.c`: the HIP kernels are compared with that oracle, and an error made consistently in both would pass
every such comparison.  This file removes the common mode: it is written from the FORMULAS of DESIGN.md section 3 only --
it imports nothing from nextsimdg_amd (no basis tables, no generated constants), shares no code with the oracle, and takes
different routes wherever the mathematics allows one:

  * polynomials are numpy Polynomial objects (Lagrange functions built from their nodes, derivatives by .deriv());
  * the strain rate at a Gauss point is the derivative of the biquadratic velocity THERE (the scheme projects the
    derivative onto the 8-coefficient space and evaluates the projection; the space contains the derivatives, so the two
    must agree);
  * the viscous-plastic law is written with zeta, eta and the ellipse ratio e (sigma = 2 eta eps + (zeta - eta) tr(eps) I
    - P/2 I), not with the 5/8, 3/8, 1/4 the kernels use;
  * L2 projections solve with the FULL mass matrix from a 6-point quadrature (orthogonality of the basis is not assumed);
  * the ice-free-node rule (round 5) is decided on the TRUE thickness cgH / cgA where the kernels compare cgH with min_thick cgA, and
    the free-drift node simply has no stress-divergence term (the kernels weight it by 2^-100); the scaling limiter finds its
    theta as the minimum of the admissible scalings of every quadrature point, each from its own value (the kernels take the
    extrema first);
  * the nodal divergence is the weak form -(sigma, grad phi_n) integrated over each adjacent element with a 5-point rule,
    the lumped mass the integral of phi_n; the volume term of the transport is integrated with a 5-point rule as well
    (the scheme's 3-point rule is exact for it); only where the scheme's quadrature IS the definition -- the Gauss points
    at which the stress is evaluated, the 3 edge points of the upwind flux -- the same points are used.

It does NOT make parity "green" (nothing can, without a reference implementation); it says that two independently written
statements of DESIGN.md section 3 agree to round-off.  Pure-Python loops: small grids only.
"""
import numpy as np
from numpy.polynomial import Polynomial as Poly
from numpy.polynomial.legendre import leggauss

E_RATIO = 2.0  # ellipse ratio of the VP rheology


def gauss_unit(n):
    """n-point Gauss-Legendre rule on the reference interval [-1/2, 1/2]"""
    x, w = leggauss(n)
    return 0.5 * x, 0.5 * w


NODES = (-0.5, 0.0, 0.5)


def lagrange(k):
    p = Poly([1.0])
    for j in range(3):
        if j != k:
            p = p * Poly([-NODES[j], 1.0]) / (NODES[k] - NODES[j])
    return p


LAG = [lagrange(k) for k in range(3)]
DLAG = [p.deriv() for p in LAG]

# DG basis of DESIGN.md section 3 as products px(xi) * py(eta): 1, xi, eta, xi^2 - 1/12, eta^2 - 1/12, xi eta,
# eta (xi^2 - 1/12), xi (eta^2 - 1/12)
_P0, _P1, _P2 = Poly([1.0]), Poly([0.0, 1.0]), Poly([-1.0 / 12.0, 0.0, 1.0])
PSI = [(_P0, _P0), (_P1, _P0), (_P0, _P1), (_P2, _P0), (_P0, _P2), (_P1, _P1), (_P2, _P1), (_P1, _P2)]


def psi(i, x, y):
    return PSI[i][0](x) * PSI[i][1](y)


def psi_dx(i, x, y):
    return PSI[i][0].deriv()(x) * PSI[i][1](y)


def psi_dy(i, x, y):
    return PSI[i][0](x) * PSI[i][1].deriv()(y)


def mass_matrix(n):
    x, w = gauss_unit(6)
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            M[i, j] = sum(w[a] * w[b] * psi(i, x[a], x[b]) * psi(j, x[a], x[b]) for a in range(6) for b in range(6))
    return M


def dg_value(coef, x, y):
    """value of a DG function with coefficients coef[0..n) at the reference point (x, y)"""
    return sum(coef[i] * psi(i, x, y) for i in range(len(coef)))


def local_nodes(f, ix, iy):
    """the 3 x 3 nodal values of a CG2 field on element (ix, iy): [ay][ax]"""
    return f[2 * iy:2 * iy + 3, 2 * ix:2 * ix + 3]


def cg_value(loc, x, y, dx=0, dy=0):
    fx = DLAG if dx else LAG
    fy = DLAG if dy else LAG
    return sum(loc[ay, ax] * fx[ax](x) * fy[ay](y) for ay in range(3) for ax in range(3))


# ------------------------------------------------------------------------------------------------ per-step preparation
def ice_strength(par, H, A):
    """P = P* max(h, 0) exp(-C (1 - clamp(a, 0, 1))) at the 3 x 3 Gauss points of every element; result [9, ny, nx], q = 3 qy + qx"""
    _, ny, nx = H.shape
    g, _ = gauss_unit(3)
    P = np.zeros((9, ny, nx))
    for iy in range(ny):
        for ix in range(nx):
            for qy in range(3):
                for qx in range(3):
                    h = max(dg_value(H[:, iy, ix], g[qx], g[qy]), 0.0)
                    a = min(max(dg_value(A[:, iy, ix], g[qx], g[qy]), 0.0), 1.0)
                    P[3 * qy + qx, iy, ix] = par["pstar"] * h * np.exp(-par["compaction"] * (1.0 - a))
    return P


def nodal_mean(F):
    """CG2 nodal field of a DG field: at every node the mean of the values the adjacent elements take there"""
    _, ny, nx = F.shape
    out = np.zeros((2 * ny + 1, 2 * nx + 1))
    cnt = np.zeros_like(out)
    for iy in range(ny):
        for ix in range(nx):
            for ay in range(3):
                for ax in range(3):
                    out[2 * iy + ay, 2 * ix + ax] += dg_value(F[:, iy, ix], NODES[ax], NODES[ay])
                    cnt[2 * iy + ay, 2 * ix + ax] += 1
    return out / cnt


def wind_stress(par, ua, va):
    mag = np.hypot(ua, va)
    return par["c_atm"] * par["rho_atm"] * mag * ua, par["c_atm"] * par["rho_atm"] * mag * va


# ------------------------------------------------------------------------------------------------ one mEVP sub-iteration
def mevp_stress(par, hx, hy, u, v, P, S, dt=None, cgh=None, cga=None):
    """S <- (1 - 1/alpha) S + (1/alpha) Proj sigma(u, v); S = [3][8, ny, nx] (s11, s12, s22), returns new arrays.
    Adaptive form (par["aevp_c"] > 0; DESIGN.md section 3.5, after Kimmritz, Danilov & Losch 2016): every element relaxes with the
    alpha its own largest viscosity of this sub-iteration asks for, alpha_e^2 = max(alpha_min^2, c zeta_e dt / (m_e |K|)), m_e the
    nodal mass at its centre node (alpha_min where that node is ice-free); returns (new arrays, alpha_e [ny, nx])"""
    _, ny, nx = S[0].shape
    g, w = gauss_unit(3)
    Minv = np.linalg.inv(mass_matrix(8))
    out = [s.copy() for s in S]
    ia = 1.0 / par["alpha"]
    adaptive = par.get("aevp_c", 0.0) > 0.0
    alpha_e = np.zeros((ny, nx))
    for iy in range(ny):
        for ix in range(nx):
            ul, vl = local_nodes(u, ix, iy), local_nodes(v, ix, iy)
            rhs = np.zeros((3, 8))
            zetas = []
            for qy in range(3):
                for qx in range(3):
                    x, y = g[qx], g[qy]
                    e11 = cg_value(ul, x, y, dx=1) / hx
                    e22 = cg_value(vl, x, y, dy=1) / hy
                    e12 = 0.5 * (cg_value(ul, x, y, dy=1) / hy + cg_value(vl, x, y, dx=1) / hx)
                    p = P[3 * qy + qx, iy, ix]
                    delta = np.sqrt(par["delta_min"] ** 2 + (e11 + e22) ** 2 + ((e11 - e22) ** 2 + 4.0 * e12 ** 2) / E_RATIO ** 2)
                    zeta = p / (2.0 * delta)
                    zetas.append(zeta)
                    eta = zeta / E_RATIO ** 2
                    tr = e11 + e22
                    sig = (2.0 * eta * e11 + (zeta - eta) * tr - 0.5 * p, 2.0 * eta * e12, 2.0 * eta * e22 + (zeta - eta) * tr - 0.5 * p)
                    for i in range(8):
                        for c in range(3):
                            rhs[c, i] += w[qx] * w[qy] * psi(i, x, y) * sig[c]
            if adaptive:
                hc, ac = cgh[2 * iy + 1, 2 * ix + 1], cga[2 * iy + 1, 2 * ix + 1]  # the element's centre node
                alpha = par["aevp_alpha_min"]
                if not ice_free(par, hc, ac):
                    mass = par["rho_ice"] * max(hc, par["h_min"])
                    alpha = max(alpha, np.sqrt(par["aevp_c"] * max(zetas) * dt / (mass * hx * hy)))
                alpha_e[iy, ix] = alpha
                ia = 1.0 / alpha
            for c in range(3):
                out[c][:, iy, ix] = (S[c][:, iy, ix] + ia * (Minv @ rhs[c] - S[c][:, iy, ix])) if adaptive else ((1.0 - ia) * S[c][:, iy, ix] + ia * (Minv @ rhs[c]))
    return (out, alpha_e) if adaptive else out


def mevp_velocity(par, hx, hy, dt, S, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, alpha_e=None):
    """the momentum update of DESIGN.md section 3.2 at every interior node, v = 0 on the boundary; adaptive form: beta of a node from the
    alphas of the elements it belongs to, scaled by the mass ratios"""
    _, ny, nx = S[0].shape
    g, w = gauss_unit(5)
    nn, nm = 2 * nx + 1, 2 * ny + 1
    divx, divy, lump = np.zeros((nm, nn)), np.zeros((nm, nn)), np.zeros((nm, nn))
    beta_n = np.full((nm, nn), par["beta"])
    if par.get("aevp_c", 0.0) > 0.0:
        # every element offers its nine nodes alpha_e times ITS mass (at its centre node); a node divides the largest offer by its own mass:
        # alpha_e beta_n then meets the stability bound of every element-node pair
        hn = np.maximum(cgh, par["h_min"])
        offer = np.zeros((nm, nn))
        for iy in range(ny):
            for ix in range(nx):
                blk = offer[2 * iy:2 * iy + 3, 2 * ix:2 * ix + 3]
                np.maximum(blk, alpha_e[iy, ix] * hn[2 * iy + 1, 2 * ix + 1], out=blk)
        beta_n = np.maximum(offer / hn, par["aevp_alpha_min"])
        for gy in range(nm):
            for gx in range(nn):
                if ice_free(par, cgh[gy, gx], cga[gy, gx]):
                    beta_n[gy, gx] = par["aevp_alpha_min"]
    for iy in range(ny):
        for ix in range(nx):
            for ay in range(3):
                for ax in range(3):
                    ix_, iy_, l = 0.0, 0.0, 0.0
                    for qy in range(5):
                        for qx in range(5):
                            x, y, wq = g[qx], g[qy], w[qx] * w[qy]
                            s11, s12, s22 = (dg_value(S[c][:, iy, ix], x, y) for c in range(3))
                            gxp = DLAG[ax](x) * LAG[ay](y) / hx  # d phi / dx in physical coordinates
                            gyp = LAG[ax](x) * DLAG[ay](y) / hy
                            ix_ -= wq * (s11 * gxp + s12 * gyp)
                            iy_ -= wq * (s12 * gxp + s22 * gyp)
                            l += wq * LAG[ax](x) * LAG[ay](y)
                    divx[2 * iy + ay, 2 * ix + ax] += hx * hy * ix_
                    divy[2 * iy + ay, 2 * ix + ax] += hx * hy * iy_
                    lump[2 * iy + ay, 2 * ix + ax] += hx * hy * l
    un, vn = np.zeros((nm, nn)), np.zeros((nm, nn))
    for gy in range(1, nm - 1):
        for gx in range(1, nn - 1):
            m = par["rho_ice"] * max(cgh[gy, gx], par["h_min"])
            a = min(max(cga[gy, gx], 0.0), 1.0)
            fx, fy = divx[gy, gx] / lump[gy, gx], divy[gy, gx] / lump[gy, gx]
            if ice_free(par, cgh[gy, gx], cga[gy, gx]):  # free drift: full exposure, no stress from the neighbours
                a, fx, fy = 1.0, 0.0, 0.0
            c = a * par["c_ocean"] * par["rho_ocean"] * np.hypot(uo[gy, gx] - u[gy, gx], vo[gy, gx] - v[gy, gx])
            beta = beta_n[gy, gx]
            den = (m / dt) * (1.0 + beta) + c
            un[gy, gx] = ((m / dt) * (beta * u[gy, gx] + u0[gy, gx]) + a * tax[gy, gx] + c * uo[gy, gx]
                          + m * par["fc"] * (v[gy, gx] - vo[gy, gx]) + fx) / den
            vn[gy, gx] = ((m / dt) * (beta * v[gy, gx] + v0[gy, gx]) + a * tay[gy, gx] + c * vo[gy, gx]
                          - m * par["fc"] * (u[gy, gx] - uo[gy, gx]) + fy) / den
    return un, vn


def ice_free(par, h_node, a_node):
    """no ice to speak of at this node -- concentration below min_conc, or a true thickness (mean thickness
    over concentration) below min_thick; the shape of the column model's cut-off"""
    if par.get("min_conc", 0.0) <= 0.0 and par.get("min_thick", 0.0) <= 0.0:
        return False
    if a_node < par["min_conc"] or h_node <= par["h_min"]:  # no cover, or a mass that would be the floor's
        return True
    return h_node / a_node < par["min_thick"]  # a_node >= min_conc > 0 here (min_conc = 0: a_node >= 0, 0 / 0 and x / 0 compare False / as inf)


# ------------------------------------------------------------------------------------------------ closure of the transport
def limit(F, lo, hi, cap):
    """DESIGN.md section 3.3 on a DG2 field [6, ny, nx], returns a new array: cell means above hi are set to hi (cap), then the
    higher coefficients are scaled by the largest theta <= 1 that keeps the values at the 9 volume and 12 edge Gauss points and at the
    4 corners in [lo, hi]"""
    _, ny, nx = F.shape
    g, _ = gauss_unit(3)
    pts = [(x, y) for y in g for x in g] + [(0.5, s) for s in g] + [(-0.5, s) for s in g] + [(s, 0.5) for s in g] + [(s, -0.5) for s in g]
    pts += [(x, y) for y in (-0.5, 0.5) for x in (-0.5, 0.5)]  # the corners: every CG2 node of the element is then among the points
    out = F.copy()
    for iy in range(ny):
        for ix in range(nx):
            c = out[:, iy, ix]
            if cap and c[0] > hi:
                c[0] = hi
            mean = c[0]
            theta = 1.0
            for (x, y) in pts:
                dev = dg_value(c, x, y) - mean  # what theta scales
                if mean + dev < lo:
                    theta = min(theta, (mean - lo) / -dev if mean > lo else 0.0)
                if mean + dev > hi:
                    theta = min(theta, (hi - mean) / dev if mean < hi else 0.0)
            c[1:] *= theta
    return out


# ------------------------------------------------------------------------------------------------ DG2 transport
def advection_velocity(u, v, nx, ny):
    """L2 projection of the CG2 velocity on DG2 and its normal component at the 3 Gauss points of every edge"""
    Minv = np.linalg.inv(mass_matrix(6))
    g6, w6 = gauss_unit(6)
    g3, _ = gauss_unit(3)
    vx, vy = np.zeros((6, ny, nx)), np.zeros((6, ny, nx))
    for iy in range(ny):
        for ix in range(nx):
            ul, vl = local_nodes(u, ix, iy), local_nodes(v, ix, iy)
            bx, by = np.zeros(6), np.zeros(6)
            for qy in range(6):
                for qx in range(6):
                    x, y, wq = g6[qx], g6[qy], w6[qx] * w6[qy]
                    for i in range(6):
                        bx[i] += wq * psi(i, x, y) * cg_value(ul, x, y)
                        by[i] += wq * psi(i, x, y) * cg_value(vl, x, y)
            vx[:, iy, ix], vy[:, iy, ix] = Minv @ bx, Minv @ by
    unx, uny = np.zeros((3, ny, nx + 1)), np.zeros((3, ny + 1, nx))
    for k in range(3):
        for iy in range(ny):
            for ex in range(nx + 1):
                unx[k, iy, ex] = sum(LAG[a](g3[k]) * u[2 * iy + a, 2 * ex] for a in range(3))
        for ey in range(ny + 1):
            for ix in range(nx):
                uny[k, ey, ix] = sum(LAG[a](g3[k]) * v[2 * ey, 2 * ix + a] for a in range(3))
    return vx, vy, unx, uny


def transport_stage(hx, hy, dt, a, b, phi0, phis, adv):
    """out = a phi0 + b (phis + dt L(phis)), L the upwind DG2 operator with zero inflow at the edge of the array"""
    vx, vy, unx, uny = adv
    _, ny, nx = phis.shape
    Minv = np.linalg.inv(mass_matrix(6))
    g5, w5 = gauss_unit(5)
    g3, w3 = gauss_unit(3)
    out = np.zeros_like(phis)

    def val(ix, iy, x, y):  # the field in element (ix, iy); outside the array: nothing flows in
        if ix < 0 or ix >= nx or iy < 0 or iy >= ny:
            return 0.0
        return dg_value(phis[:, iy, ix], x, y)

    for iy in range(ny):
        for ix in range(nx):
            rhs = np.zeros(6)
            for qy in range(5):
                for qx in range(5):
                    x, y, wq = g5[qx], g5[qy], w5[qx] * w5[qy]
                    f = val(ix, iy, x, y)
                    ux, uy = dg_value(vx[:, iy, ix], x, y), dg_value(vy[:, iy, ix], x, y)
                    for i in range(6):
                        rhs[i] += wq * f * (ux * psi_dx(i, x, y) / hx + uy * psi_dy(i, x, y) / hy)
            for k in range(3):
                s, wk = g3[k], w3[k]
                # outward normal velocity, upwind value, on the right / left / top / bottom edge
                for (vn, inner, outer, px, py, h) in (
                        (unx[k, iy, ix + 1], val(ix, iy, 0.5, s), val(ix + 1, iy, -0.5, s), 0.5, s, hx),
                        (-unx[k, iy, ix], val(ix, iy, -0.5, s), val(ix - 1, iy, 0.5, s), -0.5, s, hx),
                        (uny[k, iy + 1, ix], val(ix, iy, s, 0.5), val(ix, iy + 1, s, -0.5), s, 0.5, hy),
                        (-uny[k, iy, ix], val(ix, iy, s, -0.5), val(ix, iy - 1, s, 0.5), s, -0.5, hy)):
                    up = inner if vn >= 0.0 else outer
                    for i in range(6):
                        rhs[i] -= wk * vn * up * psi(i, px, py) / h
            out[:, iy, ix] = a * phi0[:, iy, ix] + b * (phis[:, iy, ix] + dt * (Minv @ rhs))
    return out


# ------------------------------------------------------------------------------------------------ the fixture's case
PARAMS = dict(rho_ice=900.0, rho_atm=1.3, rho_ocean=1026.0, c_atm=1.2e-3, c_ocean=5.5e-3, pstar=27.5e3, compaction=20.0,
              delta_min=2e-9, fc=1.46e-4, alpha=300.0, beta=300.0, h_min=1e-4, min_conc=1e-12, min_thick=0.01)
ADAPTIVE = dict(aevp_c=(2.4 * np.pi) ** 2, aevp_alpha_min=8.0)
CASE = dict(nx=6, ny=5, hx=700.0, hy=900.0, dt=120.0, seed=20261004, rk_a=0.75, rk_b=0.25)


def case_inputs():
    """seeded random fields on the 6 x 5 grid: every term of the scheme is exercised (A above 1 and H below 0 at some
    Gauss points for the clamps and the limiter, cell means of A above 1 for the cap, a node with a thickness below h_min and
    ice-free nodes around two thin elements, hx != hy, non-zero Coriolis, ocean and wind)"""
    c = CASE
    nx, ny = c["nx"], c["ny"]
    rng = np.random.default_rng(c["seed"])
    nodal = (2 * ny + 1, 2 * nx + 1)

    def vel(scale):
        f = scale * rng.standard_normal(nodal)
        f[0, :] = f[-1, :] = 0.0
        f[:, 0] = f[:, -1] = 0.0
        return f

    inp = {"u": vel(0.1), "v": vel(0.1), "u0": vel(0.1), "v0": vel(0.1)}
    inp["uo"], inp["vo"] = 0.05 * rng.standard_normal(nodal), 0.05 * rng.standard_normal(nodal)
    inp["ua"], inp["va"] = 8.0 * rng.standard_normal(nodal), 8.0 * rng.standard_normal(nodal)
    H = np.zeros((6, ny, nx))
    A = np.zeros((6, ny, nx))
    H[0] = 0.3 + 0.1 * rng.standard_normal((ny, nx))
    H[1:] = 0.02 * rng.standard_normal((5, ny, nx))
    H[:, 2, 3] = 0.0
    H[0, 2, 3] = 5e-5  # thinner than h_min at the centre node of this element
    H[0, 1, 1], H[1, 1, 1] = 0.002, 0.05  # ... and negative at the left Gauss points of this one (max(h, 0) in the ice strength)
    A[0] = 0.9 + 0.08 * rng.standard_normal((ny, nx))
    A[1:] = 0.05 * rng.standard_normal((5, ny, nx))
    inp["H"], inp["A"] = H, A
    inp["S"] = [1e3 * rng.standard_normal((8, ny, nx)) for _ in range(3)]
    inp["phi"] = np.concatenate([0.5 + 0.2 * rng.standard_normal((1, ny, nx)), 0.05 * rng.standard_normal((5, ny, nx))])
    inp["phi0"] = inp["phi"] + 0.01 * rng.standard_normal((6, ny, nx))
    return inp


def case_outputs(inp=None):
    """everything the fixture holds: the per-step preparation, ONE mEVP sub-iteration and ONE DG2 transport stage"""
    c, par = CASE, PARAMS
    inp = case_inputs() if inp is None else inp
    out = {}
    out["pg"] = ice_strength(par, inp["H"], inp["A"])
    out["cgh"], out["cga"] = nodal_mean(inp["H"]), nodal_mean(inp["A"])
    out["tax"], out["tay"] = wind_stress(par, inp["ua"], inp["va"])
    S = mevp_stress(par, c["hx"], c["hy"], inp["u"], inp["v"], out["pg"], inp["S"])
    out["s11"], out["s12"], out["s22"] = S
    out["u_new"], out["v_new"] = mevp_velocity(par, c["hx"], c["hy"], c["dt"], S, inp["u"], inp["v"], inp["u0"], inp["v0"], out["tax"],
                                               out["tay"], inp["uo"], inp["vo"], out["cgh"], out["cga"])
    adv = advection_velocity(inp["u"], inp["v"], c["nx"], c["ny"])
    out["vx_dg"], out["vy_dg"], out["un_x"], out["un_y"] = adv
    out["phi_stage"] = transport_stage(c["hx"], c["hy"], c["dt"], c["rk_a"], c["rk_b"], inp["phi0"], inp["phi"], adv)
    # the closure of the transport on the case's thickness (bounded below) and concentration (both bounds, capped mean)
    out["H_limited"] = limit(inp["H"], 0.0, np.inf, False)
    out["A_limited"] = limit(inp["A"], 0.0, 1.0, True)
    # the same sub-iteration with local, solution-adaptive alpha and beta (v3 of the fixture).  The constant is chosen so that the case
    # has elements at the lower bound, elements far above it and the ice-free centre node of element (3, 2)
    pa = dict(par, **ADAPTIVE)
    Sa, alpha_e = mevp_stress(pa, c["hx"], c["hy"], inp["u"], inp["v"], out["pg"], inp["S"], dt=c["dt"], cgh=out["cgh"], cga=out["cga"])
    out["ad_s11"], out["ad_s12"], out["ad_s22"] = Sa
    out["ad_alpha"] = alpha_e
    out["ad_u_new"], out["ad_v_new"] = mevp_velocity(pa, c["hx"], c["hy"], c["dt"], Sa, inp["u"], inp["v"], inp["u0"], inp["v0"], out["tax"],
                                                     out["tay"], inp["uo"], inp["vo"], out["cgh"], out["cga"], alpha_e=alpha_e)
    return out
