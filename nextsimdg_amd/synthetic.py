"""Synthetic inputs: the seeded column-physics fields of SURVEY.md section 8(d) and the 512 km box
test (cyclone wind over a circular ocean current) used for BASELINE configs 3-5.  numpy only."""
import numpy as np

from . import basis

COLUMN_SEED = 0x5EA1CE


def column_fields(n, seed=COLUMN_SEED):
    """state (hice,cice,hsnow,tice0), forcing (10 arrays) and newice for n elements."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: rng.uniform(lo, hi, n)
    hice = u(0, 3)
    hice[rng.random(n) < 0.10] = 0.0
    cice = u(0, 1)
    r = rng.random(n)
    cice[r < 0.10] = 0.0
    cice[(r >= 0.10) & (r < 0.15)] = 1.0
    hsnow = u(0, 0.5) * cice
    tice0 = u(-30, 0)
    tair = u(-35, 5)
    state = dict(hice=hice, cice=cice, hsnow=hsnow, tice0=tice0)
    forcing = dict(sst=u(-1.9, 2), sss=u(28, 36), tair=tair, tdew=tair - u(0, 5), slp=u(9.6e4, 1.04e5),
                   qsw=u(0, 300), qlw=u(150, 350), mld=u(5, 50), snowfall=u(0, 1e-4), wind=u(0, 25))
    return state, forcing, np.zeros(n)


def column_fields_smooth(nx, ny, L=512e3):
    """Thermodynamic state (hsnow, tice0) and forcing for the COUPLED runs (BASELINE config 5): the same variables
    and ranges as column_fields(), but smooth analytic functions of position instead of independent random numbers
    per element -- element-wise noise in the forcing makes the thermodynamics write grid-scale noise into H and A,
    which the dynamics then (rightly) cannot digest: the coupled model blows up within ~10 steps.  Winter
    conditions: cold air, little short-wave.  Returns dicts of [ny, nx] arrays."""
    x = (np.arange(nx) + 0.5)[None, :] / nx
    y = (np.arange(ny) + 0.5)[:, None] / ny
    s1, c1 = np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), np.cos(2 * np.pi * x) * np.sin(np.pi * y)
    one = np.ones((ny, nx))
    tair = -15.0 + 8.0 * s1
    state = dict(hsnow=0.05 * one + 0.03 * c1, tice0=-10.0 + 4.0 * s1)
    sss = 32.0 + 1.5 * s1
    # the mixed layer sits AT the freezing point (linear law, -0.055 S): the reference never updates sst
    # (SURVEY App. A.7 quirk 4), so any excess over T_f is an inexhaustible heat source that melts ~2 cm per step
    forcing = dict(sst=-0.055 * sss, sss=sss, tair=tair, tdew=tair - 2.0 - 1.0 * c1, slp=1.0e5 + 2.0e3 * c1,
                   qsw=40.0 + 30.0 * s1, qlw=230.0 + 40.0 * c1, mld=25.0 + 10.0 * s1, snowfall=2.0e-5 * (1.0 + c1),
                   wind=6.0 + 4.0 * s1)
    return state, {k: np.ascontiguousarray(v * one) for k, v in forcing.items()}


class BoxTest:
    """Square box of side L (default 512 km), closed boundaries (v = 0).  Fields follow the usual
    VP/mEVP benchmark set-up: H0 = 0.3 + 0.005 (sin(6e-5 x) + sin(3e-5 y)), A0 = 1, circular ocean
    current of 0.01 m/s, a cyclone with 15 m/s-scale winds at time t."""

    def __init__(self, nx, ny, L=512e3):
        self.nx, self.ny, self.L = nx, ny, L
        self.hx, self.hy = L / nx, L / ny

    def stable_alpha(self, dt, pstar=27.5e3, delta_min=2e-9, rho_ice=900.0, h_ice=0.3, safety=2.4, floor=1500.0):
        """alpha = beta for which the mEVP pseudo-time iteration is linearly stable on this mesh:
        alpha*beta >= pi^2 * zeta_max * dt / (m * h^2), zeta_max = P*H/(2 Delta_min), m = rho_i H.
        (Measured on the MI355X, round 1: with alpha = beta = 1500 round-off differences between two
        kernel variants grow x150 per sub-iteration at h = 250 m; the bound -- 1.2e4 there -- is sharp:
        12000 is stable, 6000 is not.  Round 2: that is the bound of the INITIAL state; a one-day run needs margin --
        with 1.2 x the bound the dynamics at 2048^2 / 4096^2 diverge after 21-23 model hours (deformation zones sharpen,
        the local ratio of ice strength to nodal mass grows), with 1.8 x they start to (free-drift speeds at hour 24),
        with 2.4 x and 3.6 x the day completes (profiles/r02_alpha_margin.txt); the default is 2.4 x.
        The flop and byte counts do not depend on alpha.)"""
        h = min(self.hx, self.hy)
        zeta_max = pstar * h_ice / (2.0 * delta_min)
        bound = np.sqrt(np.pi ** 2 * zeta_max * dt / (rho_ice * h_ice * h * h))
        return float(max(floor, safety * bound))

    def stable_delta_min(self, dt, alpha=1500.0, pstar=27.5e3, rho_ice=900.0, safety=2.4, floor=2e-9):
        """The other way round (round 5, profiles/r05_closure.md): the SMALLEST regularisation Delta_min -- never below the
        literature's 2e-9 1/s -- for which the BASELINE's alpha = beta = 1500 satisfies the stability bound above with its margin:
        zeta_max / m = P* / (2 Delta_min rho_i) <= alpha^2 h^2 / (safety^2 pi^2 dt).  Why the hosts choose THIS way: with the bound's
        alpha on a fine mesh (14 438 at 500 m, 57 751 at 125 m) 120 sub-iterations move the stress and the velocity 1 % or less of
        the way to their viscous-plastic state per model step, and a compressible cover (A0 = 0.9) then leaves the physical range
        within a model day or two whatever closes the transport; with alpha = 1500 and the viscosity capped accordingly
        (Delta_min 1.9e-7 at 500 m, 7.4e-7 at 250 m, 3.0e-6 at 125 m: creep below ~1 - 25 % per day) the same runs complete.
        stable_alpha(dt, delta_min=stable_delta_min(dt, alpha)) == alpha."""
        h = min(self.hx, self.hy)
        return float(max(floor, safety ** 2 * np.pi ** 2 * pstar * dt / (2.0 * rho_ice * h * h * alpha * alpha)))

    def subcycle_parameters(self, dt, alpha=1500.0, delta_min=None):
        """alpha = beta and Delta_min of the mEVP sub-cycle as the hosts set them: the named alpha with the regularisation the mesh
        needs for it (default), or -- delta_min given -- that regularisation with the alpha its stability bound asks for"""
        if delta_min is None:
            return dict(alpha=float(alpha), beta=float(alpha), delta_min=self.stable_delta_min(dt, alpha))
        a = self.stable_alpha(dt, delta_min=delta_min)
        return dict(alpha=a, beta=a, delta_min=float(delta_min))

    def H0(self, x, y):
        return 0.3 + 0.005 * (np.sin(6e-5 * x) + np.sin(3e-5 * y))

    def A0(self, x, y):
        return 1.0 + 0 * x

    def dg_fields(self, ncoef=6):
        H = basis.project_dg(self.H0, self.nx, self.ny, self.L, self.L, ncoef, nq=3)
        A = basis.project_dg(self.A0, self.nx, self.ny, self.L, self.L, ncoef, nq=3)
        A[1:] = 0.0
        return H, A

    def ocean(self):
        X, Y = basis.node_coords(self.nx, self.ny, self.L, self.L)
        vmax = 0.01
        return vmax * (2 * Y - self.L) / self.L, vmax * (self.L - 2 * X) / self.L

    def wind(self, t=0.0):
        X, Y = basis.node_coords(self.nx, self.ny, self.L, self.L)
        cm = 0.5 * self.L + 0.1 * self.L * t / 86400.0 / 4.0
        alpha = np.deg2rad(72.0)
        dx, dy = (cm - X), (cm - Y)
        r = np.sqrt(dx * dx + dy * dy)
        s = 15.0 * np.exp(-r / (0.2 * self.L)) / (0.2 * self.L) * 1.0
        ua = s * (np.cos(alpha) * dx + np.sin(alpha) * dy) * (0.2 * self.L) / 1e5
        va = s * (-np.sin(alpha) * dx + np.cos(alpha) * dy) * (0.2 * self.L) / 1e5
        return ua, va


def rotating_patch(nx, ny, order, kind="bump"):
    """BASELINE config 2: unit square, rigid rotation (omega = 2 pi) about the centre multiplied by a
    smooth cut-off so that v = 0 on the boundary; initial patch centred at (0.5, 0.25)."""
    def phi0(x, y):
        r2 = (x - 0.5) ** 2 + (y - 0.25) ** 2
        if kind == "bump":
            return np.exp(-50.0 * r2 / 0.25)
        if kind == "narrow":  # tails < 1e-7 where the velocity cut-off starts: the exact solution is a rigid rotation
            return np.exp(-800.0 * r2)
        r = np.sqrt(r2) / 0.15
        return np.where(r < 1, 0.5 * (1 + np.cos(np.pi * np.minimum(r, 1))), 0.0)

    X, Y = basis.node_coords(nx, ny, 1.0, 1.0)
    w = 2 * np.pi
    r = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2)
    cut = np.where(r < 0.4, 1.0, np.where(r < 0.5, 0.5 * (1 + np.cos(np.pi * (r - 0.4) / 0.1)), 0.0))
    u = -w * (Y - 0.5) * cut
    v = w * (X - 0.5) * cut
    phi = basis.project_dg(phi0, nx, ny, 1.0, 1.0, basis.NCOEF[order], nq=4)
    return phi, np.ascontiguousarray(u), np.ascontiguousarray(v), phi0
