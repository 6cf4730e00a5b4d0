"""Synthetic inputs: the seeded column-physics fields of SURVEY.md section 8(d) and the 512 km box
test (cyclone wind over a circular ocean current) used for BASELINE configs 3-5.  numpy only (the sub-cycle's
stability rule is asked of the library: it is host code, no device needed)."""
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

    # The stability rule of the explicit sub-cycle lives in the library, ONCE (nsdg_mevp_stable_params, include/nsdg.h; round 5 had a
    # copy here, one in the C++ host and one in the header's comment): alpha beta >= (2.4 pi)^2 zeta dt / (m h^2).  What was measured:
    # with alpha = beta = 1500 round-off differences between two kernel variants grow x150 per sub-iteration at h = 250 m; the bound
    # -- 1.2e4 there -- is sharp: 12000 is stable, 6000 is not (round 1).  That is the bound of the INITIAL state; a one-day run needs
    # margin: with 1.2 x the bound the dynamics at 2048^2 / 4096^2 diverge after 21-23 model hours, with 2.4 x the day completes
    # (profiles/r02_alpha_margin.txt).  The flop and byte counts do not depend on any of it.
    def _stable(self, mode, dt, **kw):
        import ctypes as C

        from . import abi

        p = abi.MevpParams()
        abi.load_library().nsdg_mevp_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, float(v))
        return abi.stable_mevp_params(p, mode, min(self.hx, self.hy), dt)

    def stable_alpha(self, dt, delta_min=2e-9):
        """uniform alpha = beta the stability bound asks for with this Delta_min (at least 1500): the configuration of rounds 1-4"""
        from . import abi

        return float(self._stable(abi.SUBCYCLE_KEEP_DELTA_MIN, dt, delta_min=delta_min).alpha)

    def stable_delta_min(self, dt, alpha=1500.0):
        """the smallest Delta_min (never below the literature's 2e-9) for which the uniform alpha = beta is stable on this mesh: the
        configuration of round 5 (1.9e-7 at 500 m, 7.4e-7 at 250 m, 3.0e-6 at 125 m: below that strain rate the ice creeps)"""
        from . import abi

        return float(self._stable(abi.SUBCYCLE_KEEP_ALPHA, dt, alpha=alpha).delta_min)

    def subcycle_parameters(self, dt, mode="adaptive", alpha=1500.0, delta_min=None):
        """keyword arguments for Context.mevp_default_params, as the hosts set the sub-cycle:
        "adaptive" (default since round 6): local, solution-adaptive alpha and beta with the stability bound's own constant, Delta_min
        the literature's 2e-9 (or `delta_min`), alpha_min = 50; "adaptive_converged": the same with the lower bound of alpha that lets 120
        sub-iterations converge on the mesh (include/nsdg.h: needs a time step that fits the mesh); "keep_alpha" (round 5): uniform alpha = beta = `alpha`, Delta_min raised to what the mesh
        needs for it; "keep_delta_min" (rounds 1-4): uniform alpha = beta from the bound for `delta_min` (2e-9)"""
        from . import abi

        dm = 2e-9 if delta_min is None else float(delta_min)
        if mode == "adaptive":
            p = self._stable(abi.SUBCYCLE_ADAPTIVE, dt, delta_min=dm)
        elif mode == "adaptive_converged":
            p = self._stable(abi.SUBCYCLE_ADAPTIVE_CONVERGED, dt, delta_min=dm)
        elif mode == "keep_alpha":
            p = self._stable(abi.SUBCYCLE_KEEP_ALPHA, dt, alpha=alpha, delta_min=dm)
        elif mode == "keep_delta_min":
            p = self._stable(abi.SUBCYCLE_KEEP_DELTA_MIN, dt, delta_min=dm)
        else:
            raise ValueError("mode must be adaptive, adaptive_converged, keep_alpha or keep_delta_min")
        return dict(alpha=float(p.alpha), beta=float(p.beta), delta_min=float(p.delta_min), aevp_c=float(p.aevp_c), aevp_alpha_min=float(p.aevp_alpha_min))

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
