"""ctypes bindings of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package nextsimdg_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_double_p = C.POINTER(C.c_double)


class ColumnParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "drag_ocean_q", "drag_ocean_t", "drag_ice_t", "ocean_albedo", "i0", "min_conc", "min_thick",
        "ks", "h0", "phi_m", "ccsm_ice_albedo", "ccsm_snow_albedo")] + [
        ("flooding", C.c_int), ("albedo_kind", C.c_int), ("freezing_kind", C.c_int), ("reserved", C.c_int)]


class MevpParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "rho_ice", "rho_atm", "rho_ocean", "c_atm", "c_ocean", "pstar", "compaction", "delta_min", "fc",
        "alpha", "beta", "h_min", "min_conc", "min_thick", "aevp_c", "aevp_alpha_min")]


ALBEDO = {"smu": 0, "smu2": 1, "ccsm": 2}
FREEZING = {"linear": 0, "unesco": 1}
DIAG = ["rho", "qa", "qw", "qi", "cspec", "tau", "hi", "hs", "cnew", "qia", "qio", "subl", "dqdt",
        "hifroms", "qow"]
STATE = ["hice", "cice", "hsnow", "tice0"]
FORCING = ["sst", "sss", "tair", "tdew", "slp", "qsw", "qlw", "mld", "snowfall", "wind"]


def build(omp=False):
    target = "liboracle_omp.so" if omp else "liboracle.so"
    path = os.path.join(ORACLE_DIR, target)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("column_oracle.c", "dyn_oracle.c", "column_oracle.h", "dyn_oracle.h")]
    if (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, target)])
    return path


_libs = {}


def dp(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def lib(omp=False):
    if omp in _libs:
        return _libs[omp]
    L = C.CDLL(build(omp))
    L.oracle_freezing_point.restype = C.c_double
    L.oracle_freezing_point.argtypes = [C.c_int, C.c_double]
    L.oracle_albedo.restype = C.c_double
    L.oracle_albedo.argtypes = [C.POINTER(ColumnParams), C.c_double, C.c_double]
    L.oracle_column_default_params.argtypes = [C.POINTER(ColumnParams)]
    L.oracle_column_step.argtypes = [C.POINTER(ColumnParams), C.c_long, C.c_double] + [c_double_p] * 16
    L.oracle_mevp_default_params.argtypes = [C.POINTER(MevpParams)]
    L.oracle_dg_ncoef.restype = C.c_int
    L.oracle_prepare_advection.argtypes = [C.c_int] * 3 + [c_double_p] * 6
    L.oracle_transport_stage.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] + [C.c_double] * 3 + [c_double_p] * 7
    L.oracle_transport_step.argtypes = [C.c_int] * 2 + [C.c_double] * 2 + [C.c_int, C.c_double] + [c_double_p] * 6
    L.oracle_transport_limit.argtypes = [C.c_int] * 5 + [c_double_p] + [C.c_double] * 2 + [C.c_int]
    L.oracle_dg_to_cg.argtypes = [C.c_int] * 3 + [c_double_p] * 2
    L.oracle_ice_strength.argtypes = [C.c_int] * 4 + [C.POINTER(MevpParams)] + [c_double_p] * 3
    L.oracle_mevp_stress.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.POINTER(MevpParams)] + [c_double_p] * 6 + [C.c_double] + [c_double_p] * 3
    L.oracle_mevp_velocity.argtypes = [C.c_int] * 4 + [C.c_double] * 3 + [C.POINTER(MevpParams)] + [c_double_p] * 16
    L.oracle_mevp_subcycle.argtypes = [C.c_int] * 2 + [C.c_double] * 3 + [C.c_int, C.POINTER(MevpParams)] + [c_double_p] * 15
    L.oracle_wind_stress.argtypes = [C.c_long, C.POINTER(MevpParams)] + [c_double_p] * 4
    L.oracle_dyn_init()
    _libs[omp] = L
    return L


def column_params(**kw):
    p = ColumnParams()
    lib().oracle_column_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "albedo":
            p.albedo_kind = ALBEDO[v]
        elif k == "freezing":
            p.freezing_kind = FREEZING[v]
        else:
            assert hasattr(p, k), k
            setattr(p, k, v)
    return p


def mevp_params(**kw):
    p = MevpParams()
    lib().oracle_mevp_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def column_step(params, dt, state, forcing, newice, want_diag=False, omp=False):
    """state: dict of 4 arrays (updated in place); forcing: dict of 10 arrays; newice in place."""
    n = state["hice"].size
    diag = np.zeros((len(DIAG), n)) if want_diag else None
    lib(omp).oracle_column_step(C.byref(params), n, float(dt), *[dp(state[k]) for k in STATE],
                                *[dp(forcing[k]) for k in FORCING], dp(newice), dp(diag))
    return {k: diag[i] for i, k in enumerate(DIAG)} if want_diag else None


def ncoef(order):
    return {0: 1, 1: 3, 2: 6}[order]


def prepare_advection(nx, ny, order, u, v):
    nc, ng = ncoef(order), order + 1
    vx = np.zeros((nc, ny, nx))
    vy = np.zeros((nc, ny, nx))
    unx = np.zeros((ng, ny, nx + 1))
    uny = np.zeros((ng, ny + 1, nx))
    lib().oracle_prepare_advection(nx, ny, order, dp(u), dp(v), dp(vx), dp(vy), dp(unx), dp(uny))
    return vx, vy, unx, uny


def transport_stage(nx, ny, j0, j1, hx, hy, order, dt, a, b, phi0, phis, out, adv, omp=False):
    vx, vy, unx, uny = adv
    lib(omp).oracle_transport_stage(nx, ny, j0, j1, hx, hy, order, dt, a, b, dp(phi0), dp(phis), dp(out),
                                    dp(vx), dp(vy), dp(unx), dp(uny))


def transport_step(nx, ny, hx, hy, order, dt, phi, adv, omp=False):
    vx, vy, unx, uny = adv
    scratch = np.zeros(2 * phi.size)
    lib(omp).oracle_transport_step(nx, ny, hx, hy, order, dt, dp(phi), dp(vx), dp(vy), dp(unx), dp(uny), dp(scratch))


def transport_limit(nx, ny, order, phi, lo, hi, cap, j0=0, j1=None):
    """closure of the transport on rows [j0, j1), in place: cell-mean cap (cap) + scaling limiter to [lo, hi]"""
    lib().oracle_transport_limit(nx, ny, j0, ny if j1 is None else j1, order, dp(phi), float(lo), float(hi), int(bool(cap)))


def dg_to_cg(nx, ny, f_dg):
    out = np.zeros((2 * ny + 1, 2 * nx + 1))
    lib().oracle_dg_to_cg(nx, ny, f_dg.shape[0], dp(f_dg), dp(out))
    return out


def ice_strength(nx, ny, params, H, A, j0=0, j1=None):
    pg = np.zeros((9, ny, nx))
    lib().oracle_ice_strength(nx, ny, j0, ny if j1 is None else j1, C.byref(params), dp(H), dp(A), dp(pg))
    return pg


def mevp_stress(nx, ny, k0, k1, hx, hy, params, u, v, pg, s11, s12, s22, omp=False, dt=0.0, cgh=None, cga=None, alpha_e=None):
    """adaptive form (params.aevp_c > 0): dt, cgh, cga are needed and alpha_e [ny, nx] receives every element's alpha"""
    if params.aevp_c > 0 and (cgh is None or cga is None or alpha_e is None or not dt > 0):
        raise ValueError("the adaptive form needs dt, cgh, cga and alpha_e")
    null = C.cast(None, c_double_p)
    lib(omp).oracle_mevp_stress(nx, ny, k0, k1, hx, hy, C.byref(params), dp(u), dp(v), dp(pg), dp(s11), dp(s12), dp(s22), float(dt),
                                null if cgh is None else dp(cgh), null if cga is None else dp(cga), null if alpha_e is None else dp(alpha_e))


def mevp_velocity(nx, ny, j0, j1, hx, hy, dt, params, s, uv_old, uv_new, u0v0, tau, ocean, cgh, cga, omp=False, alpha_e=None):
    if params.aevp_c > 0 and alpha_e is None:
        raise ValueError("the adaptive form needs alpha_e (from mevp_stress)")
    lib(omp).oracle_mevp_velocity(nx, ny, j0, j1, hx, hy, dt, C.byref(params), dp(s[0]), dp(s[1]), dp(s[2]),
                                  dp(uv_old[0]), dp(uv_old[1]), dp(uv_new[0]), dp(uv_new[1]), dp(u0v0[0]), dp(u0v0[1]),
                                  dp(tau[0]), dp(tau[1]), dp(ocean[0]), dp(ocean[1]), dp(cgh), dp(cga),
                                  C.cast(None, c_double_p) if alpha_e is None else dp(alpha_e))


def mevp_subcycle(nx, ny, hx, hy, dt, nsub, params, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, omp=False):
    scratch = np.zeros(2 * u.size)
    lib(omp).oracle_mevp_subcycle(nx, ny, hx, hy, dt, nsub, C.byref(params), dp(s[0]), dp(s[1]), dp(s[2]), dp(u), dp(v),
                                  dp(u0), dp(v0), dp(tax), dp(tay), dp(uo), dp(vo), dp(cgh), dp(cga), dp(pg), dp(scratch))


def wind_stress(params, ua, va):
    tax = np.zeros_like(ua)
    tay = np.zeros_like(va)
    lib().oracle_wind_stress(ua.size, C.byref(params), dp(ua), dp(va), dp(tax), dp(tay))
    return tax, tay


def ref_leaf():
    """oracle/_ref/libref_leaf.so: the reference's own header-only functions, or None if not built."""
    path = os.path.join(ORACLE_DIR, "_ref", "libref_leaf.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    for f in ("ref_freezing_linear", "ref_freezing_unesco"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.c_double]
    L.ref_constant.restype = C.c_double
    L.ref_constant.argtypes = [C.c_int]
    return L
