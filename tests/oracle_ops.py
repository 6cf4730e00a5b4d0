"""CPU stand-in for nextsimdg_amd.abi.Context used ONLY by the multi-rank CPU tests: the same method
names, implemented with the oracle on CPU torch tensors.  It lets world_size-2 `gloo` tests exercise
the row-block decomposition and halo exchange (nextsimdg_amd/rowblock.py) without a GPU.  It lives
under tests/ because it calls the oracle; the product never imports it."""
import numpy as np

import oracle_lib as O


def _np(t):
    return t.numpy()


class OracleOps:
    def __init__(self, mevp_variant=1, **mevp):
        self.p = O.mevp_params(**mevp)
        self.cp = O.column_params()
        self.mevp_variant = mevp_variant  # 2: the driver uses mevp_iterate2 (two sub-iterations per pass)
        self._alpha_e = None

    # adaptive alpha / beta (params.aevp_c > 0): the stress update needs the step's dt and nodal means and leaves every element's alpha
    # for the velocity update of the same sub-iteration
    def _ad_stress(self):
        if not self.p.aevp_c > 0:
            return {}
        dt, _, _, _, cgh, cga = self.nodal
        if self._alpha_e is None or self._alpha_e.shape != (self.ny, self.nx):
            self._alpha_e = np.zeros((self.ny, self.nx))
        return dict(dt=dt, cgh=cgh, cga=cga, alpha_e=self._alpha_e)

    def _ad_velocity(self):
        return dict(alpha_e=self._alpha_e) if self.p.aevp_c > 0 else {}

    def column_step(self, dt, state, forcing, newice, diag=None):
        # plane views ([ny, nx] slices of the DG arrays) are contiguous: flatten without copying
        st = {k: _np(v).reshape(-1) for k, v in state.items()}
        fo = {k: _np(v).reshape(-1) for k, v in forcing.items()}
        O.column_step(self.cp, dt, st, fo, _np(newice).reshape(-1))

    def private_zeros(self, nc, ny, nx, device):
        import torch

        return torch.zeros(nc, ny, nx, dtype=torch.float64, device=device)  # the oracle keeps coefficient planes

    @staticmethod
    def private_rows(f, j0, j1):
        return f[:, j0:j1]

    @staticmethod
    def private_to_planes(f, nx):
        return f

    @staticmethod
    def planes_to_private(a):
        return a

    def set_grid(self, nx, ny, hx, hy):
        self.nx, self.ny, self.hx, self.hy = nx, ny, hx, hy

    def dg_to_cg(self, f_dg, f_cg):
        _np(f_cg)[:] = O.dg_to_cg(self.nx, self.ny, _np(f_dg))

    def ice_strength(self, H, A, pg, j0=0, j1=None):
        j1 = self.ny if j1 is None else j1
        _np(pg)[:, j0:j1] = O.ice_strength(self.nx, self.ny, self.p, _np(H), _np(A), j0, j1)[:, j0:j1]

    def wind_stress(self, ua, va, tax, tay):
        a, b = O.wind_stress(self.p, _np(ua), _np(va))
        _np(tax)[:] = a
        _np(tay)[:] = b

    def mevp_prepare(self, dt, H, A, wind, ocean, u0v0, packed):
        cgh, cga = O.dg_to_cg(self.nx, self.ny, _np(H)), O.dg_to_cg(self.nx, self.ny, _np(A))
        tau = O.wind_stress(self.p, _np(wind[0]), _np(wind[1]))
        # u0, v0 alias the iterate in the driver: keep copies, as the packed coefficients do on the device
        self.nodal = (dt, [_np(x).copy() for x in u0v0], list(tau), [_np(x) for x in ocean], cgh, cga)

    def mevp_pack_nodal(self, dt, u0v0, tau, ocean, cgh, cga, packed):
        # the oracle has no packed layout: remember the per-step fields the coefficients are made from
        self.nodal = (dt, [_np(x) for x in u0v0], [_np(x) for x in tau], [_np(x) for x in ocean], _np(cgh), _np(cga))

    def mevp_iterate(self, k0, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        dt, u0v0, tau, ocean, cgh, cga = self.nodal
        for a, b in zip(s_in, s_out):  # the oracle's stress update is in place: seed the output rows
            _np(b)[:, k0:j1] = _np(a)[:, k0:j1]
        so = [_np(x) for x in s_out]
        O.mevp_stress(self.nx, self.ny, k0, j1, self.hx, self.hy, self.p, _np(uv_old[0]), _np(uv_old[1]), _np(pg), *so, **self._ad_stress())
        O.mevp_velocity(self.nx, self.ny, j0, j1, self.hx, self.hy, dt, self.p, so, [_np(x) for x in uv_old],
                        [_np(x) for x in uv_new], u0v0, tau, ocean, cgh, cga, **self._ad_velocity())

    def mevp_iterate2(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """two sub-iterations on the owned rows [j0, j1), reading two rows below / one above, exactly the
        dependency region of a two-stage pass of the pipelined kernel (csrc/mevp_fused4.hip, nst = 2)"""
        dt, u0v0, tau, ocean, cgh, cga = self.nodal
        ny = self.ny
        a0, a1 = max(j0 - 2, 0), min(j1, ny - 1)  # rows of sub-iteration p
        sp = [_np(x).copy() for x in s_in]
        uo = [_np(x) for x in uv_old]
        O.mevp_stress(self.nx, ny, a0, a1 + 1, self.hx, self.hy, self.p, uo[0], uo[1], _np(pg), *sp, **self._ad_stress())
        up = [x.copy() for x in uo]
        O.mevp_velocity(self.nx, ny, max(j0 - 1, 0), a1 + 1, self.hx, self.hy, dt, self.p, sp, uo, up, u0v0, tau, ocean, cgh, cga, **self._ad_velocity())
        O.mevp_stress(self.nx, ny, max(j0 - 1, 0), j1, self.hx, self.hy, self.p, up[0], up[1], _np(pg), *sp, **self._ad_stress())
        for a, b in zip(sp, s_out):
            _np(b)[:, j0:j1] = a[:, j0:j1]
        O.mevp_velocity(self.nx, ny, j0, j1, self.hx, self.hy, dt, self.p, sp, up, [_np(x) for x in uv_new], u0v0, tau, ocean,
                        cgh, cga, **self._ad_velocity())

    def mevp_iterate3(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """three sub-iterations on the owned rows [j0, j1), reading three rows below / two above, exactly the
        dependency region of a three-stage pass of the pipelined kernel (csrc/mevp_fused4.hip, nst = 3)"""
        dt, u0v0, tau, ocean, cgh, cga = self.nodal
        ny = self.ny
        top = lambda r: min(r, ny - 1) + 1  # exclusive end of a row range clipped to the array
        sp = [_np(x).copy() for x in s_in]
        uo = [_np(x) for x in uv_old]
        args = (self.nx, ny)
        O.mevp_stress(*args, max(j0 - 3, 0), top(j1 + 1), self.hx, self.hy, self.p, uo[0], uo[1], _np(pg), *sp, **self._ad_stress())
        up = [x.copy() for x in uo]
        O.mevp_velocity(*args, max(j0 - 2, 0), top(j1 + 1), self.hx, self.hy, dt, self.p, sp, uo, up, u0v0, tau, ocean, cgh, cga, **self._ad_velocity())
        O.mevp_stress(*args, max(j0 - 2, 0), top(j1), self.hx, self.hy, self.p, up[0], up[1], _np(pg), *sp, **self._ad_stress())
        up2 = [x.copy() for x in up]
        O.mevp_velocity(*args, max(j0 - 1, 0), top(j1), self.hx, self.hy, dt, self.p, sp, up, up2, u0v0, tau, ocean, cgh, cga, **self._ad_velocity())
        O.mevp_stress(*args, max(j0 - 1, 0), j1, self.hx, self.hy, self.p, up2[0], up2[1], _np(pg), *sp, **self._ad_stress())
        for a, b in zip(sp, s_out):
            _np(b)[:, j0:j1] = a[:, j0:j1]
        O.mevp_velocity(*args, j0, j1, self.hx, self.hy, dt, self.p, sp, up2, [_np(x) for x in uv_new], u0v0, tau, ocean, cgh, cga, **self._ad_velocity())

    def mevp_iterate4(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """four sub-iterations on the owned rows [j0, j1), reading four rows below / three above, exactly the
        dependency region of the four-iterations-per-pass kernel (csrc/mevp_fused4.hip): sub-iteration k of v = 4 updates
        the stress on rows j0-(v-k) .. j1+(v-2-k) and the velocity of the nodes owned by rows j0-(v-1-k) .. j1+(v-2-k)"""
        v = 4
        dt, u0v0, tau, ocean, cgh, cga = self.nodal
        ny = self.ny
        top = lambda r: min(r, ny - 1) + 1  # exclusive end of a row range clipped to the array
        sp = [_np(x).copy() for x in s_in]
        u = [_np(x) for x in uv_old]
        args = (self.nx, ny)
        for k in range(v):
            last = k == v - 1
            end = j1 if last else top(j1 + (v - 2 - k))
            O.mevp_stress(*args, max(j0 - (v - k), 0), end, self.hx, self.hy, self.p, u[0], u[1], _np(pg), *sp, **self._ad_stress())
            if last:
                for a, b in zip(sp, s_out):
                    _np(b)[:, j0:j1] = a[:, j0:j1]
            un = [_np(x) for x in uv_new] if last else [x.copy() for x in u]
            O.mevp_velocity(*args, max(j0 - (v - 1 - k), 0), end, self.hx, self.hy, dt, self.p, sp, u, un, u0v0, tau, ocean, cgh, cga, **self._ad_velocity())
            u = un

    def prepare_advection(self, order, u, v, vx, vy, unx, uny):
        res = O.prepare_advection(self.nx, self.ny, order, _np(u), _np(v))
        for dst, src in zip((vx, vy, unx, uny), res):
            _np(dst)[:] = src

    def set_transport_bounds(self, bounds):
        self.transport_bounds = tuple(bounds or ())

    def transport_limit(self, order, j0, j1, fields):
        assert len(fields) == len(self.transport_bounds)
        for f, (lo, hi, cap) in zip(fields, self.transport_bounds):
            O.transport_limit(self.nx, self.ny, order, _np(f), lo, hi, cap, j0, j1)

    def transport_stage(self, order, j0, j1, dt, a, b, phi0, phis, out, adv):
        advn = tuple(_np(x) for x in adv)
        for p0, ps, o in zip(phi0, phis, out):
            O.transport_stage(self.nx, self.ny, j0, j1, self.hx, self.hy, order, dt, a, b, _np(p0), _np(ps), _np(o), advn)
