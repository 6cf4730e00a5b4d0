"""numpy helpers for the DG / CG2 discretisation on a uniform rectangular mesh: basis functions,
Gauss rules, L2 projection of analytic fields and evaluation of DG fields.  Host-side utilities for
building synthetic inputs and analysing results; no kernel work happens here.

Layout conventions (shared with the HIP kernels and DESIGN.md section 2):
  element (ix, iy), ix fastest:           e = iy*nx + ix
  DG field with nc coefficients:          f[c, iy, ix]           (coefficient-major planes)
  CG2 nodal field:                        g[gy, gx], (2ny+1) x (2nx+1), node (2ix+ax, 2iy+ay)
"""
import numpy as np

NCOEF = {0: 1, 1: 3, 2: 6}


def psi(i, x, y):
    return [1.0 + 0 * x, x, y, x * x - 1 / 12, y * y - 1 / 12, x * y, y * (x * x - 1 / 12), x * (y * y - 1 / 12)][i]


MASS = np.array([1, 1 / 12, 1 / 12, 1 / 180, 1 / 180, 1 / 144, 1 / 2160, 1 / 2160])


def gauss(n):
    """points on [-1/2, 1/2], weights summing to one"""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * x, 0.5 * w


def cell_centres(nx, ny, lx, ly):
    hx, hy = lx / nx, ly / ny
    return (np.arange(nx) + 0.5) * hx, (np.arange(ny) + 0.5) * hy


def project_dg(func, nx, ny, lx, ly, ncoef, nq=4):
    """L2 projection of func(x, y) (vectorised) onto the DG space: returns [ncoef, ny, nx]"""
    hx, hy = lx / nx, ly / ny
    xc, yc = cell_centres(nx, ny, lx, ly)
    gp, gw = gauss(nq)
    out = np.zeros((ncoef, ny, nx))
    for qy in range(nq):
        for qx in range(nq):
            X = xc[None, :] + gp[qx] * hx
            Y = yc[:, None] + gp[qy] * hy
            F = func(X + 0 * Y, Y + 0 * X)
            for c in range(ncoef):
                out[c] += gw[qx] * gw[qy] * psi(c, gp[qx], gp[qy]) * F / MASS[c]
    return out


def eval_dg(f, xi, eta):
    """evaluate a DG field [nc, ny, nx] at reference point (xi, eta) of every cell"""
    return sum(f[c] * psi(c, xi, eta) for c in range(f.shape[0]))


def l2_error(f, func, lx, ly, nq=4):
    nc, ny, nx = f.shape
    hx, hy = lx / nx, ly / ny
    xc, yc = cell_centres(nx, ny, lx, ly)
    gp, gw = gauss(nq)
    err = 0.0
    for qy in range(nq):
        for qx in range(nq):
            X = xc[None, :] + gp[qx] * hx + 0 * yc[:, None]
            Y = yc[:, None] + gp[qy] * hy + 0 * xc[None, :]
            d = eval_dg(f, gp[qx], gp[qy]) - func(X, Y)
            err += gw[qx] * gw[qy] * np.sum(d * d) * hx * hy
    return np.sqrt(err)


def node_coords(nx, ny, lx, ly):
    """coordinates of the CG2 nodes: X[gy, gx], Y[gy, gx]"""
    x = np.arange(2 * nx + 1) * (lx / (2 * nx))
    y = np.arange(2 * ny + 1) * (ly / (2 * ny))
    return np.meshgrid(x, y)


def mass_total(f, hx, hy):
    return float(np.sum(f[0]) * hx * hy)
