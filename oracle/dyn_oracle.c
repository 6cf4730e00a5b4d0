/*
 * dyn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the DG transport step and the mEVP stress/velocity sub-cycle on a
 * uniform rectangular mesh.
 *
 * PARITY UNPINNED.  /root/reference holds NO dynamics code: CMakeLists.txt:43-46 comments the
 * "dynamics" component out, there is no dynamics/ directory, and no test or fixture for DG or mEVP
 * exists anywhere in the tree.  There is therefore no reference
 * file:line this file could follow and no golden vector it could be pinned to.  It restates the
 * published formulation (DG upwind transport with SSP-RK time stepping; Mehlmann & Richter 2017 mEVP;
 * Richter et al., GMD 2023 discretisation: CG2 velocity, 8-coefficient DG stress, 3x3 Gauss points),
 * as laid out in DESIGN.md section 3, and is validated by analytic properties only
 * (tests/test_oracle_dynamics.py).
 *
 * Independence from the product: the basis tables are rebuilt here at run time by numerical
 * quadrature from the basis-function definitions; the product's generated header
 * (nextsimdg_amd/csrc/dg_tables.h) is not included.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dyn_oracle.h"

#ifdef ORACLE_OMP
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#else
#define OMP_FOR
#endif

/* ------------------------------------------------------------------ basis definitions */
static double psi(int i, double x, double y)
{
    switch (i) {
    case 0: return 1.;
    case 1: return x;
    case 2: return y;
    case 3: return x * x - 1. / 12.;
    case 4: return y * y - 1. / 12.;
    case 5: return x * y;
    case 6: return y * (x * x - 1. / 12.);
    default: return x * (y * y - 1. / 12.);
    }
}
static double psi_x(int i, double x, double y)
{
    switch (i) {
    case 1: return 1.;
    case 3: return 2. * x;
    case 5: return y;
    case 6: return 2. * x * y;
    case 7: return y * y - 1. / 12.;
    default: return 0.;
    }
}
static double psi_y(int i, double x, double y)
{
    switch (i) {
    case 2: return 1.;
    case 4: return 2. * y;
    case 5: return x;
    case 6: return x * x - 1. / 12.;
    case 7: return 2. * x * y;
    default: return 0.;
    }
}
/* quadratic Lagrange basis on s in [-1/2,1/2], nodes at -1/2, 0, 1/2 */
static double lag(int k, double s)
{
    switch (k) {
    case 0: return 2. * s * s - s;
    case 1: return 1. - 4. * s * s;
    default: return 2. * s * s + s;
    }
}
static double dlag(int k, double s)
{
    switch (k) {
    case 0: return 4. * s - 1.;
    case 1: return -8. * s;
    default: return 4. * s + 1.;
    }
}
static double cg2(int a, double x, double y) { return lag(a % 3, x) * lag(a / 3, y); }
static double cg2_x(int a, double x, double y) { return dlag(a % 3, x) * lag(a / 3, y); }
static double cg2_y(int a, double x, double y) { return lag(a % 3, x) * dlag(a / 3, y); }

static void gauss(int n, double* pt, double* wt)
{
    if (n == 1) {
        pt[0] = 0.;
        wt[0] = 1.;
    } else if (n == 2) {
        const double a = 0.5 / sqrt(3.);
        pt[0] = -a;
        pt[1] = a;
        wt[0] = wt[1] = 0.5;
    } else if (n == 3) {
        const double a = 0.5 * sqrt(0.6);
        pt[0] = -a;
        pt[1] = 0.;
        pt[2] = a;
        wt[0] = wt[2] = 5. / 18.;
        wt[1] = 8. / 18.;
    } else { /* n == 4, used only to integrate the constant operator tables exactly (degree 7) */
        const double a = 0.5 * sqrt(3. / 7. - 2. / 7. * sqrt(6. / 5.));
        const double b = 0.5 * sqrt(3. / 7. + 2. / 7. * sqrt(6. / 5.));
        pt[0] = -b;
        pt[1] = -a;
        pt[2] = a;
        pt[3] = b;
        wt[0] = wt[3] = 0.5 * (18. - sqrt(30.)) / 36.;
        wt[1] = wt[2] = 0.5 * (18. + sqrt(30.)) / 36.;
    }
}

typedef struct {
    int ready;
    double mass[8], imass[8];
    double dx[8][9], dy[8][9]; /* (1/m_i) int psi_i d phi_a */
    double pv[8][9]; /* (1/m_i) int psi_i phi_a */
    double psin[9][8]; /* psi_i at local CG2 node a */
    double lump[9];
} optab_t;
static optab_t OT;

static void init_tables(void)
{
    if (OT.ready)
        return;
    double pt[4], wt[4];
    gauss(4, pt, wt);
    for (int i = 0; i < 8; ++i) {
        double m = 0;
        for (int qy = 0; qy < 4; ++qy)
            for (int qx = 0; qx < 4; ++qx)
                m += wt[qx] * wt[qy] * psi(i, pt[qx], pt[qy]) * psi(i, pt[qx], pt[qy]);
        OT.mass[i] = m;
        OT.imass[i] = 1. / m;
    }
    for (int a = 0; a < 9; ++a) {
        double l = 0;
        for (int qy = 0; qy < 4; ++qy)
            for (int qx = 0; qx < 4; ++qx)
                l += wt[qx] * wt[qy] * cg2(a, pt[qx], pt[qy]);
        OT.lump[a] = l;
        for (int i = 0; i < 8; ++i) {
            double sx = 0, sy = 0, sv = 0;
            for (int qy = 0; qy < 4; ++qy)
                for (int qx = 0; qx < 4; ++qx) {
                    const double w = wt[qx] * wt[qy], p = psi(i, pt[qx], pt[qy]);
                    sx += w * p * cg2_x(a, pt[qx], pt[qy]);
                    sy += w * p * cg2_y(a, pt[qx], pt[qy]);
                    sv += w * p * cg2(a, pt[qx], pt[qy]);
                }
            OT.dx[i][a] = sx * OT.imass[i];
            OT.dy[i][a] = sy * OT.imass[i];
            OT.pv[i][a] = sv * OT.imass[i];
            OT.psin[a][i] = psi(i, -0.5 + 0.5 * (a % 3), -0.5 + 0.5 * (a / 3));
        }
    }
    OT.ready = 1;
}

void oracle_dyn_init(void) { init_tables(); }

void oracle_mevp_default_params(oracle_mevp_params* p)
{
    /* SURVEY.md App. B.2 "typical constants" (box-test values of the mEVP literature) */
    p->rho_ice = 900.;
    p->rho_atm = 1.3;
    p->rho_ocean = 1026.;
    p->c_atm = 1.2e-3;
    p->c_ocean = 5.5e-3;
    p->pstar = 27.5e3;
    p->compaction = 20.;
    p->delta_min = 2e-9;
    p->fc = 1.46e-4;
    p->alpha = 1500.;
    p->beta = 1500.;
    p->h_min = 1e-4;
    /* the column model's own cut-off values (nextsim_thermo.min_conc / min_thick, physics/src/modules/NextsimPhysics.cpp:81-82) */
    p->min_conc = 1e-12;
    p->min_thick = 0.01;
    p->aevp_c = 0.; /* uniform alpha, beta */
    p->aevp_alpha_min = 50.;
}

/* ice-free-node rule (dyn_oracle.h) */
static int node_ice_free(const oracle_mevp_params* p, double cgh, double cga)
{
    const int rule = p->min_conc > 0. || p->min_thick > 0.;
    return rule && (cga < p->min_conc || cgh < p->min_thick * cga || cgh <= p->h_min);
}

int oracle_dg_ncoef(int order) { return order == 0 ? 1 : (order == 1 ? 3 : 6); }

#define NN(nx) (2 * (nx) + 1)

/* ------------------------------------------------------------------ advection velocity */
void oracle_prepare_advection(int nx, int ny, int order, const double* u, const double* v,
    double* vx_dg, double* vy_dg, double* un_x, double* un_y)
{
    init_tables();
    const int nc = oracle_dg_ncoef(order), ng = order + 1;
    const long N = (long)nx * ny;
    const int nn = NN(nx);
    double gp[4], gw[4];
    gauss(ng, gp, gw);
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const long e = (long)iy * nx + ix;
            for (int i = 0; i < nc; ++i) {
                double sx = 0, sy = 0;
                for (int a = 0; a < 9; ++a) {
                    const long n = (long)(2 * iy + a / 3) * nn + 2 * ix + a % 3;
                    sx += OT.pv[i][a] * u[n];
                    sy += OT.pv[i][a] * v[n];
                }
                vx_dg[i * N + e] = sx;
                vy_dg[i * N + e] = sy;
            }
        }
    /* x-normal velocity on the (nx+1)*ny vertical edges */
    for (int iy = 0; iy < ny; ++iy)
        for (int ex = 0; ex <= nx; ++ex)
            for (int g = 0; g < ng; ++g) {
                double s = 0;
                for (int k = 0; k < 3; ++k)
                    s += lag(k, gp[g]) * u[(long)(2 * iy + k) * nn + 2 * ex];
                un_x[(long)g * (nx + 1) * ny + (long)iy * (nx + 1) + ex] = s;
            }
    /* y-normal velocity on the nx*(ny+1) horizontal edges */
    for (int ey = 0; ey <= ny; ++ey)
        for (int ix = 0; ix < nx; ++ix)
            for (int g = 0; g < ng; ++g) {
                double s = 0;
                for (int k = 0; k < 3; ++k)
                    s += lag(k, gp[g]) * v[(long)(2 * ey) * nn + 2 * ix + k];
                un_y[(long)g * nx * (ny + 1) + (long)ey * nx + ix] = s;
            }
}

/* ------------------------------------------------------------------ DG transport */
static double dg_eval(const double* f, long N, long e, int nc, double x, double y)
{
    double s = 0;
    for (int c = 0; c < nc; ++c)
        s += f[c * N + e] * psi(c, x, y);
    return s;
}

/* out = a*phi0 + b*(phis + dt*L(phis)) on element rows [j0,j1); the edges of the local array are
 * treated as the physical boundary with zero inflow. */
void oracle_transport_stage(int nx, int ny, int j0, int j1, double hx, double hy, int order, double dt,
    double a, double b, const double* phi0, const double* phis, double* out, const double* vx_dg,
    const double* vy_dg, const double* un_x, const double* un_y)
{
    const int nc = oracle_dg_ncoef(order), ng = order + 1, nq = order + 1;
    const long N = (long)nx * ny;
    const long NEX = (long)(nx + 1) * ny, NEY = (long)nx * (ny + 1);
    double gp[4], gw[4];
    gauss(ng, gp, gw);
    init_tables();
    OMP_FOR
    for (int iy = j0; iy < j1; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const long e = (long)iy * nx + ix;
            double rhs[6] = { 0, 0, 0, 0, 0, 0 };
            if (order > 0) { /* cell term */
                for (int qy = 0; qy < nq; ++qy)
                    for (int qx = 0; qx < nq; ++qx) {
                        const double x = gp[qx], y = gp[qy], w = gw[qx] * gw[qy];
                        const double f = dg_eval(phis, N, e, nc, x, y);
                        const double vx = dg_eval(vx_dg, N, e, nc, x, y);
                        const double vy = dg_eval(vy_dg, N, e, nc, x, y);
                        for (int i = 0; i < nc; ++i)
                            rhs[i] += w * f * (vx * psi_x(i, x, y) / hx + vy * psi_y(i, x, y) / hy);
                    }
            }
            for (int g = 0; g < ng; ++g) {
                const double s = gp[g], w = gw[g];
                /* right edge */
                {
                    const double un = un_x[g * NEX + (long)iy * (nx + 1) + ix + 1];
                    const double fin = dg_eval(phis, N, e, nc, 0.5, s);
                    const double fout = (ix + 1 < nx) ? dg_eval(phis, N, e + 1, nc, -0.5, s) : 0.;
                    const double flux = fmax(un, 0.) * fin + fmin(un, 0.) * fout;
                    for (int i = 0; i < nc; ++i)
                        rhs[i] -= w * flux * psi(i, 0.5, s) / hx;
                }
                /* left edge */
                {
                    const double un = un_x[g * NEX + (long)iy * (nx + 1) + ix];
                    const double fin = dg_eval(phis, N, e, nc, -0.5, s);
                    const double fout = (ix > 0) ? dg_eval(phis, N, e - 1, nc, 0.5, s) : 0.;
                    const double flux = fmax(un, 0.) * fout + fmin(un, 0.) * fin;
                    for (int i = 0; i < nc; ++i)
                        rhs[i] += w * flux * psi(i, -0.5, s) / hx;
                }
                /* top edge */
                {
                    const double un = un_y[g * NEY + (long)(iy + 1) * nx + ix];
                    const double fin = dg_eval(phis, N, e, nc, s, 0.5);
                    const double fout = (iy + 1 < ny) ? dg_eval(phis, N, e + nx, nc, s, -0.5) : 0.;
                    const double flux = fmax(un, 0.) * fin + fmin(un, 0.) * fout;
                    for (int i = 0; i < nc; ++i)
                        rhs[i] -= w * flux * psi(i, s, 0.5) / hy;
                }
                /* bottom edge */
                {
                    const double un = un_y[g * NEY + (long)iy * nx + ix];
                    const double fin = dg_eval(phis, N, e, nc, s, -0.5);
                    const double fout = (iy > 0) ? dg_eval(phis, N, e - nx, nc, s, 0.5) : 0.;
                    const double flux = fmax(un, 0.) * fout + fmin(un, 0.) * fin;
                    for (int i = 0; i < nc; ++i)
                        rhs[i] += w * flux * psi(i, s, -0.5) / hy;
                }
            }
            for (int i = 0; i < nc; ++i)
                out[i * N + e] = a * phi0[i * N + e] + b * (phis[i * N + e] + dt * OT.imass[i] * rhs[i]);
        }
}

/* SSP Runge-Kutta of order (order+1): Euler / Heun / Shu-Osher RK3 */
void oracle_transport_step(int nx, int ny, double hx, double hy, int order, double dt, double* phi,
    const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y, double* scratch)
{
    const int nc = oracle_dg_ncoef(order);
    const long M = (long)nc * nx * ny;
    double* t1 = scratch;
    double* t2 = scratch + M;
    if (order == 0) {
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 0., 1., phi, phi, t1, vx_dg, vy_dg, un_x, un_y);
        memcpy(phi, t1, M * sizeof(double));
    } else if (order == 1) {
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 0., 1., phi, phi, t1, vx_dg, vy_dg, un_x, un_y);
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 0.5, 0.5, phi, t1, t2, vx_dg, vy_dg, un_x, un_y);
        memcpy(phi, t2, M * sizeof(double));
    } else {
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 0., 1., phi, phi, t1, vx_dg, vy_dg, un_x, un_y);
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 0.75, 0.25, phi, t1, t2, vx_dg, vy_dg, un_x, un_y);
        oracle_transport_stage(nx, ny, 0, ny, hx, hy, order, dt, 1. / 3., 2. / 3., phi, t2, t1, vx_dg, vy_dg, un_x, un_y);
        memcpy(phi, t1, M * sizeof(double));
    }
}

/* ------------------------------------------------------------------ closure of the transport: cap + scaling limiter */
void oracle_transport_limit(int nx, int ny, int j0, int j1, int order, double* phi, double lo, double hi, int cap)
{
    const int nc = oracle_dg_ncoef(order), ng = order + 1;
    const long N = (long)nx * ny;
    double gp[4], gw[4];
    gauss(ng, gp, gw);
    for (int iy = j0; iy < j1; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const long e = (long)iy * nx + ix;
            if (cap && phi[e] > hi)
                phi[e] = hi;
            if (order == 0)
                continue;
            const double mean = phi[e];
            double mn = INFINITY, mx = -INFINITY;
            for (int qy = 0; qy < ng; ++qy)
                for (int qx = 0; qx < ng; ++qx) {
                    const double v = dg_eval(phi, N, e, nc, gp[qx], gp[qy]);
                    mn = fmin(mn, v), mx = fmax(mx, v);
                }
            for (int g = 0; g < ng; ++g) {
                const double v[4] = { dg_eval(phi, N, e, nc, 0.5, gp[g]), dg_eval(phi, N, e, nc, -0.5, gp[g]),
                    dg_eval(phi, N, e, nc, gp[g], 0.5), dg_eval(phi, N, e, nc, gp[g], -0.5) };
                for (int k = 0; k < 4; ++k)
                    mn = fmin(mn, v[k]), mx = fmax(mx, v[k]);
            }
            for (int k = 0; k < 4; ++k) { /* the four corners: with the edge mid-points and the centre, the CG2 nodes of the element */
                const double v = dg_eval(phi, N, e, nc, k % 2 ? 0.5 : -0.5, k / 2 ? 0.5 : -0.5);
                mn = fmin(mn, v), mx = fmax(mx, v);
            }
            double theta = 1.;
            if (mn < lo)
                theta = fmin(theta, mean > lo ? (mean - lo) / (mean - mn) : 0.);
            if (mx > hi)
                theta = fmin(theta, mean < hi ? (hi - mean) / (mx - mean) : 0.);
            if (theta < 1.)
                for (int c = 1; c < nc; ++c)
                    phi[c * N + e] *= theta;
        }
}

/* ------------------------------------------------------------------ DG -> CG2 nodal average */
void oracle_dg_to_cg(int nx, int ny, int ncoef, const double* f_dg, double* f_cg)
{
    init_tables();
    const long N = (long)nx * ny;
    const int nn = NN(nx), nm = NN(ny);
    for (int gy = 0; gy < nm; ++gy)
        for (int gx = 0; gx < nn; ++gx) {
            double s = 0;
            int cnt = 0;
            /* adjacent elements: ix in {(gx-1)/2, gx/2} for even gx, gx/2 for odd gx */
            const int ix_hi = gx / 2, ix_lo = (gx % 2 == 0) ? gx / 2 - 1 : gx / 2;
            const int iy_hi = gy / 2, iy_lo = (gy % 2 == 0) ? gy / 2 - 1 : gy / 2;
            for (int iy = iy_lo; iy <= iy_hi; ++iy)
                for (int ix = ix_lo; ix <= ix_hi; ++ix) {
                    if (ix < 0 || ix >= nx || iy < 0 || iy >= ny)
                        continue;
                    const int a = (gy - 2 * iy) * 3 + (gx - 2 * ix);
                    const long e = (long)iy * nx + ix;
                    double val = 0;
                    for (int c = 0; c < ncoef; ++c)
                        val += f_dg[c * N + e] * OT.psin[a][c];
                    s += val;
                    ++cnt;
                }
            f_cg[(long)gy * nn + gx] = s / cnt;
        }
}

/* ------------------------------------------------------------------ ice strength at the 3x3 Gauss points */
void oracle_ice_strength(int nx, int ny, int j0, int j1, const oracle_mevp_params* p, const double* H,
    const double* A, double* pg)
{
    const long N = (long)nx * ny;
    double gp[4], gw[4];
    gauss(3, gp, gw);
    for (int iy = j0; iy < j1; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const long e = (long)iy * nx + ix;
            for (int q = 0; q < 9; ++q) {
                const double x = gp[q % 3], y = gp[q / 3];
                const double h = fmax(dg_eval(H, N, e, 6, x, y), 0.);
                const double a = fmin(fmax(dg_eval(A, N, e, 6, x, y), 0.), 1.);
                pg[q * N + e] = p->pstar * h * exp(-p->compaction * (1. - a));
            }
        }
}

/* ------------------------------------------------------------------ mEVP stress update */
void oracle_mevp_stress(int nx, int ny, int k0, int k1, double hx, double hy, const oracle_mevp_params* p,
    const double* u, const double* v, const double* pg, double* s11, double* s12, double* s22, double dt, const double* cgh,
    const double* cga, double* alpha_e)
{
    init_tables();
    const long N = (long)nx * ny;
    const int nn = NN(nx);
    double gp[4], gw[4];
    gauss(3, gp, gw);
    const double ialpha = 1. / p->alpha;
    const double dmin2 = p->delta_min * p->delta_min;
    OMP_FOR
    for (int iy = k0; iy < k1; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const long e = (long)iy * nx + ix;
            double ul[9], vl[9];
            for (int a = 0; a < 9; ++a) {
                const long n = (long)(2 * iy + a / 3) * nn + 2 * ix + a % 3;
                ul[a] = u[n];
                vl[a] = v[n];
            }
            double E11[8], E12[8], E22[8];
            for (int i = 0; i < 8; ++i) {
                double uxx = 0, uyy = 0, vxx = 0, vyy = 0;
                for (int a = 0; a < 9; ++a) {
                    uxx += OT.dx[i][a] * ul[a];
                    uyy += OT.dy[i][a] * ul[a];
                    vxx += OT.dx[i][a] * vl[a];
                    vyy += OT.dy[i][a] * vl[a];
                }
                E11[i] = uxx / hx;
                E22[i] = vyy / hy;
                E12[i] = 0.5 * (uyy / hy + vxx / hx);
            }
            double r11[8] = { 0 }, r12[8] = { 0 }, r22[8] = { 0 };
            double zeta_e = 0.; /* adaptive form: the largest viscosity P / (2 Delta) of the element */
            for (int q = 0; q < 9; ++q) {
                const double x = gp[q % 3], y = gp[q / 3], w = gw[q % 3] * gw[q / 3];
                double e11 = 0, e12 = 0, e22 = 0;
                for (int i = 0; i < 8; ++i) {
                    const double ps = psi(i, x, y);
                    e11 += E11[i] * ps;
                    e12 += E12[i] * ps;
                    e22 += E22[i] * ps;
                }
                const double P = pg[q * N + e];
                const double delta = sqrt(dmin2 + 1.25 * (e11 * e11 + e22 * e22) + 1.5 * e11 * e22 + e12 * e12);
                const double pd = P / delta;
                zeta_e = fmax(zeta_e, 0.5 * pd);
                const double t11 = pd * (0.625 * e11 + 0.375 * e22) - 0.5 * P;
                const double t22 = pd * (0.625 * e22 + 0.375 * e11) - 0.5 * P;
                const double t12 = pd * 0.25 * e12;
                for (int i = 0; i < 8; ++i) {
                    const double ps = w * psi(i, x, y);
                    r11[i] += ps * t11;
                    r12[i] += ps * t12;
                    r22[i] += ps * t22;
                }
            }
            double ia = ialpha;
            if (p->aevp_c > 0.) { /* local, solution-adaptive alpha (dyn_oracle.h) */
                const long nc = (long)(2 * iy + 1) * nn + 2 * ix + 1; /* the element's centre node */
                double a2 = p->aevp_alpha_min * p->aevp_alpha_min;
                if (!node_ice_free(p, cgh[nc], cga[nc]))
                    a2 = fmax(a2, p->aevp_c * zeta_e * dt / (p->rho_ice * fmax(cgh[nc], p->h_min) * hx * hy));
                alpha_e[e] = sqrt(a2);
                ia = 1. / alpha_e[e];
            }
            for (int i = 0; i < 8; ++i) {
                s11[i * N + e] = (1. - ia) * s11[i * N + e] + ia * OT.imass[i] * r11[i];
                s12[i * N + e] = (1. - ia) * s12[i * N + e] + ia * OT.imass[i] * r12[i];
                s22[i * N + e] = (1. - ia) * s22[i * N + e] + ia * OT.imass[i] * r22[i];
            }
        }
}

/* ------------------------------------------------------------------ mEVP velocity update */
void oracle_mevp_velocity(int nx, int ny, int j0, int j1, double hx, double hy, double dt,
    const oracle_mevp_params* p, const double* s11, const double* s12, const double* s22,
    const double* u_old, const double* v_old, double* u_new, double* v_new, const double* u0,
    const double* v0, const double* tax, const double* tay, const double* uo, const double* vo,
    const double* cgh, const double* cga, const double* alpha_e)
{
    init_tables();
    const long N = (long)nx * ny;
    const int nn = NN(nx), nm = NN(ny);
    const double f_ocean = p->c_ocean * p->rho_ocean;
    OMP_FOR
    for (int gy = 2 * j0; gy < 2 * j1; ++gy)
        for (int gx = 0; gx < nn - 1; ++gx) {
            const long n = (long)gy * nn + gx;
            if (gx == 0 || gy == 0 || gy == nm - 1) { /* Dirichlet boundary (gx == nn-1 is never owned) */
                u_new[n] = 0.;
                v_new[n] = 0.;
                continue;
            }
            double divx = 0, divy = 0, lumped = 0;
            double beta = p->beta; /* adaptive form: the largest alpha_e h'_c(e) / h'_n of the adjacent elements, at least alpha_min */
            if (p->aevp_c > 0.)
                beta = 0.;
            const int ix_hi = gx / 2, ix_lo = (gx % 2 == 0) ? gx / 2 - 1 : gx / 2;
            const int iy_hi = gy / 2, iy_lo = (gy % 2 == 0) ? gy / 2 - 1 : gy / 2;
            for (int iy = iy_lo; iy <= iy_hi; ++iy)
                for (int ix = ix_lo; ix <= ix_hi; ++ix) {
                    if (ix < 0 || ix >= nx || iy < 0 || iy >= ny)
                        continue;
                    const int a = (gy - 2 * iy) * 3 + (gx - 2 * ix);
                    const long e = (long)iy * nx + ix;
                    double gx11 = 0, gy12 = 0, gx12 = 0, gy22 = 0;
                    for (int i = 0; i < 8; ++i) {
                        const double mdx = OT.mass[i] * OT.dx[i][a], mdy = OT.mass[i] * OT.dy[i][a];
                        gx11 += mdx * s11[i * N + e];
                        gy12 += mdy * s12[i * N + e];
                        gx12 += mdx * s12[i * N + e];
                        gy22 += mdy * s22[i * N + e];
                    }
                    /* -(sigma, grad phi_a)_K */
                    divx -= hy * gx11 + hx * gy12;
                    divy -= hy * gx12 + hx * gy22;
                    lumped += hx * hy * OT.lump[a];
                    if (p->aevp_c > 0.) { /* alpha_e times the mass ratio (element's centre node : this node): alpha_e beta_n >= the pair's bound */
                        const long nce = (long)(2 * iy + 1) * nn + 2 * ix + 1;
                        beta = fmax(beta, alpha_e[e] * fmax(cgh[nce], p->h_min) / fmax(cgh[n], p->h_min));
                    }
                }
            const double uu = u_old[n], vv = v_old[n];
            const double du = uo[n] - uu, dv = vo[n] - vv;
            const double absocn = sqrt(du * du + dv * dv);
            const double h = fmax(cgh[n], p->h_min);
            /* ice-free-node rule (dyn_oracle.h): free drift at full exposure, the neighbours' stress divergence weighted by 2^-100 */
            const int ice_free = node_ice_free(p, cgh[n], cga[n]);
            if (p->aevp_c > 0.)
                beta = ice_free ? p->aevp_alpha_min : fmax(beta, p->aevp_alpha_min); /* (an ice-free node does not feel its neighbours' stress) */
            const double a_ = ice_free ? 1. : fmin(fmax(cga[n], 0.), 1.);
            const double wdiv = ice_free ? 0x1p-100 : 1.;
            const double mdt = p->rho_ice * h / dt;
            const double cdrag = a_ * f_ocean * absocn;
            const double denom = 1. / (mdt * (1. + beta) + cdrag);
            const double cor = p->rho_ice * h * p->fc;
            u_new[n] = denom * (mdt * (beta * uu + u0[n]) + a_ * tax[n] + cdrag * uo[n] + cor * (vv - vo[n]) + wdiv * (divx / lumped));
            v_new[n] = denom * (mdt * (beta * vv + v0[n]) + a_ * tay[n] + cdrag * vo[n] - cor * (uu - uo[n]) + wdiv * (divy / lumped));
        }
    if (j1 == ny) /* the right column and the top row are boundary nodes: keep them at zero */
        for (int gx = 0; gx < nn; ++gx) {
            u_new[(long)(nm - 1) * nn + gx] = 0.;
            v_new[(long)(nm - 1) * nn + gx] = 0.;
        }
    for (int gy = 2 * j0; gy < 2 * j1; ++gy) {
        u_new[(long)gy * nn + nn - 1] = 0.;
        v_new[(long)gy * nn + nn - 1] = 0.;
    }
}

void oracle_mevp_subcycle(int nx, int ny, double hx, double hy, double dt, int nsub,
    const oracle_mevp_params* p, double* s11, double* s12, double* s22, double* u, double* v,
    const double* u0, const double* v0, const double* tax, const double* tay, const double* uo,
    const double* vo, const double* cgh, const double* cga, const double* pg, double* scratch)
{
    const long nnodes = (long)NN(nx) * NN(ny);
    double* un = scratch;
    double* vn = scratch + nnodes;
    double* alpha_e = p->aevp_c > 0. ? (double*)malloc(sizeof(double) * (size_t)nx * ny) : NULL;
    for (int it = 0; it < nsub; ++it) {
        oracle_mevp_stress(nx, ny, 0, ny, hx, hy, p, u, v, pg, s11, s12, s22, dt, cgh, cga, alpha_e);
        oracle_mevp_velocity(nx, ny, 0, ny, hx, hy, dt, p, s11, s12, s22, u, v, un, vn, u0, v0, tax, tay,
            uo, vo, cgh, cga, alpha_e);
        memcpy(u, un, nnodes * sizeof(double));
        memcpy(v, vn, nnodes * sizeof(double));
    }
    free(alpha_e);
}

void oracle_wind_stress(long nnodes, const oracle_mevp_params* p, const double* ua, const double* va,
    double* tax, double* tay)
{
    const double f_atm = p->c_atm * p->rho_atm;
    for (long n = 0; n < nnodes; ++n) {
        const double absatm = sqrt(ua[n] * ua[n] + va[n] * va[n]);
        tax[n] = f_atm * absatm * ua[n];
        tay[n] = f_atm * absatm * va[n];
    }
}
