// transport.hip -- DG upwind transport on a uniform rectangular mesh: advection-velocity preparation
// and the Runge-Kutta stage kernel (DG0 / DG1 / DG2).
//
// No counterpart in the reference snapshot (CMakeLists.txt:43-46 comments the dynamics component
// out); the scheme is the one stated in DESIGN.md section 3.1 and restated on the CPU in
// oracle/dyn_oracle.c (parity unpinned).
//
// Stage kernel shape: one lane per element, gather formulation -- every element evaluates the
// upwind flux on its own four edges (a shared edge is evaluated twice with bit-identical
// arithmetic), so there are no write conflicts and the result does not depend on the launch
// geometry or on a row-block decomposition.  Coefficients are coefficient-major planes, so each of
// the nc + 4*nc neighbour loads and the nc stores is a unit-stride wave access.  Basis values at the
// quadrature points are compile-time constants (tools/gen_tables.py) folded into the instruction
// stream by full unrolling; zero entries cost nothing.  HBM-bound: 1008 B / element-step for
// DG2 x 2 fields x RK3.
#include <cmath>
#include <cstdint>

#include "dg_tables.h"
#include "nsdg_internal.h"

namespace {

using namespace nsdg_tab;

template <int ORDER>
struct DG {
    static constexpr int NC = ORDER == 0 ? 1 : (ORDER == 1 ? 3 : 6);
    static constexpr int NG = ORDER + 1; // edge Gauss points
    static constexpr int NQ = (ORDER + 1) * (ORDER + 1); // volume Gauss points
};

// table accessors resolved at compile time after unrolling
template <int ORDER> __device__ __forceinline__ constexpr double t_psi(int q, int i) { return ORDER == 1 ? PSI_G2[q & 3][i] : PSI_G3[q][i]; }
template <int ORDER> __device__ __forceinline__ constexpr double t_psix(int q, int i) { return ORDER == 1 ? PSIX_G2[q & 3][i] : PSIX_G3[q][i]; }
template <int ORDER> __device__ __forceinline__ constexpr double t_psiy(int q, int i) { return ORDER == 1 ? PSIY_G2[q & 3][i] : PSIY_G3[q][i]; }
template <int ORDER> __device__ __forceinline__ constexpr double t_w(int q) { return ORDER == 1 ? W_G2[q & 3] : W_G3[q]; }
template <int ORDER> __device__ __forceinline__ constexpr double t_gw(int g) { return ORDER == 0 ? GW1[0] : (ORDER == 1 ? GW2[g & 1] : GW3[g]); }
template <int ORDER> __device__ __forceinline__ constexpr double t_l(int g, int i) { return ORDER == 0 ? PSI_L1[0][i] : (ORDER == 1 ? PSI_L2[g & 1][i] : PSI_L3[g][i]); }
template <int ORDER> __device__ __forceinline__ constexpr double t_r(int g, int i) { return ORDER == 0 ? PSI_R1[0][i] : (ORDER == 1 ? PSI_R2[g & 1][i] : PSI_R3[g][i]); }
template <int ORDER> __device__ __forceinline__ constexpr double t_b(int g, int i) { return ORDER == 0 ? PSI_B1[0][i] : (ORDER == 1 ? PSI_B2[g & 1][i] : PSI_B3[g][i]); }
template <int ORDER> __device__ __forceinline__ constexpr double t_t(int g, int i) { return ORDER == 0 ? PSI_T1[0][i] : (ORDER == 1 ? PSI_T2[g & 1][i] : PSI_T3[g][i]); }
template <int ORDER> __device__ __forceinline__ constexpr double t_le(int g, int k) { return ORDER == 0 ? L_E1[0][k] : (ORDER == 1 ? L_E2[g & 1][k] : L_E3[g][k]); }

#define MAXF 4
struct FieldPtrs {
    const double* phi0[MAXF];
    const double* phis[MAXF];
    double* out[MAXF];
    // closure of a full step (nsdg_transport_bounds_set; unused by a bare stage): bounds of field f at the quadrature points and
    // whether its cell mean is capped at hi; limit == 0: none
    double lo[MAXF], hi[MAXF];
    int cap[MAXF];
    int limit;
};

// acc += tab * val where tab is a compile-time table entry: after full unrolling the load of the
// constexpr table folds to a literal and the branch disappears, so zero entries cost nothing.
#define FMA_TAB(acc, tab, val)   \
    do {                         \
        const double t_ = (tab); \
        if (t_ != 0.0)           \
            acc += t_ * (val);   \
    } while (0)

// out = a phi0 + b (c + dt L_i), ONE expression for every kernel that performs a Runge-Kutta stage (the stage kernels and the
// fused-stages kernel round it identically); a == 0: the first stage, which does not read phi0
__device__ __forceinline__ double rk_update(double a, double b, double dt, double imass, double phi0, double c, double rhs)
{
    return a != 0. ? a * phi0 + b * (c + dt * imass * rhs) : b * (c + dt * imass * rhs);
}

// Closure of a transport step on ONE element (oracle_transport_limit): cap of the cell mean (ridging, for a
// field with cap != 0: a mean above hi becomes hi), then the Zhang-Shu scaling limiter -- the higher coefficients are scaled by
// the largest theta <= 1 that keeps the values at the scheme's own quadrature points -- the NQ volume Gauss points and the NG Gauss
// points of each edge -- and at the four corners inside [lo, hi]; the cell mean is never changed by the limiter.  ONE function for the marching kernel's
// epilogue and for the stand-alone kernel behind nsdg_transport_limit: the two round identically (fused step == staged step).
// The division runs only on lanes whose element leaves the range (rare: a handful of elements of a model step).
template <int ORDER>
__device__ __forceinline__ void limit_cell(double (&c)[DG<ORDER>::NC], double lo, double hi, bool cap)
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG, NQ = DG<ORDER>::NQ;
    if (cap)
        c[0] = fmin(c[0], hi);
    if constexpr (ORDER > 0) {
        // deviations from the mean at the points (the basis functions 1.. have zero mean: value = mean + deviation)
        double dmin = 0., dmax = 0.; // the mean itself is a convex combination of the volume-point values: the extrema straddle 0
        if constexpr (ORDER == 2) {
            // sum-factorised: d(x, y) = X(x) + Y(y) + c5 x y with X = c1 x + c3 (x^2 - 1/12), Y = c2 y + c4 (y^2 - 1/12) at the abscissae
            // -g, 0, g (Gauss) and -1/2, 1/2 (edges) -- seven distinct fp64 literals instead of the ~100 of the dense point tables,
            // which the marching kernel has no scalar registers left for (literals live in scalar register pairs)
            constexpr double G = 0.3872983346207417, P2E = 1. / 15., P2M = -1. / 12., P2H = 1. / 6.; // sqrt(3/5)/2; x^2 - 1/12 at g, 0, 1/2
            double X[5], Y[5];
            X[0] = c[3] * P2E - G * c[1], X[1] = c[3] * P2M, X[2] = c[3] * P2E + G * c[1], X[3] = c[3] * P2H - 0.5 * c[1], X[4] = c[3] * P2H + 0.5 * c[1];
            Y[0] = c[4] * P2E - G * c[2], Y[1] = c[4] * P2M, Y[2] = c[4] * P2E + G * c[2], Y[3] = c[4] * P2H - 0.5 * c[2], Y[4] = c[4] * P2H + 0.5 * c[2];
            const double GG = (G * G) * c[5], GH = (0.5 * G) * c[5];
#pragma unroll
            for (int qy = 0; qy < 3; ++qy)
#pragma unroll
                for (int qx = 0; qx < 3; ++qx) {
                    const int sg = (qx - 1) * (qy - 1);
                    const double d = sg == 0 ? X[qx] + Y[qy] : (sg > 0 ? (X[qx] + Y[qy]) + GG : (X[qx] + Y[qy]) - GG);
                    dmin = fmin(dmin, d), dmax = fmax(dmax, d);
                }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double e = q == 1 ? 0. : (q == 2 ? GH : -GH); // c5 x y on an edge x = 1/2 (y = 1/2) at the Gauss point q of the edge
                const double dr = (X[4] + Y[q]) + e, dl = (X[3] + Y[q]) - e, dt = (X[q] + Y[4]) + e, db = (X[q] + Y[3]) - e;
                dmin = fmin(dmin, fmin(fmin(dr, dl), fmin(dt, db)));
                dmax = fmax(dmax, fmax(fmax(dr, dl), fmax(dt, db)));
            }
            {
                // the four corners: with the edge mid-points and the centre (Gauss points above) these are the CG2 NODES of the element,
                // so that the nodal means the momentum equation takes its mass and concentration from stay inside the bounds as well
                const double Q = 0.25 * c[5];
                const double d00 = (X[3] + Y[3]) + Q, d10 = (X[4] + Y[3]) - Q, d01 = (X[3] + Y[4]) - Q, d11 = (X[4] + Y[4]) + Q;
                dmin = fmin(dmin, fmin(fmin(d00, d10), fmin(d01, d11)));
                dmax = fmax(dmax, fmax(fmax(d00, d10), fmax(d01, d11)));
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double d = 0.;
#pragma unroll
                for (int k = 1; k < NC; ++k)
                    FMA_TAB(d, t_psi<ORDER>(q, k), c[k]);
                dmin = fmin(dmin, d), dmax = fmax(dmax, d);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double dr = 0., dl = 0., dt = 0., db = 0.;
#pragma unroll
                for (int k = 1; k < NC; ++k) {
                    FMA_TAB(dr, t_r<ORDER>(g, k), c[k]);
                    FMA_TAB(dl, t_l<ORDER>(g, k), c[k]);
                    FMA_TAB(dt, t_t<ORDER>(g, k), c[k]);
                    FMA_TAB(db, t_b<ORDER>(g, k), c[k]);
                }
                dmin = fmin(dmin, fmin(fmin(dr, dl), fmin(dt, db)));
                dmax = fmax(dmax, fmax(fmax(dr, dl), fmax(dt, db)));
            }
            { // the four corners (DG1: c1 x + c2 y at x, y = -1/2, 1/2)
                const double a = 0.5 * c[1], b = 0.5 * c[2];
                const double d00 = -a - b, d10 = a - b, d01 = b - a, d11 = a + b;
                dmin = fmin(dmin, fmin(fmin(d00, d10), fmin(d01, d11)));
                dmax = fmax(dmax, fmax(fmax(d00, d10), fmax(d01, d11)));
            }
        }
        const double mean = c[0];
        if (mean + dmin < lo || mean + dmax > hi) {
            double theta = 1.;
            if (mean + dmin < lo)
                theta = fmin(theta, mean > lo ? (mean - lo) / -dmin : 0.);
            if (mean + dmax > hi)
                theta = fmin(theta, mean < hi ? (hi - mean) / dmax : 0.);
#pragma unroll
            for (int k = 1; k < NC; ++k)
                c[k] *= theta;
        }
    }
}

// edge-normal velocities of one element at the NG edge Gauss points
template <int NG>
struct EdgeVel {
    double l[NG], r[NG], b[NG], t[NG];
};

// What an element needs of a neighbour is the neighbour's TRACE at the NG Gauss points of the shared edge, not its NC
// coefficients: the traces are formed as soon as a neighbour's coefficients are loaded and the coefficients die
// (round 2: 12 instead of 24 live doubles for the four DG2 neighbours; the kernel was latency-bound at 2 waves per
// SIMD with 191-197 registers).  Same sums in the same order as before: bit-identical results.
template <int NG>
struct NbTrace {
    double l[NG], r[NG], b[NG], t[NG]; // the left / right / bottom / top neighbour's values on my edges
};
template <int ORDER>
__device__ __forceinline__ void trace_of_left(const double (&cl)[DG<ORDER>::NC], double (&tr)[DG<ORDER>::NG])
{
#pragma unroll
    for (int g = 0; g < DG<ORDER>::NG; ++g) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < DG<ORDER>::NC; ++k)
            FMA_TAB(s, t_r<ORDER>(g, k), cl[k]); // the left neighbour's right trace
        tr[g] = s;
    }
}
template <int ORDER>
__device__ __forceinline__ void trace_of_right(const double (&cr)[DG<ORDER>::NC], double (&tr)[DG<ORDER>::NG])
{
#pragma unroll
    for (int g = 0; g < DG<ORDER>::NG; ++g) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < DG<ORDER>::NC; ++k)
            FMA_TAB(s, t_l<ORDER>(g, k), cr[k]);
        tr[g] = s;
    }
}
template <int ORDER>
__device__ __forceinline__ void trace_of_bottom(const double (&cb)[DG<ORDER>::NC], double (&tr)[DG<ORDER>::NG])
{
#pragma unroll
    for (int g = 0; g < DG<ORDER>::NG; ++g) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < DG<ORDER>::NC; ++k)
            FMA_TAB(s, t_t<ORDER>(g, k), cb[k]);
        tr[g] = s;
    }
}
template <int ORDER>
__device__ __forceinline__ void trace_of_top(const double (&ct)[DG<ORDER>::NC], double (&tr)[DG<ORDER>::NG])
{
#pragma unroll
    for (int g = 0; g < DG<ORDER>::NG; ++g) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < DG<ORDER>::NC; ++k)
            FMA_TAB(s, t_b<ORDER>(g, k), ct[k]);
        tr[g] = s;
    }
}

// dphi_i/dt * m_i of one element from its own coefficients c, the four neighbours' traces on its edges (zero
// outside the array), its DG velocity (already divided by hx, hy) and its edge-normal velocities
template <int ORDER>
__device__ __forceinline__ void transport_rhs(const double (&c)[DG<ORDER>::NC], const NbTrace<DG<ORDER>::NG>& nb,
    const double (&vx)[DG<ORDER>::NC], const double (&vy)[DG<ORDER>::NC], const EdgeVel<DG<ORDER>::NG>& E, double ihx, double ihy,
    double (&rhs)[DG<ORDER>::NC])
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG, NQ = DG<ORDER>::NQ;
#pragma unroll
    for (int k = 0; k < NC; ++k)
        rhs[k] = 0.;

    if constexpr (ORDER > 0) { // cell term: int phi v . grad psi_i
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double f = 0., wx = 0., wy = 0.;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                FMA_TAB(f, t_psi<ORDER>(q, k), c[k]);
                FMA_TAB(wx, t_psi<ORDER>(q, k), vx[k]);
                FMA_TAB(wy, t_psi<ORDER>(q, k), vy[k]);
            }
            const double fx = f * wx, fy = f * wy;
#pragma unroll
            for (int i = 1; i < NC; ++i) {
                FMA_TAB(rhs[i], t_w<ORDER>(q) * t_psix<ORDER>(q, i), fx);
                FMA_TAB(rhs[i], t_w<ORDER>(q) * t_psiy<ORDER>(q, i), fy);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double unl = E.l[g], unr = E.r[g], unb = E.b[g], unt = E.t[g];
        double in_r = 0., in_l = 0., in_t = 0., in_b = 0.;
        const double out_r = nb.r[g], out_l = nb.l[g], out_t = nb.t[g], out_b = nb.b[g];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            FMA_TAB(in_r, t_r<ORDER>(g, k), c[k]);
            FMA_TAB(in_l, t_l<ORDER>(g, k), c[k]);
            FMA_TAB(in_t, t_t<ORDER>(g, k), c[k]);
            FMA_TAB(in_b, t_b<ORDER>(g, k), c[k]);
        }
        // upwind fluxes (outward normal velocity is +un on right/top, the flux direction is +x/+y)
        const double fr = (fmax(unr, 0.) * in_r + fmin(unr, 0.) * out_r) * ihx;
        const double fl = (fmax(unl, 0.) * out_l + fmin(unl, 0.) * in_l) * ihx;
        const double ft = (fmax(unt, 0.) * in_t + fmin(unt, 0.) * out_t) * ihy;
        const double fb = (fmax(unb, 0.) * out_b + fmin(unb, 0.) * in_b) * ihy;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            FMA_TAB(rhs[i], -t_gw<ORDER>(g) * t_r<ORDER>(g, i), fr);
            FMA_TAB(rhs[i], t_gw<ORDER>(g) * t_l<ORDER>(g, i), fl);
            FMA_TAB(rhs[i], -t_gw<ORDER>(g) * t_t<ORDER>(g, i), ft);
            FMA_TAB(rhs[i], t_gw<ORDER>(g) * t_b<ORDER>(g, i), fb);
        }
    }
}

// Default stage kernel: one lane per element, all five coefficient sets gathered from memory (the vertical
// neighbours hit L2: the time does not depend on the workgroup height).  The fields advected by the same
// velocity are processed by the same lane, so the DG velocity and the edge velocities are loaded once.
template <int ORDER>
#ifndef NSDG_TR_WAVES
#define NSDG_TR_WAVES 3 // 157 registers with the edge-trace form: 3 waves per SIMD without scratch (4 spill: slower)
#endif
__global__ __launch_bounds__(256, NSDG_TR_WAVES) void transport_stage_kernel(int nx, int ny, int j0, int j1, int nfields, double ihx, double ihy,
    double dt, double a, double b, FieldPtrs fp, const double* __restrict__ vx_dg, const double* __restrict__ vy_dg,
    const double* __restrict__ un_x, const double* __restrict__ un_y)
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG;
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = j0 + blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= nx || iy >= j1)
        return;
    const long N = (long)nx * ny;
    const long e = (long)iy * nx + ix;
    const bool hasL = ix > 0, hasR = ix + 1 < nx, hasB = iy > 0, hasT = iy + 1 < ny;

    // the advecting velocity is shared by all fields: load it once per element
    double vx[NC], vy[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        vx[k] = ORDER > 0 ? vx_dg[k * N + e] * ihx : 0.;
        vy[k] = ORDER > 0 ? vy_dg[k * N + e] * ihy : 0.;
    }
    const long NEX = (long)(nx + 1) * ny, NEY = (long)nx * (ny + 1);
    const long exl = (long)iy * (nx + 1) + ix; // left vertical edge; right = exl + 1
    const long eyb = (long)iy * nx + ix; // bottom horizontal edge; top = eyb + nx
    EdgeVel<NG> E;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        E.l[g] = un_x[g * NEX + exl], E.r[g] = un_x[g * NEX + exl + 1];
        E.b[g] = un_y[g * NEY + eyb], E.t[g] = un_y[g * NEY + eyb + nx];
    }
    for (int f = 0; f < nfields; ++f) {
        const double* __restrict__ phis = fp.phis[f];
        const double* __restrict__ phi0 = fp.phi0[f];
        double* __restrict__ out = fp.out[f];
        double c[NC];
        NbTrace<NG> nb;
        {
            double w[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w[k] = hasL ? phis[k * N + e - 1] : 0.;
            trace_of_left<ORDER>(w, nb.l);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w[k] = hasR ? phis[k * N + e + 1] : 0.;
            trace_of_right<ORDER>(w, nb.r);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w[k] = hasB ? phis[k * N + e - nx] : 0.;
            trace_of_bottom<ORDER>(w, nb.b);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w[k] = hasT ? phis[k * N + e + nx] : 0.;
            trace_of_top<ORDER>(w, nb.t);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k)
            c[k] = phis[k * N + e];
        double rhs[NC];
        transport_rhs<ORDER>(c, nb, vx, vy, E, ihx, ihy, rhs);
        if (a != 0.) {
#pragma unroll
            for (int i = 0; i < NC; ++i)
                out[i * N + e] = rk_update(a, b, dt, IMASS[i], phi0[i * N + e], c[i], rhs[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i)
                out[i * N + e] = rk_update(0., b, dt, IMASS[i], 0., c[i], rhs[i]);
        }
    }
}

// Two-elements-per-lane stage kernel (nsdg_transport_variant_set(ctx, 2, rows); needs an even nx and 16-byte aligned
// arrays, the launcher falls back to the gather kernel otherwise): a lane owns the elements (ix, ix + 1), ix even.
// Every coefficient plane of the pair -- own, below, above, phi0, the DG velocity, the horizontal-edge velocities -- is one
// 16-byte access per lane instead of two 8-byte ones, the neighbour across the inner edge comes from the lane's own
// registers, and only the outer left / right neighbours and the three vertical-edge velocities stay 8-byte accesses:
// 99 vector-memory instructions per pair and two fields against 192 for the two lanes of the gather kernel.  The
// arithmetic per element is the very transport_rhs() of the gather kernel: bit-identical results.
template <int ORDER>
__global__ __launch_bounds__(256, 2) void transport_pair_kernel(int nx, int ny, int j0, int j1, int nfields, double ihx, double ihy, double dt,
    double a, double b, FieldPtrs fp, const double* __restrict__ vx_dg, const double* __restrict__ vy_dg, const double* __restrict__ un_x,
    const double* __restrict__ un_y)
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG;
    const int ix = 2 * (blockIdx.x * 64 + threadIdx.x); // first element of the pair
    const int iy = j0 + blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= nx || iy >= j1)
        return;
    const long N = (long)nx * ny;
    const long e = (long)iy * nx + ix;
    const bool hasL = ix > 0, hasR = ix + 2 < nx, hasB = iy > 0, hasT = iy + 1 < ny;
    auto ld2 = [](const double* p) { return *reinterpret_cast<const double2*>(p); };

    double vx0[NC], vy0[NC], vx1[NC], vy1[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (ORDER > 0) {
            const double2 x = ld2(vx_dg + k * N + e), y = ld2(vy_dg + k * N + e);
            vx0[k] = x.x * ihx, vx1[k] = x.y * ihx, vy0[k] = y.x * ihy, vy1[k] = y.y * ihy;
        } else
            vx0[k] = vx1[k] = vy0[k] = vy1[k] = 0.;
    }
    const long NEX = (long)(nx + 1) * ny, NEY = (long)nx * (ny + 1);
    const long exl = (long)iy * (nx + 1) + ix; // left vertical edge of the first element; + 1 the inner edge; + 2 the right edge of the second
    EdgeVel<NG> E0, E1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double l = un_x[g * NEX + exl], m = un_x[g * NEX + exl + 1], r = un_x[g * NEX + exl + 2];
        const double2 bt = ld2(un_y + g * NEY + e), tp = ld2(un_y + g * NEY + e + nx);
        E0.l[g] = l, E0.r[g] = m, E1.l[g] = m, E1.r[g] = r;
        E0.b[g] = bt.x, E1.b[g] = bt.y, E0.t[g] = tp.x, E1.t[g] = tp.y;
    }
    for (int f = 0; f < nfields; ++f) {
        const double* __restrict__ phis = fp.phis[f];
        const double* __restrict__ phi0 = fp.phi0[f];
        double* __restrict__ out = fp.out[f];
        double c0[NC], c1[NC];
        NbTrace<NG> nb0, nb1;
        {
            double w0[NC], w1[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w0[k] = hasL ? phis[k * N + e - 1] : 0.;
            trace_of_left<ORDER>(w0, nb0.l);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                w1[k] = hasR ? phis[k * N + e + 2] : 0.;
            trace_of_right<ORDER>(w1, nb1.r);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const double2 t = hasB ? ld2(phis + k * N + e - nx) : make_double2(0., 0.);
                w0[k] = t.x, w1[k] = t.y;
            }
            trace_of_bottom<ORDER>(w0, nb0.b);
            trace_of_bottom<ORDER>(w1, nb1.b);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const double2 t = hasT ? ld2(phis + k * N + e + nx) : make_double2(0., 0.);
                w0[k] = t.x, w1[k] = t.y;
            }
            trace_of_top<ORDER>(w0, nb0.t);
            trace_of_top<ORDER>(w1, nb1.t);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const double2 t = ld2(phis + k * N + e);
            c0[k] = t.x, c1[k] = t.y;
        }
        trace_of_right<ORDER>(c1, nb0.r); // across the inner edge: the partner element's coefficients are in registers
        trace_of_left<ORDER>(c0, nb1.l);
        double r0[NC], r1[NC];
        transport_rhs<ORDER>(c0, nb0, vx0, vy0, E0, ihx, ihy, r0);
        transport_rhs<ORDER>(c1, nb1, vx1, vy1, E1, ihx, ihy, r1);
        if (a != 0.) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const double2 p = ld2(phi0 + i * N + e);
                *reinterpret_cast<double2*>(out + i * N + e)
                    = make_double2(rk_update(a, b, dt, IMASS[i], p.x, c0[i], r0[i]), rk_update(a, b, dt, IMASS[i], p.y, c1[i], r1[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i)
                *reinterpret_cast<double2*>(out + i * N + e)
                    = make_double2(rk_update(0., b, dt, IMASS[i], 0., c0[i], r0[i]), rk_update(0., b, dt, IMASS[i], 0., c1[i], r1[i]));
        }
    }
}

// wave-uniform plane base + 32-bit byte offset of the lane: one address register per element instead of a 64-bit pair per load
__device__ __forceinline__ double plane_load(const double* __restrict__ plane, unsigned byte_off)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(plane) + byte_off);
}

// ---------------------------------------------------------------------------------------------------------------
// ALL Runge-Kutta stages of a step in ONE launch, as a MARCH (round 4; nsdg_transport_step_oop[_rows]).  A 512 x 512 DG1 step
// was two stage launches of ~11 us each (BASELINE config 2), a 2048 x 2048 DG2 step three of ~0.24 ms.  Two tile forms were
// tried first -- 32 x 16 tiles with the intermediate stages on a halo in LDS, reading memory directly (14.5 us / 0.63 ms) or
// through LDS-staged inputs (16.1 us / 0.95 ms) -- and lost to this one (11.5 us / 0.32 ms): profiles/r04_transport_fused_forms.md.
// A wave owns a window of 64 - 2 S columns (S = the number of stages) and a
// strip of R rows and walks up the rows.  A lane is an element column; the left / right neighbours' edge traces come from the
// adjacent lanes (DPP), the bottom / top neighbours are the rows the lane itself holds, and stage k runs k rows behind the
// newest row of the field, on the stage k-1 values of the three rows below it -- all in registers: no LDS, no barrier, every
// value loaded once per step (NC field values, 2 NC + 2 NG velocities per element and march step instead of 5 NC + 2 NC + 4 NG
// per element and STAGE).  S lanes on either side and S rows below / above a strip are recomputed instead of exchanged, as in
// the mEVP sub-cycle.  Rows and columns outside the array are read at clamped addresses and ignored through the has* flags of
// their neighbours: the march has no divergent branch.  Same inlined transport_rhs / rk_update on the same operands, with the
// neighbour's trace formed on the neighbour's lane by the same sum: bit-identical to the staged step.
__device__ __forceinline__ double dpp_from_left(double x) // the value of lane - 1 (0 in lane 0)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true); // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_right(double x) // the value of lane + 1 (0 in lane 63)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true); // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int ORDER>
struct March {
    static constexpr int S = ORDER + 1, NC = DG<ORDER>::NC, NG = DG<ORDER>::NG;
    static constexpr int OWN = 64 - 2 * S; // columns a wave owns
    static constexpr int NR0 = S + 1 > 3 ? S + 1 : 3; // rows of the field held: r - NR0 + 1 .. r
};

// velocities that belong to one element row: the element's DG velocity, its left edge, its bottom edge
template <int ORDER>
struct RowVel {
    double vx[DG<ORDER>::NC], vy[DG<ORDER>::NC], el[DG<ORDER>::NG], eb[DG<ORDER>::NG];
};

// L(c) of the row `c` between `below` and `above`; eb_above = the bottom edge of the row above = this row's top edge
template <int ORDER>
__device__ __forceinline__ void march_rhs(const double (&below)[DG<ORDER>::NC], const double (&c)[DG<ORDER>::NC], const double (&above)[DG<ORDER>::NC],
    const RowVel<ORDER>& V, const double (&eb_above)[DG<ORDER>::NG], bool hasL, bool hasR, bool hasB, bool hasT, double ihx, double ihy,
    double (&rhs)[DG<ORDER>::NC])
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG;
    NbTrace<NG> nb;
    EdgeVel<NG> E;
    double mine_r[NG], mine_l[NG];
    trace_of_left<ORDER>(c, mine_r); // my right trace: what my right neighbour calls "the left neighbour's trace"
    trace_of_right<ORDER>(c, mine_l);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double fl = dpp_from_left(mine_r[g]), fr = dpp_from_right(mine_l[g]);
        nb.l[g] = hasL ? fl : 0., nb.r[g] = hasR ? fr : 0.;
        E.l[g] = V.el[g], E.r[g] = dpp_from_right(V.el[g]), E.b[g] = V.eb[g], E.t[g] = eb_above[g];
    }
    // hasB / hasT are wave-uniform (a row of the array): a branch instead of NC selects; the trace of "no neighbour" is the sum over
    // zero coefficients of the other kernels, +0
    if (hasB)
        trace_of_bottom<ORDER>(below, nb.b);
    else {
#pragma unroll
        for (int g = 0; g < NG; ++g)
            nb.b[g] = 0.;
    }
    if (hasT)
        trace_of_top<ORDER>(above, nb.t);
    else {
#pragma unroll
        for (int g = 0; g < NG; ++g)
            nb.t[g] = 0.;
    }
    transport_rhs<ORDER>(c, nb, V.vx, V.vy, E, ihx, ihy, rhs);
}

#ifndef NSDG_MARCH_AHEAD
#define NSDG_MARCH_AHEAD 2 // march steps between the request of a row and its use
#endif
#ifndef NSDG_MARCH_WG_WAVES
#define NSDG_MARCH_WG_WAVES 4 // independent waves per workgroup (fewer, larger workgroups are dispatched faster)
#endif
template <int ORDER>
__global__ __launch_bounds__(64 * NSDG_MARCH_WG_WAVES) void transport_march_kernel(int nx, int ny, int j0, int j1, int R, int ncw, int nwaves, int nfields, double ihx,
    double ihy, double dt, FieldPtrs fp, const double* __restrict__ vx_dg, const double* __restrict__ vy_dg, const double* __restrict__ un_x,
    const double* __restrict__ un_y)
{
    using M = March<ORDER>;
    constexpr int S = M::S, NC = M::NC, NG = M::NG, NR0 = M::NR0;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * NSDG_MARCH_WG_WAVES + (threadIdx.x >> 6); // the waves of a workgroup share nothing
    if (wave >= nwaves)
        return;
    const int strip = wave / ncw, cw = wave - strip * ncw;
    const int y0 = j0 + strip * R, y1 = min(y0 + R, j1); // the rows [j0, j1) of the array are advanced (a row block: its own rows)
    const int x = cw * M::OWN - S + lane;
    const bool hasL = x > 0, hasR = x + 1 < nx;
    const bool own = lane >= S && lane < 64 - S && x < nx;
    const long N = (long)nx * ny, NEX = (long)(nx + 1) * ny, NEY = (long)nx * (ny + 1);
    const int xc = min(max(x, 0), nx - 1), xe = min(max(x, 0), nx);
    // byte offsets of this lane in row `row` of an element plane / of the x-edge plane / of the y-edge plane (clamped rows)
    auto off_el = [&](int row) { return 8u * (unsigned)(min(max(row, 0), ny - 1) * nx + xc); };
    auto off_ex = [&](int row) { return 8u * (unsigned)(min(max(row, 0), ny - 1) * (nx + 1) + xe); };
    auto off_ey = [&](int row) { return 8u * (unsigned)(min(max(row, 0), ny) * nx + xc); };
    for (int f = 0; f < nfields; ++f) {
        const double* __restrict__ phi = fp.phis[f]; // the field at the start of the step (phis == phi0)
        double* __restrict__ out = fp.out[f];
        double PH[NR0][NC]; // the field in the rows r - NR0 + 1 .. r
        double T1[3][NC], T2[3][NC]; // stage 1 in the rows r - 3 .. r - 1, stage 2 (RK3) in the rows r - 4 .. r - 2
        RowVel<ORDER> VR[S]; // VR[k - 1]: the velocities of the row stage k works on, r - k
        double ebn[NG]; // bottom edge of the row r
#pragma unroll
        for (int k = 0; k < NC; ++k) {
#pragma unroll
            for (int i = 0; i < NR0; ++i)
                PH[i][k] = 0.;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                T1[i][k] = T2[i][k] = 0.;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
#pragma unroll
            for (int k = 0; k < NC; ++k)
                VR[j].vx[k] = VR[j].vy[k] = 0.;
#pragma unroll
            for (int g = 0; g < NG; ++g)
                VR[j].el[g] = VR[j].eb[g] = 0.;
        }
        // requested ahead of the step that consumes them: the field in the row r + 1, the velocities of the row r, the bottom edge
        // of the row r + 1 (what the step r + 1 consumes)
        struct Ahead {
            double pn[NC], ebnn[NG];
            RowVel<ORDER> vn; // el, vx, vy as loaded (vx, vy are scaled when they are consumed, not behind the load)
        };
        const int r0 = y0 - S, r1 = y1 - 1 + S;
        auto request = [&](int r, Ahead& Q) {
            const unsigned a = off_el(r + 1), b = off_el(r), c = off_ex(r), d = off_ey(r + 1);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                Q.pn[k] = plane_load(phi + k * N, a);
                Q.vn.vx[k] = ORDER > 0 ? plane_load(vx_dg + k * N, b) : 0.;
                Q.vn.vy[k] = ORDER > 0 ? plane_load(vy_dg + k * N, b) : 0.;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                Q.vn.el[g] = plane_load(un_x + g * NEX, c);
                Q.ebnn[g] = plane_load(un_y + g * NEY, d);
            }
        };
        // the first row any stage works on is r0 + 1 (stage 1, at the step r0 + 2): the rows r0, r0 + 1 and what that step consumes
        // are requested together -- one trip to memory instead of three before the first arithmetic
        {
            const unsigned a0 = off_el(r0), a1 = off_el(r0 + 1), d = off_ey(r0 + 1);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                PH[NR0 - 2][k] = plane_load(phi + k * N, a0), PH[NR0 - 1][k] = plane_load(phi + k * N, a1);
#pragma unroll
            for (int g = 0; g < NG; ++g)
                ebn[g] = plane_load(un_y + g * NEY, d);
        }
        auto step = [&](int r, Ahead& Q) {
            // ---- the rows move down by one; the requested values arrive
#pragma unroll
            for (int k = 0; k < NC; ++k) {
#pragma unroll
                for (int i = 0; i + 1 < NR0; ++i)
                    PH[i][k] = PH[i + 1][k];
                PH[NR0 - 1][k] = Q.pn[k];
                T1[0][k] = T1[1][k], T1[1][k] = T1[2][k];
                T2[0][k] = T2[1][k], T2[1][k] = T2[2][k];
            }
#pragma unroll
            for (int j = S - 1; j >= 1; --j)
                VR[j] = VR[j - 1];
#pragma unroll
            for (int k = 0; k < NC; ++k)
                VR[0].vx[k] = Q.vn.vx[k] * ihx, VR[0].vy[k] = Q.vn.vy[k] * ihy; // row r - 1
#pragma unroll
            for (int g = 0; g < NG; ++g)
                VR[0].el[g] = Q.vn.el[g], VR[0].eb[g] = ebn[g], ebn[g] = Q.ebnn[g]; // ebn was the bottom edge of the row r - 1, becomes that of the row r
            request(r + NSDG_MARCH_AHEAD - 1, Q); // consumed NSDG_MARCH_AHEAD steps from now (past the last step: clamped, unused)
            // ---- stage 1 on the row r - 1: t1 = phi + dt L(phi)
            {
                const int a = r - 1;
                if (a >= max(y0 - (S - 1), 0) && a <= min(y1 - 1 + (S - 1), ny - 1)) { // wave-uniform
                    double rhs[NC];
                    march_rhs<ORDER>(PH[NR0 - 3], PH[NR0 - 2], PH[NR0 - 1], VR[0], ebn, hasL, hasR, a > 0, a + 1 < ny, ihx, ihy, rhs);
                    if (S == 1) {
                        double o[NC];
#pragma unroll
                        for (int i = 0; i < NC; ++i)
                            o[i] = rk_update(0., 1., dt, IMASS[i], 0., PH[NR0 - 2][i], rhs[i]);
                        if (fp.limit) // wave-uniform
                            limit_cell<ORDER>(o, fp.lo[f], fp.hi[f], fp.cap[f] != 0);
                        if (own) {
#pragma unroll
                            for (int i = 0; i < NC; ++i)
                                out[i * N + (long)a * nx + x] = o[i];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NC; ++i)
                            T1[2][i] = rk_update(0., 1., dt, IMASS[i], 0., PH[NR0 - 2][i], rhs[i]);
                    }
                }
            }
            if (S == 2) {
                // ---- Heun on the row r - 2: out = phi / 2 + (t1 + dt L(t1)) / 2
                const int a = r - 2;
                if (a >= y0 && a <= y1 - 1) {
                    double rhs[NC];
                    march_rhs<ORDER>(T1[0], T1[1], T1[2], VR[S - 1], VR[S - 2].eb, hasL, hasR, a > 0, a + 1 < ny, ihx, ihy, rhs);
                    double o[NC];
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                        o[i] = rk_update(0.5, 0.5, dt, IMASS[i], PH[NR0 - 3][i], T1[1][i], rhs[i]);
                    if (fp.limit) // wave-uniform
                        limit_cell<ORDER>(o, fp.lo[f], fp.hi[f], fp.cap[f] != 0);
                    if (own) {
#pragma unroll
                        for (int i = 0; i < NC; ++i)
                            out[i * N + (long)a * nx + x] = o[i];
                    }
                }
            }
            if (S == 3) {
                // ---- Shu-Osher RK3: t2 = 3/4 phi + (t1 + dt L(t1)) / 4 on the row r - 2, out = phi / 3 + 2 (t2 + dt L(t2)) / 3 on the row r - 3
                {
                    const int a = r - 2;
                    if (a >= max(y0 - 1, 0) && a <= min(y1, ny - 1)) {
                        double rhs[NC];
                        march_rhs<ORDER>(T1[0], T1[1], T1[2], VR[1], VR[0].eb, hasL, hasR, a > 0, a + 1 < ny, ihx, ihy, rhs);
#pragma unroll
                        for (int i = 0; i < NC; ++i)
                            T2[2][i] = rk_update(0.75, 0.25, dt, IMASS[i], PH[NR0 - 3][i], T1[1][i], rhs[i]);
                    }
                }
                {
                    const int a = r - 3;
                    if (a >= y0 && a <= y1 - 1) {
                        double rhs[NC];
                        march_rhs<ORDER>(T2[0], T2[1], T2[2], VR[S - 1], VR[S - 2].eb, hasL, hasR, a > 0, a + 1 < ny, ihx, ihy, rhs);
                        double o[NC];
#pragma unroll
                        for (int i = 0; i < NC; ++i)
                            o[i] = rk_update(1. / 3., 2. / 3., dt, IMASS[i], PH[0][i], T2[1][i], rhs[i]);
                        if (fp.limit) // wave-uniform
                            limit_cell<ORDER>(o, fp.lo[f], fp.hi[f], fp.cap[f] != 0);
                        if (own) {
#pragma unroll
                            for (int i = 0; i < NC; ++i)
                                out[i * N + (long)a * nx + x] = o[i];
                        }
                    }
                }
            }
        };
        // NSDG_MARCH_AHEAD request sets in rotation: the set a step has consumed is requested again for the step that many steps
        // later (two: at one wave per SIMD a march step is shorter than a trip to memory)
        Ahead Q[NSDG_MARCH_AHEAD];
#pragma unroll
        for (int i = 0; i < NSDG_MARCH_AHEAD; ++i)
            request(r0 + 1 + i, Q[i]);
        int r = r0 + 2;
        for (; r + NSDG_MARCH_AHEAD - 1 <= r1; r += NSDG_MARCH_AHEAD) {
#pragma unroll
            for (int i = 0; i < NSDG_MARCH_AHEAD; ++i)
                step(r + i, Q[i]);
        }
#pragma unroll
        for (int i = 0; i + 1 < NSDG_MARCH_AHEAD; ++i) {
            if (r + i <= r1)
                step(r + i, Q[i]);
        }
    }
}

// the closure as a pass of its own, in place on the rows [j0, j1): for callers that compose a step from stage launches (the staged
// nsdg_transport_step, a row block whose ghost zones are too shallow for the march).  One lane per element; 2 x NC x 8 bytes per
// element and field against the march's none
template <int ORDER>
__global__ __launch_bounds__(256) void transport_limit_kernel(int nx, int ny, int j0, int j1, int nfields, FieldPtrs fp)
{
    constexpr int NC = DG<ORDER>::NC;
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = j0 + blockIdx.y * 4 + threadIdx.y;
    if (ix >= nx || iy >= j1)
        return;
    const long N = (long)nx * ny, e = (long)iy * nx + ix;
    for (int f = 0; f < nfields; ++f) {
        double* __restrict__ phi = fp.out[f];
        double c[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k)
            c[k] = phi[k * N + e];
        limit_cell<ORDER>(c, fp.lo[f], fp.hi[f], fp.cap[f] != 0);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            phi[k * N + e] = c[k];
    }
}

#ifndef NSDG_MARCH_ROWS
#define NSDG_MARCH_ROWS 0 // rows per strip of the march; 0: from the grid (A/B builds)
#endif
#ifndef NSDG_MARCH_WAVES
#define NSDG_MARCH_WAVES 4 // waves per CU the strips are cut for
#endif
template <int ORDER>
int launch_march(nsdg_ctx* ctx, int j0, int j1, double dt, int nfields, const FieldPtrs& fp, const double* vx, const double* vy, const double* unx,
    const double* uny)
{
    NSDG_CHECK_ARG(8L * (ctx->nx + 1L) * (ctx->ny + 1L) < (1L << 32), "grid too large for 32-bit byte offsets within a coefficient plane");
    using M = March<ORDER>;
    const int ncw = nsdg_div_up(ctx->nx, M::OWN);
    // strips: one wave per SIMD -- a strip of R rows takes R + 2 S - 2 march steps and recomputes up to 2 (S - 1) rows of the
    // earlier stages, more waves than SIMDs make the last ones wait (512 x 512 DG1: 13.3 / 11.5 / 14.3 us per step for
    // 2 / 4 / 8 waves per CU; 2048 x 2048 DG2: 514 / 319 / 330 us) -- at least 4 rows
    int R = NSDG_MARCH_ROWS;
    if (R <= 0) {
        const int want = (NSDG_MARCH_WAVES * ctx->num_cus) / ncw;
        R = nsdg_div_up(j1 - j0, want < 1 ? 1 : want);
        if (R < 4)
            R = 4;
    }
    const int ns = nsdg_div_up(j1 - j0, R);
    const int nwaves = ncw * ns;
    hipLaunchKernelGGL(transport_march_kernel<ORDER>, dim3(nsdg_div_up(nwaves, NSDG_MARCH_WG_WAVES)), dim3(64 * NSDG_MARCH_WG_WAVES), 0, ctx->stream, ctx->nx,
        ctx->ny, j0, j1, R, ncw, nwaves, nfields, 1. / ctx->hx, 1. / ctx->hy, dt, fp, vx, vy, unx, uny);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

// CG2 nodal velocity -> DG velocity (L2 projection) per element, and edge-normal velocities at the
// edge Gauss points.  One lane per element; the lane also owns its left and bottom edge, the last
// column / row additionally writes the right / top boundary edge.
template <int ORDER>
__global__ __launch_bounds__(256) void prepare_advection_kernel(int nx, int ny, const double* __restrict__ u,
    const double* __restrict__ v, double* __restrict__ vx_dg, double* __restrict__ vy_dg, double* __restrict__ un_x,
    double* __restrict__ un_y)
{
    constexpr int NC = DG<ORDER>::NC, NG = DG<ORDER>::NG;
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = blockIdx.y * 4 + threadIdx.y;
    if (ix >= nx || iy >= ny)
        return;
    const long N = (long)nx * ny;
    const long e = (long)iy * nx + ix;
    const int nn = 2 * nx + 1;
    double ul[9], vl[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const long n = (long)(2 * iy + a / 3) * nn + 2 * ix + a % 3;
        ul[a] = u[n];
        vl[a] = v[n];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        double sx = 0., sy = 0.;
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            FMA_TAB(sx, PV[i][a], ul[a]);
            FMA_TAB(sy, PV[i][a], vl[a]);
        }
        vx_dg[i * N + e] = sx;
        vy_dg[i * N + e] = sy;
    }
    const long NEX = (long)(nx + 1) * ny, NEY = (long)nx * (ny + 1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double sl = 0., sr = 0., sb = 0., st = 0.;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            FMA_TAB(sl, t_le<ORDER>(g, k), ul[3 * k]);
            FMA_TAB(sr, t_le<ORDER>(g, k), ul[3 * k + 2]);
            FMA_TAB(sb, t_le<ORDER>(g, k), vl[k]);
            FMA_TAB(st, t_le<ORDER>(g, k), vl[6 + k]);
        }
        un_x[g * NEX + (long)iy * (nx + 1) + ix] = sl;
        if (ix == nx - 1)
            un_x[g * NEX + (long)iy * (nx + 1) + nx] = sr;
        un_y[g * NEY + (long)iy * nx + ix] = sb;
        if (iy == ny - 1)
            un_y[g * NEY + (long)ny * nx + ix] = st;
    }
}

template <int ORDER>
int launch_stage(nsdg_ctx* ctx, int j0, int j1, double dt, double a, double b, int nfields, const FieldPtrs& fp,
    const double* vx, const double* vy, const double* unx, const double* uny)
{
    bool pairs = ctx->transport_variant == 2 && ctx->nx % 2 == 0;
    if (pairs) { // 16-byte accesses: every array 16-byte aligned (plane and row offsets are even because nx is)
        uintptr_t bits = (uintptr_t)vx | (uintptr_t)vy | (uintptr_t)uny;
        for (int f = 0; f < nfields; ++f)
            bits |= (uintptr_t)fp.phi0[f] | (uintptr_t)fp.phis[f] | (uintptr_t)fp.out[f];
        pairs = (bits & 15) == 0;
    }
    if (pairs) {
        const int br = ctx->transport_rows > 0 ? ctx->transport_rows : 4;
        const dim3 block(64, br), grid(nsdg_div_up(ctx->nx / 2, 64), nsdg_div_up(j1 - j0, br));
        hipLaunchKernelGGL(transport_pair_kernel<ORDER>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, nfields, 1. / ctx->hx, 1. / ctx->hy, dt, a,
            b, fp, vx, vy, unx, uny);
    } else {
        // rows per workgroup: the rows above / below a workgroup's band are read a second time by the neighbouring
        // workgroup, so taller bands mean fewer redundant reads (band + 2 rows read per band)
        const int br = ctx->transport_rows > 0 ? ctx->transport_rows : 4;
        const dim3 block(64, br), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(j1 - j0, br));
        hipLaunchKernelGGL(transport_stage_kernel<ORDER>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, nfields, 1. / ctx->hx,
            1. / ctx->hy, dt, a, b, fp, vx, vy, unx, uny);
    }
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int stage_dispatch(nsdg_ctx* ctx, int order, int j0, int j1, double dt, double a, double b, int nfields, const FieldPtrs& fp,
    const double* vx, const double* vy, const double* unx, const double* uny)
{
    switch (order) {
    case 0: return launch_stage<0>(ctx, j0, j1, dt, a, b, nfields, fp, vx, vy, unx, uny);
    case 1: return launch_stage<1>(ctx, j0, j1, dt, a, b, nfields, fp, vx, vy, unx, uny);
    default: return launch_stage<2>(ctx, j0, j1, dt, a, b, nfields, fp, vx, vy, unx, uny);
    }
}

// the context's bounds for the nfields fields of a full step (nsdg_transport_bounds_set); false: the field counts disagree
bool bounds_into(const nsdg_ctx* ctx, int nfields, FieldPtrs& fp)
{
    fp.limit = ctx->nbounds > 0;
    for (int f = 0; f < MAXF; ++f) {
        const int s = (fp.limit && f < ctx->nbounds) ? f : 0;
        fp.lo[f] = fp.limit ? ctx->bounds[s].lo : 0.;
        fp.hi[f] = fp.limit ? ctx->bounds[s].hi : 0.;
        fp.cap[f] = fp.limit ? ctx->bounds[s].cap_mean : 0;
    }
    return !fp.limit || ctx->nbounds == nfields;
}
#define NSDG_BOUNDS_MISMATCH "nsdg_transport_bounds_set was given a different number of fields than this call advances"

} // namespace

extern "C" {

int nsdg_transport_bounds_set(nsdg_ctx* ctx, int32_t nfields, const nsdg_field_bounds* b)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(nfields >= 0 && nfields <= MAXF, "nfields must be 0 (no closure) .. 4");
    NSDG_CHECK_ARG(nfields == 0 || b != nullptr, "null bounds");
    for (int f = 0; f < nfields; ++f) {
        NSDG_CHECK_ARG(b[f].lo <= b[f].hi, "bounds need lo <= hi (hi = +infinity: no upper bound)"); // false for a NaN
        NSDG_CHECK_ARG(!b[f].cap_mean || b[f].hi < HUGE_VAL, "a capped cell mean needs a finite upper bound");
    }
    ctx->nbounds = nfields;
    for (int f = 0; f < nfields; ++f)
        ctx->bounds[f] = b[f];
    return NSDG_OK;
}

int nsdg_transport_limit(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, int32_t nfields, double* const* phi)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(order >= 0 && order <= 2, "order must be 0, 1 or 2");
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(nfields >= 1 && nfields <= MAXF && phi, "nfields must be 1..4");
    if (ctx->nbounds == 0) {
        nsdg_set_error("nsdg_transport_limit: no bounds set (nsdg_transport_bounds_set)");
        return NSDG_ERR_STATE;
    }
    FieldPtrs fp;
    NSDG_CHECK_ARG(bounds_into(ctx, nfields, fp), NSDG_BOUNDS_MISMATCH);
    for (int f = 0; f < MAXF; ++f) {
        const int s = f < nfields ? f : 0;
        NSDG_CHECK_ARG(phi[s] != nullptr, "null field pointer");
        fp.phi0[f] = fp.phis[f] = fp.out[f] = phi[s];
    }
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(j1 - j0, 4));
    if (order == 0)
        hipLaunchKernelGGL(transport_limit_kernel<0>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, nfields, fp);
    else if (order == 1)
        hipLaunchKernelGGL(transport_limit_kernel<1>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, nfields, fp);
    else
        hipLaunchKernelGGL(transport_limit_kernel<2>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, nfields, fp);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_transport_variant_set(nsdg_ctx* ctx, int32_t variant, int32_t strip_rows)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(variant == 0 || variant == 2,
        "variant must be 0 (one element per lane) or 2 (two elements per lane); 1, the marching kernel of rounds 1-2, was removed: never faster");
    NSDG_CHECK_ARG(strip_rows >= 0 && strip_rows <= 4, "rows per workgroup must be in 0..4 (0 = default)");
    ctx->transport_variant = variant;
    ctx->transport_rows = strip_rows;
    return NSDG_OK;
}

int nsdg_prepare_advection(nsdg_ctx* ctx, int32_t order, const double* u, const double* v, double* vx_dg, double* vy_dg,
    double* un_x, double* un_y)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(order >= 0 && order <= 2, "order must be 0, 1 or 2");
    NSDG_CHECK_ARG(u && v && vx_dg && vy_dg && un_x && un_y, "null field pointer");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(ctx->ny, 4));
    if (order == 0)
        hipLaunchKernelGGL(prepare_advection_kernel<0>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, u, v, vx_dg, vy_dg, un_x, un_y);
    else if (order == 1)
        hipLaunchKernelGGL(prepare_advection_kernel<1>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, u, v, vx_dg, vy_dg, un_x, un_y);
    else
        hipLaunchKernelGGL(prepare_advection_kernel<2>, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, u, v, vx_dg, vy_dg, un_x, un_y);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_transport_stage(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, double dt, double a, double b, int32_t nfields,
    const double* const* phi0, const double* const* phis, double* const* out, const double* vx_dg, const double* vy_dg,
    const double* un_x, const double* un_y)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(order >= 0 && order <= 2, "order must be 0, 1 or 2");
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(nfields >= 1 && nfields <= MAXF, "nfields must be 1..4");
    NSDG_CHECK_ARG(phi0 && phis && out && vx_dg && vy_dg && un_x && un_y, "null pointer");
    if (j0 == j1)
        return NSDG_OK;
    FieldPtrs fp;
    for (int f = 0; f < MAXF; ++f) {
        const int s = f < nfields ? f : 0;
        NSDG_CHECK_ARG(phi0[s] && phis[s] && out[s], "null field pointer");
        NSDG_CHECK_ARG(out[s] != phis[s], "out must not alias phis (neighbours are read)");
        fp.phi0[f] = phi0[s];
        fp.phis[f] = phis[s];
        fp.out[f] = out[s];
        fp.lo[f] = fp.hi[f] = 0., fp.cap[f] = 0;
    }
    fp.limit = 0; // a bare stage: the closure belongs to the END of a step (nsdg_transport_limit, or the step entry points)
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    return stage_dispatch(ctx, order, j0, j1, dt, a, b, nfields, fp, vx_dg, vy_dg, un_x, un_y);
}

int nsdg_transport_step(nsdg_ctx* ctx, int32_t order, double dt, int32_t nfields, double* const* phi, const double* vx_dg,
    const double* vy_dg, const double* un_x, const double* un_y, double* scratch)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(order >= 0 && order <= 2, "order must be 0, 1 or 2");
    NSDG_CHECK_ARG(nfields >= 1 && nfields <= MAXF, "nfields must be 1..4");
    NSDG_CHECK_ARG(phi && scratch && vx_dg && vy_dg && un_x && un_y, "null pointer");
    NSDG_CHECK_ARG(ctx->nbounds == 0 || ctx->nbounds == nfields, NSDG_BOUNDS_MISMATCH); // before anything is advanced
    const int nc = order == 0 ? 1 : (order == 1 ? 3 : 6);
    const long M = (long)nc * ctx->nx * ctx->ny;
    const double *p0[MAXF], *ps[MAXF];
    double *t1[MAXF], *t2[MAXF], *ph[MAXF];
    for (int f = 0; f < nfields; ++f) {
        NSDG_CHECK_ARG(phi[f] != nullptr, "null field pointer");
        ph[f] = phi[f];
        t1[f] = scratch + (2 * f) * M;
        t2[f] = scratch + (2 * f + 1) * M;
        p0[f] = phi[f];
    }
    const int ny = ctx->ny;
    int rc;
    // SSP Runge-Kutta of order (order+1): Euler / Heun / Shu-Osher RK3; the last stage writes phi
    if (order == 0) {
        for (int f = 0; f < nfields; ++f) ps[f] = phi[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 0., 1., nfields, p0, ps, t1, vx_dg, vy_dg, un_x, un_y))) return rc;
        for (int f = 0; f < nfields; ++f)
            NSDG_CHECK_HIP(hipMemcpyAsync(phi[f], t1[f], M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    } else if (order == 1) {
        for (int f = 0; f < nfields; ++f) ps[f] = phi[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 0., 1., nfields, p0, ps, t1, vx_dg, vy_dg, un_x, un_y))) return rc;
        // the LAST stage writes phi itself: it reads phi only as phi0, at the element it writes (the neighbours come from t1)
        for (int f = 0; f < nfields; ++f) ps[f] = t1[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 0.5, 0.5, nfields, p0, ps, ph, vx_dg, vy_dg, un_x, un_y))) return rc;
    } else {
        for (int f = 0; f < nfields; ++f) ps[f] = phi[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 0., 1., nfields, p0, ps, t1, vx_dg, vy_dg, un_x, un_y))) return rc;
        for (int f = 0; f < nfields; ++f) ps[f] = t1[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 0.75, 0.25, nfields, p0, ps, t2, vx_dg, vy_dg, un_x, un_y))) return rc;
        for (int f = 0; f < nfields; ++f) ps[f] = t2[f];
        if ((rc = nsdg_transport_stage(ctx, order, 0, ny, dt, 1. / 3., 2. / 3., nfields, p0, ps, ph, vx_dg, vy_dg, un_x, un_y))) return rc;
    }
    if (ctx->nbounds > 0) // the closure of the step (nsdg_transport_bounds_set): one more pass over the new state
        return nsdg_transport_limit(ctx, order, 0, ny, nfields, phi);
    return NSDG_OK;
}

int nsdg_transport_step_oop_rows(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, double dt, int32_t nfields, const double* const* phi_in,
    double* const* phi_out, const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(order >= 0 && order <= 2, "order must be 0, 1 or 2");
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(nfields >= 1 && nfields <= MAXF, "nfields must be 1..4");
    NSDG_CHECK_ARG(phi_in && phi_out && vx_dg && vy_dg && un_x && un_y, "null pointer");
    FieldPtrs fp;
    for (int f = 0; f < MAXF; ++f) {
        const int s = f < nfields ? f : 0;
        NSDG_CHECK_ARG(phi_in[s] && phi_out[s], "null field pointer");
        fp.phi0[f] = fp.phis[f] = phi_in[s];
        fp.out[f] = phi_out[s];
    }
    {
        // the fields of a launch are advanced one after the other by waves that run concurrently: an output that overlaps ANY
        // input (not only its own) or another output would be read half-written -- compared as ranges of nc nx ny doubles
        const long len = (long)(order == 0 ? 1 : (order == 1 ? 3 : 6)) * ctx->nx * ctx->ny;
        auto overlap = [len](const double* a, const double* b) { return a < b + len && b < a + len; };
        for (int i = 0; i < nfields; ++i) {
            for (int j = 0; j < nfields; ++j)
                NSDG_CHECK_ARG(!overlap(phi_out[i], phi_in[j]),
                    "phi_out must not alias or overlap ANY phi_in (other waves read the rows and columns around theirs, and the fields of a launch run concurrently)");
            for (int j = 0; j < i; ++j)
                NSDG_CHECK_ARG(!overlap(phi_out[i], phi_out[j]), "two phi_out arrays alias or overlap");
        }
    }
    NSDG_CHECK_ARG(bounds_into(ctx, nfields, fp), NSDG_BOUNDS_MISMATCH); // the closure runs in the epilogue of the march
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    switch (order) {
    case 0: return launch_march<0>(ctx, j0, j1, dt, nfields, fp, vx_dg, vy_dg, un_x, un_y);
    case 1: return launch_march<1>(ctx, j0, j1, dt, nfields, fp, vx_dg, vy_dg, un_x, un_y);
    default: return launch_march<2>(ctx, j0, j1, dt, nfields, fp, vx_dg, vy_dg, un_x, un_y);
    }
}

int nsdg_transport_step_oop(nsdg_ctx* ctx, int32_t order, double dt, int32_t nfields, const double* const* phi_in, double* const* phi_out,
    const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y)
{
    NSDG_NEED_GRID(ctx);
    return nsdg_transport_step_oop_rows(ctx, order, 0, ctx->ny, dt, nfields, phi_in, phi_out, vx_dg, vy_dg, un_x, un_y);
}

} // extern "C"
