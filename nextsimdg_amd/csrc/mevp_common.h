// mevp_common.h -- per-element / per-node device functions shared by the two mEVP kernel variants
// (mevp.hip: two kernels per sub-iteration; mevp_fused.hip: fused marching kernel).  Using the very
// same inlined functions in both variants keeps their results bit-identical.
#pragma once
#include "dg_tables.h"
#include "nsdg_internal.h"

namespace nsdg_mevp_detail {

using namespace nsdg_tab;

// Tiled layout of the element-wise arrays that only the mEVP kernels touch (stress coefficients,
// Gauss-point ice strength): tiles of 64 consecutive elements of a row, all nc coefficients of a tile
// stored together, and inside a tile the coefficients in PAIRS interleaved by element,
//     tile (iy, tx = ix/64) at ((iy*ntx + tx) * nc) * 64,   ntx = ceil(nx/64)   (rows padded to whole tiles)
//     coefficient c < 2*(nc/2) of element l = ix%64 at  (c/2)*128 + 2*l + c%2
//     the odd last coefficient (nc = 9: c = 8) at        (nc/2)*128 + l
// A lane then reaches every coefficient of its element from ONE base address plus compile-time immediates,
// a wave streams 4 KB of contiguous HBM per array and element row, and every access is a 16-byte
// global_load_dwordx4 / global_store_dwordx4 of two coefficients: vector-memory instructions cost the issuing
// wave the same whether they move 8 or 16 bytes per lane (profiles/r01_vmem_issue_microbench.txt), so pairing
// halves the issue cost of the stress and ice-strength traffic.  Arrays must be 16-byte aligned.
__host__ __device__ __forceinline__ int tiles_per_row(int nx) { return (nx + 63) >> 6; }
__device__ __forceinline__ long tile_off(int ix, int iy, int ntx, int nc)
{
    return ((long)iy * ntx + (ix >> 6)) * (nc * 64) + 2 * (ix & 63); // address of coefficient 0; the pair (2k, 2k+1) is at + 128 k
}
// 8 stress coefficients of one element: four 16-byte accesses
__device__ __forceinline__ void tile_load8(const double* __restrict__ a, long t, double (&c)[8])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double2 v = *reinterpret_cast<const double2*>(a + t + 128 * k);
        c[2 * k] = v.x, c[2 * k + 1] = v.y;
    }
}
__device__ __forceinline__ void tile_store8(double* __restrict__ a, long t, const double (&c)[8])
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<double2*>(a + t + 128 * k) = make_double2(c[2 * k], c[2 * k + 1]);
}
// 9 Gauss-point values of one element: four 16-byte accesses and one 8-byte access (t = tile_off(ix, .., 9), l = ix % 64)
__device__ __forceinline__ void tile_load9(const double* __restrict__ a, long t, int l, double (&c)[9])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double2 v = *reinterpret_cast<const double2*>(a + t + 128 * k);
        c[2 * k] = v.x, c[2 * k + 1] = v.y;
    }
    c[8] = a[t + 512 - l];
}
__device__ __forceinline__ void tile_store9(double* __restrict__ a, long t, int l, const double (&c)[9])
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<double2*>(a + t + 128 * k) = make_double2(c[2 * k], c[2 * k + 1]);
    a[t + 512 - l] = c[8];
}

// Workgroups are dealt round-robin to the 8 XCDs of the chip, each with its own L2.  The three-iteration kernel
// gives every XCD a CONTIGUOUS range of (strip, column-wave) indices, so that the waves that share halo columns and
// strip-boundary rows -- neighbours in that index -- read them through the same L2 (34.0-35.2 -> 33.0-33.3 ms per
// model step, A/B on one box).  The same mapping is neutral for the two-iteration kernel and 3-6 % slower for
// the single-iteration one, which streams at the HBM ceiling: those keep the round-robin order.  Bijective for
// any grid size.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblocks)
{
    const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7;
    return xcd * q + min(xcd, r) + (b >> 3);
}

#define FMA_TAB(acc, tab, val)   \
    do {                         \
        const double t_ = (tab); \
        if (t_ != 0.0)           \
            acc += t_ * (val);   \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Sum-factorised element operators.  Every 2-D basis function is a product of 1-D ones,
//   psi_i(xi,eta) = p_a(xi) p_b(eta),  p0 = 1, p1 = s, p2 = s^2 - 1/12,
//   i -> (a,b):  0:(0,0) 1:(1,0) 2:(0,1) 3:(2,0) 4:(0,2) 5:(1,1) 6:(2,1) 7:(1,2)
//   phi_n(xi,eta) = L_ax(xi) L_ay(eta), n = 3*ay + ax (quadratic Lagrange),
// so the dense 8x9 / 9x8 element matrices of dg_tables.h (DX, DY, PSI_G3, ...) factor into two sweeps
// of 3x3 one-dimensional operators.  Compared with the dense tables this halves the flop count and,
// more importantly on CDNA, needs a handful of distinct fp64 literals instead of ~100 (fp64
// literals live in SGPR pairs; the dense form spilled >300 SGPRs).  The 1-D operators are
//   int p_a L'_n / m_a :  a=0: (-1, 0, 1)        a=1: (4, -8, 4)       a=2: 0
//   int p_a L_n  / m_a :  a=0: (1, 4, 1)/6       a=1: (-1, 0, 1)       a=2: (2, -4, 2)
// and the same without the 1/m_a for the divergence.  tests compare against the oracle's dense,
// quadrature-built operators.
// Division-free elementary functions for the hot loop.  The hardware seeds (v_rcp_f64, v_rsq_f64) are
// refined by Newton steps to full double accuracy (relative error <~ 2e-16, not correctly rounded);
// the library routines add range scaling and special-case handling (v_div_scale/fmas/fixup, v_ldexp,
// v_cmp_class) that the mEVP operands never need: the arguments are finite, positive and far from the
// subnormal range (Delta^2 >= Delta_min^2 = 4e-18; denominators >= rho h_min/dt).
__device__ __forceinline__ double fast_rcp(double x)
{
#ifdef NSDG_OLD_ELEMENTARY
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#else
    // the seed is good to 2^-24 (profiles/r02_seed_accuracy.txt): ONE second-order step r (1 + e + e^2), e = 1 - x r,
    // leaves an error of e^3 = 2^-72 -- 3 instructions instead of the 4 of two Newton steps, same 1.0 ulp maximum
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(__builtin_fma(e, e, e), r, r);
#endif
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
#ifdef NSDG_OLD_ELEMENTARY
    double y = __builtin_amdgcn_rsq(x);
    // y <- y + y*(1 - x y^2)/2, twice
    double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    return y;
#else
    // one third-order step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2: 5 instructions instead of 8, the same 1.24 ulp maximum
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
#endif
}
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = fast_rsqrt(x);
#ifdef NSDG_OLD_ELEMENTARY
    // sqrt(x) = x * rsqrt(x), one Goldschmidt-style correction on the product; exact zero stays zero
    double g = x * y;
    g = __builtin_fma(__builtin_fma(-g, g, x), 0.5 * y, g);
    return x > 0. ? g : 0.;
#else
    // sqrt(x) = x * rsqrt(x) (within 2 ulp: it only scales the ocean drag); exact zero stays zero
    return x > 0. ? x * y : 0.;
#endif
}

// Neighbour-lane exchange of the marching kernels as DPP moves (GFX9 wave_shr:1 / wave_shl:1 shift the
// whole 64-lane wavefront by one lane): one VALU move per dword instead of two ds_bpermute round
// trips through the LDS crossbar with their s_waitcnt.  The lane without a neighbour reads ZERO (bound_ctrl): it is a
// redundant column whose result is discarded.  (Rounds 1-3 let that lane keep its own value, `old` = the source: the
// compiler then needs a copy before every DPP move because the instruction overwrites its destination in place -- 24
// extra moves per element row of a pipeline stage.)
__device__ __forceinline__ double lane_from_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true); // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true); // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr double SF_G = 0.3872983346207417; // Gauss abscissa sqrt(3/5)/2 on [-1/2, 1/2]
constexpr double SF_P2E = 1. / 15.; // p2 at the outer Gauss points
constexpr double SF_P2M = -1. / 12.; // p2 at the middle Gauss point
constexpr double SF_W0 = 5. / 18., SF_W1 = 8. / 18.; // Gauss weights

// DG8 coefficients of d/dxi and d/deta of a CG2 function given by its 9 nodal values
__device__ __forceinline__ void sf_grad(const double (&U)[9], double (&dxi)[8], double (&deta)[8])
{
    double d[3], Q[3], B0[3];
#pragma unroll
    for (int ay = 0; ay < 3; ++ay) {
        const double u0 = U[3 * ay], u1 = U[3 * ay + 1], u2 = U[3 * ay + 2];
        const double sm = u0 + u2;
        d[ay] = u2 - u0; // = (int p0 L') . U          (also int p1 L / m1)
        Q[ay] = sm - 2. * u1; // (int p1 L')/m1 = 4Q,   (int p2 L)/m2 = 2Q
        B0[ay] = (sm + 4. * u1) * (1. / 6.); // int p0 L
    }
    // d/dxi: x-operator L' (a = 0: d, a = 1: 4Q), y-operator L (b = 0, 1, 2)
    {
        const double sd = d[0] + d[2], sq = Q[0] + Q[2];
        dxi[0] = (sd + 4. * d[1]) * (1. / 6.);
        dxi[2] = d[2] - d[0];
        dxi[4] = 2. * (sd - 2. * d[1]);
        dxi[1] = (sq + 4. * Q[1]) * (4. / 6.);
        dxi[5] = 4. * (Q[2] - Q[0]);
        dxi[7] = 8. * (sq - 2. * Q[1]);
        dxi[3] = 0.;
        dxi[6] = 0.;
    }
    // d/deta: x-operator L (a = 0: B0, a = 1: d, a = 2: 2Q), y-operator L' (b = 0, 1)
    deta[0] = B0[2] - B0[0];
    deta[2] = 4. * ((B0[0] + B0[2]) - 2. * B0[1]);
    deta[1] = d[2] - d[0];
    deta[5] = 4. * ((d[0] + d[2]) - 2. * d[1]);
    deta[3] = 2. * (Q[2] - Q[0]);
    deta[6] = 8. * ((Q[0] + Q[2]) - 2. * Q[1]);
    deta[4] = 0.;
    deta[7] = 0.;
}

// values of a DG8 function at the 3x3 Gauss points, V[3*qy + qx].  ZERO names coefficients known to vanish (the compiler
// may not drop 0 * x): 1 = c[3], c[6] (a d/dxi of a biquadratic has no xi^2 part), 2 = c[4], c[7] (a d/deta has no eta^2 part)
template <int ZERO = 0>
__device__ __forceinline__ void sf_eval(const double (&c)[8], double (&V)[9])
{
    double T0[3], T1[3], T2[3]; // T_b[qx] = sum_a C[a][b] p_a(xi_qx)
    // every output is one fused expression: S -+ g*c = fma(-+g, c, S)
    if constexpr (ZERO == 1) {
        T0[0] = c[0] - SF_G * c[1], T0[2] = c[0] + SF_G * c[1], T0[1] = c[0];
        T1[0] = c[2] - SF_G * c[5], T1[2] = c[2] + SF_G * c[5], T1[1] = c[2];
    } else {
        {
            const double S = c[0] + SF_P2E * c[3];
            T0[0] = S - SF_G * c[1], T0[2] = S + SF_G * c[1], T0[1] = c[0] + SF_P2M * c[3];
        }
        {
            const double S = c[2] + SF_P2E * c[6];
            T1[0] = S - SF_G * c[5], T1[2] = S + SF_G * c[5], T1[1] = c[2] + SF_P2M * c[6];
        }
    }
    if constexpr (ZERO == 2) {
#pragma unroll
        for (int qx = 0; qx < 3; ++qx) {
            V[qx] = T0[qx] - SF_G * T1[qx];
            V[6 + qx] = T0[qx] + SF_G * T1[qx];
            V[3 + qx] = T0[qx];
        }
    } else {
        T2[0] = c[4] - SF_G * c[7], T2[2] = c[4] + SF_G * c[7], T2[1] = c[4];
#pragma unroll
        for (int qx = 0; qx < 3; ++qx) {
            const double S = T0[qx] + SF_P2E * T2[qx];
            V[qx] = S - SF_G * T1[qx];
            V[6 + qx] = S + SF_G * T1[qx];
            V[3 + qx] = T0[qx] + SF_P2M * T2[qx];
        }
    }
}

// L2 projection of Gauss-point values t[3*qy+qx] on the DG8 basis: R_i = (1/m_i) sum_q w_q psi_i(q) t_q
__device__ __forceinline__ void sf_project(const double (&t)[9], double (&R)[8])
{
    constexpr double K1 = 12. * SF_W0 * SF_G; // (1/m1) w0 p1(g)
    constexpr double K2S = 180. * SF_W0 * SF_P2E, K2M = 180. * SF_W1 * SF_P2M; // (1/m2) w p2
    double Y0[3], Y1[3], Y2[3]; // Y_b[qx] = (1/m_b) sum_qy w_qy p_b(eta_qy) t(qx,qy)
#pragma unroll
    for (int qx = 0; qx < 3; ++qx) {
        const double t0 = t[qx], t1 = t[3 + qx], t2 = t[6 + qx];
        const double sm = t0 + t2;
        Y0[qx] = SF_W0 * sm + SF_W1 * t1;
        Y1[qx] = K1 * (t2 - t0);
        Y2[qx] = K2S * sm + K2M * t1;
    }
    {
        const double sm = Y0[0] + Y0[2];
        R[0] = SF_W0 * sm + SF_W1 * Y0[1];
        R[1] = K1 * (Y0[2] - Y0[0]);
        R[3] = K2S * sm + K2M * Y0[1];
    }
    {
        const double sm = Y1[0] + Y1[2];
        R[2] = SF_W0 * sm + SF_W1 * Y1[1];
        R[5] = K1 * (Y1[2] - Y1[0]);
        R[6] = K2S * sm + K2M * Y1[1];
    }
    {
        const double sm = Y2[0] + Y2[2];
        R[4] = SF_W0 * sm + SF_W1 * Y2[1];
        R[7] = K1 * (Y2[2] - Y2[0]);
    }
}

// the same projection times a run-time factor f (the adaptive form's 1 / alpha_e): the factor goes into the five constants of the second
// sweep -- 5 multiplications per element (the caller forms them once for the three stress components) instead of 24 on the result
struct ProjScale {
    double w0, w1, k1, k2s, k2m;
};
__device__ __forceinline__ ProjScale proj_scale(double f)
{
    constexpr double K1 = 12. * SF_W0 * SF_G, K2S = 180. * SF_W0 * SF_P2E, K2M = 180. * SF_W1 * SF_P2M;
    return ProjScale { f * SF_W0, f * SF_W1, f * K1, f * K2S, f * K2M };
}
__device__ __forceinline__ void sf_project_scaled(const double (&t)[9], const ProjScale& F, double (&R)[8])
{
    constexpr double K1 = 12. * SF_W0 * SF_G;
    constexpr double K2S = 180. * SF_W0 * SF_P2E, K2M = 180. * SF_W1 * SF_P2M;
    double Y0[3], Y1[3], Y2[3];
#pragma unroll
    for (int qx = 0; qx < 3; ++qx) {
        const double t0 = t[qx], t1 = t[3 + qx], t2 = t[6 + qx];
        const double sm = t0 + t2;
        Y0[qx] = SF_W0 * sm + SF_W1 * t1;
        Y1[qx] = K1 * (t2 - t0);
        Y2[qx] = K2S * sm + K2M * t1;
    }
    {
        const double sm = Y0[0] + Y0[2];
        R[0] = F.w0 * sm + F.w1 * Y0[1];
        R[1] = F.k1 * (Y0[2] - Y0[0]);
        R[3] = F.k2s * sm + F.k2m * Y0[1];
    }
    {
        const double sm = Y1[0] + Y1[2];
        R[2] = F.w0 * sm + F.w1 * Y1[1];
        R[5] = F.k1 * (Y1[2] - Y1[0]);
        R[6] = F.k2s * sm + F.k2m * Y1[1];
    }
    {
        const double sm = Y2[0] + Y2[2];
        R[4] = F.w0 * sm + F.w1 * Y2[1];
        R[7] = F.k1 * (Y2[2] - Y2[0]);
    }
}

// G[3*ay+ax] = int_ref S d/dxi phi_n  for a DG8 function S (x-operator int p_a L', y-operator int p_b L)
__device__ __forceinline__ void sf_gxi(const double (&c)[8], double (&G)[9])
{
    double X0[3], X1[3], X2[3]; // X_b[ax]
    constexpr double T3 = 1. / 3.;
    X0[0] = c[1] * T3 - c[0], X0[1] = c[1] * (-2. * T3), X0[2] = c[1] * T3 + c[0];
    X1[0] = c[5] * T3 - c[2], X1[1] = c[5] * (-2. * T3), X1[2] = c[5] * T3 + c[2];
    X2[0] = c[7] * T3 - c[4], X2[1] = c[7] * (-2. * T3), X2[2] = c[7] * T3 + c[4];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double pr = X0[ax] * (1. / 6.) + X2[ax] * (1. / 90.);
        G[ax] = pr - X1[ax] * (1. / 12.);
        G[3 + ax] = X0[ax] * (4. / 6.) - X2[ax] * (2. / 90.);
        G[6 + ax] = pr + X1[ax] * (1. / 12.);
    }
}

// G[3*ay+ax] = int_ref S d/deta phi_n  (x-operator int p_a L, y-operator int p_b L')
__device__ __forceinline__ void sf_geta(const double (&c)[8], double (&G)[9])
{
    double X0[3], X1[3]; // X_b[ax], b = 0, 1
    {
        const double pr = c[0] * (1. / 6.) + c[3] * (1. / 90.);
        X0[0] = pr - c[1] * (1. / 12.), X0[1] = c[0] * (4. / 6.) - c[3] * (2. / 90.), X0[2] = pr + c[1] * (1. / 12.);
    }
    {
        const double pr = c[2] * (1. / 6.) + c[6] * (1. / 90.);
        X1[0] = pr - c[5] * (1. / 12.), X1[1] = c[2] * (4. / 6.) - c[6] * (2. / 90.), X1[2] = pr + c[5] * (1. / 12.);
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        G[ax] = X1[ax] * (1. / 3.) - X0[ax];
        G[3 + ax] = X1[ax] * (-2. / 3.);
        G[6 + ax] = X1[ax] * (1. / 3.) + X0[ax];
    }
}

// (1/alpha) Proj sigma(v): the projected viscous-plastic stress of one element from its 9 nodal velocities, ALREADY SCALED
// by 1/alpha.  Round 4 (the four-wave pipeline is bound by vector-instruction issue, so instructions are time): sigma is
// linear in the ice strength, so 1/alpha and the 1/2 of "- P/2" are folded into it once (hp = P/(2 alpha): 9 multiplies
// that replace the 9 of P/2 and the 24 of the relaxation); the strain-rate coefficients that vanish identically -- a
// d/dxi of a biquadratic has no xi^2 part, a d/deta no eta^2 part -- are neither formed nor evaluated (28 operations).
__device__ __forceinline__ void stress_projected(const double (&ul)[9], const double (&vl)[9], const double (&P)[9], double ihx,
    double ihy, double ialpha, double dmin2, double (&r11)[8], double (&r12)[8], double (&r22)[8])
{
    double E11[8], E12[8], E22[8];
    {
        double uxi[8], ueta[8], vxi[8], veta[8];
        sf_grad(ul, uxi, ueta);
        sf_grad(vl, vxi, veta);
        const double hihx = 0.5 * ihx, hihy = 0.5 * ihy;
        // d/dxi: entries 3, 6 vanish; d/deta: entries 4, 7 vanish
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool zx = i == 3 || i == 6, zy = i == 4 || i == 7;
            E11[i] = zx ? 0. : uxi[i] * ihx;
            E22[i] = zy ? 0. : veta[i] * ihy;
            E12[i] = zx ? ueta[i] * hihy : (zy ? vxi[i] * hihx : ueta[i] * hihy + vxi[i] * hihx);
        }
    }
    double e11[9], e12[9], e22[9];
    sf_eval<1>(E11, e11);
    sf_eval<0>(E12, e12);
    sf_eval<2>(E22, e22);
    double t11[9], t12[9], t22[9];
    const double hps = 0.5 * ialpha;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        // with a = e11 + e22, hb = (e11 - e22) / 2:  Delta^2 = Dmin^2 + a^2 + hb^2 + e12^2 (= Dmin^2 + 1.25 (e11^2 + e22^2) + 1.5 e11 e22
        // + e12^2 for the ellipse ratio 2) and 5/8 e11 + 3/8 e22 = (a + hb / 2) / 2: every factor is an inline constant of the
        // instruction set, so no literal has to be held in (or moved out of) scalar registers
        const double hp = P[q] * hps; // P / (2 alpha)
        const double a = e11[q] + e22[q], hb = 0.5 * (e11[q] - e22[q]);
        const double d2 = __builtin_fma(hb, hb, __builtin_fma(a, a, __builtin_fma(e12[q], e12[q], dmin2)));
        const double pd = hp * fast_rsqrt(d2); // P / (2 alpha Delta)
        t11[q] = __builtin_fma(pd, __builtin_fma(0.5, hb, a), -hp);
        t22[q] = __builtin_fma(pd, __builtin_fma(-0.5, hb, a), -hp);
        t12[q] = (0.5 * pd) * e12[q];
    }
    sf_project(t11, r11);
    sf_project(t12, r12);
    sf_project(t22, r22);
}

// S <- (1 - 1/alpha) S + r, r = (1/alpha) Proj sigma(v) from stress_projected
__device__ __forceinline__ void stress_relax(double ialpha, const double (&r11)[8], const double (&r12)[8], const double (&r22)[8],
    double (&s11)[8], double (&s12)[8], double (&s22)[8])
{
    // written with an explicit fma so that every kernel variant rounds the relaxation the same way
    // (left to the compiler, the contraction differed between variants by one ulp)
    const double keep = 1. - ialpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s11[i] = __builtin_fma(keep, s11[i], r11[i]);
        s12[i] = __builtin_fma(keep, s12[i], r12[i]);
        s22[i] = __builtin_fma(keep, s22[i], r22[i]);
    }
}

// stress of one element from its 9 nodal velocities: S <- (1-1/alpha) S + (1/alpha) Proj sigma(v)
__device__ __forceinline__ void stress_update(const double (&ul)[9], const double (&vl)[9], const double (&P)[9],
    double ihx, double ihy, double ialpha, double dmin2, double (&s11)[8], double (&s12)[8], double (&s22)[8])
{
    double r11[8], r12[8], r22[8];
    stress_projected(ul, vl, P, ihx, ihy, ialpha, dmin2, r11, r12, r22);
    stress_relax(ialpha, r11, r12, r22, s11, s12, s22);
}

// ---------------------------------------------------------------------------------------------
// LOCAL, SOLUTION-ADAPTIVE alpha and beta (round 6; Kimmritz, Danilov & Losch 2016, "The adaptive EVP method for solving the sea ice
// momentum equation", Ocean Modelling 101; parity unpinned like the rest of the dynamics).  The explicit sub-cycle is stable where
// alpha beta >= c zeta dt / (m |K|); with ONE alpha for the domain that is the rigid limit zeta_max = P / (2 Delta_min) of the
// thickest ice -- tens of thousands on a sub-kilometre mesh, and 120 sub-iterations then move the sub-cycle a fraction of a percent
// of the way.  Adaptive: every element takes the alpha its OWN viscosity of THIS sub-iteration asks for,
//     zeta_e = max_q P_q / (2 Delta_q),   alpha_e = sqrt(max(alpha_min^2, c zeta_e dt / (rho_i h'_c |K|))),
// h'_c = the mass-floor-limited nodal mean thickness at the element's centre node (the packed coefficient [0] of that node: an
// ice-free centre node stores it scaled by 2^100, which gives alpha_min exactly as the oracle's rule states it), and every node takes
//     beta_n = max(alpha_min, max over its adjacent elements of alpha_e h'_c(e) / h'_n)     (alpha_min at an ice-free node):
// alpha_e scaled by the ratio of the element's mass to the node's, so that alpha_e beta_n meets the stability bound of EVERY element-node
// pair -- a light node beside a strong element (the edge of a lead) is what ran away without it (profiles/r06_adaptive_noise.md).  The
// kernels never divide: an element offers q_e = alpha_e h'_c, the node update needs only beta_n h'_n = max(alpha_min h'_n, max q_e).  Where the ice deforms alpha is small and the stress follows the strain rate
// within a few sub-iterations; where it is rigid alpha is the old global value and the ice stays rigid.  Same limit as the uniform form.
struct AdaptConsts {
    double G; // c dt / (rho_i |K|)
    double amin2, amin; // alpha_min^2, alpha_min
};

// (1 / alpha_e) Proj sigma(v) of one element, 1 / alpha_e, and what the element offers its nine nodes: q_e = alpha_e h'_c
__device__ __forceinline__ void stress_projected_adaptive(const double (&ul)[9], const double (&vl)[9], const double (&P)[9], double ihx,
    double ihy, double dmin2, double hc, const AdaptConsts& AC, double (&r11)[8], double (&r12)[8], double (&r22)[8], double& q, double& ialpha)
{
    double E11[8], E12[8], E22[8];
    {
        double uxi[8], ueta[8], vxi[8], veta[8];
        sf_grad(ul, uxi, ueta);
        sf_grad(vl, vxi, veta);
        const double hihx = 0.5 * ihx, hihy = 0.5 * ihy;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool zx = i == 3 || i == 6, zy = i == 4 || i == 7;
            E11[i] = zx ? 0. : uxi[i] * ihx;
            E22[i] = zy ? 0. : veta[i] * ihy;
            E12[i] = zx ? ueta[i] * hihy : (zy ? vxi[i] * hihx : ueta[i] * hihy + vxi[i] * hihx);
        }
    }
    double e11[9], e12[9], e22[9];
    sf_eval<1>(E11, e11);
    sf_eval<0>(E12, e12);
    sf_eval<2>(E22, e22);
    double t11[9], t12[9], t22[9];
    double zmax = 0.;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const double hp = 0.5 * P[q]; // P / 2
        const double a = e11[q] + e22[q], hb = 0.5 * (e11[q] - e22[q]);
        const double d2 = __builtin_fma(hb, hb, __builtin_fma(a, a, __builtin_fma(e12[q], e12[q], dmin2)));
        const double z = hp * fast_rsqrt(d2); // zeta = P / (2 Delta)
        zmax = __builtin_fmax(zmax, z);
        t11[q] = __builtin_fma(z, __builtin_fma(0.5, hb, a), -hp);
        t22[q] = __builtin_fma(z, __builtin_fma(-0.5, hb, a), -hp);
        t12[q] = (0.5 * z) * e12[q];
    }
    const double a2 = __builtin_fmax(AC.amin2, (AC.G * zmax) * fast_rcp(hc));
    ialpha = fast_rsqrt(a2);
    q = (a2 * ialpha) * (hc > 0x1p50 ? hc * 0x1p-100 : hc); // alpha_e times the centre node's h' (stored scaled by 2^100 where that node is ice-free)
    const ProjScale F = proj_scale(ialpha); // 1 / alpha_e rides on the projection's constants
    sf_project_scaled(t11, F, r11);
    sf_project_scaled(t12, F, r12);
    sf_project_scaled(t22, F, r22);
}

// S <- (1 - 1/alpha_e) S + r, r = (1/alpha_e) Proj sigma(v) from stress_projected_adaptive
__device__ __forceinline__ void stress_relax_adaptive(double ialpha, const double (&r11)[8], const double (&r12)[8], const double (&r22)[8],
    double (&s11)[8], double (&s12)[8], double (&s22)[8])
{
    const double keep = 1. - ialpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s11[i] = __builtin_fma(keep, s11[i], r11[i]);
        s12[i] = __builtin_fma(keep, s12[i], r12[i]);
        s22[i] = __builtin_fma(keep, s22[i], r22[i]);
    }
}

// all 18 nodal contributions -(sigma, grad phi_n)_K of one element
__device__ __forceinline__ void node_contrib_all(const double (&s11)[8], const double (&s12)[8], const double (&s22)[8],
    double hx, double hy, double (&cx)[9], double (&cy)[9])
{
    double gx11[9], gy12[9], gx12[9], gy22[9];
    sf_gxi(s11, gx11);
    sf_geta(s12, gy12);
    sf_gxi(s12, gx12);
    sf_geta(s22, gy22);
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        cx[n] = -(hy * gx11[n] + hx * gy12[n]);
        cy[n] = -(hy * gx12[n] + hx * gy22[n]);
    }
}

// -(sigma, grad phi_a)_K for local node A of an element with stress coefficients s11/s12/s22
template <int A>
__device__ __forceinline__ void node_contrib(const double (&s11)[8], const double (&s12)[8], const double (&s22)[8],
    double hx, double hy, double& cx, double& cy)
{
    double gx11 = 0., gy12 = 0., gx12 = 0., gy22 = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        FMA_TAB(gx11, MASS[i] * DX[i][A], s11[i]);
        FMA_TAB(gy12, MASS[i] * DY[i][A], s12[i]);
        FMA_TAB(gx12, MASS[i] * DX[i][A], s12[i]);
        FMA_TAB(gy22, MASS[i] * DY[i][A], s22[i]);
    }
    cx = -(hy * gx11 + hx * gy12);
    cy = -(hy * gx12 + hx * gy22);
}

// Per-step coefficients of the momentum update, packed once per time step by
// mevp_pack_nodal_kernel / mevp_prepare_kernel as 6 doubles per CG2 node, read with three 16-byte loads.
// With h' = max(cgH,h_min), m = rho_ice*h', a = clamp(cgA,0,1), cor = m*f_c:
//   [0] h'                                      [3] c3 = (m/dt)*v0 + a*tau_y + cor*u_ocean
//   [1] cd = a * C_o * rho_o                    [4] u_ocean
//   [2] c2 = (m/dt)*u0 + a*tau_x - cor*v_ocean  [5] v_ocean
// (an ice-free node -- mevp.hip: pack_node -- stores [0]..[3] scaled by 2^100 with a = 1: free drift, the divergence weighted by 2^-100)
// and three launch constants K1 = rho_ice*beta/dt, K2 = rho_ice*(1+beta)/dt, K3 = rho_ice*f_c, so that
// the update of DESIGN.md section 3.2 reads
//   drag = cd*|v_o - v|;  u' = (K1 h' u + c2 + drag*u_o + K3 h' v + div_x/M) / (K2 h' + drag)
//                         v' = (K1 h' v + c3 + drag*v_o - K3 h' u + div_y/M) / (K2 h' + drag)
// Layout (NSDG_NODAL_LAYOUT).  Rounds 1-3: node-major, six consecutive doubles per node, three 16-byte loads at a lane stride of
// 96 bytes; the pair-plane form -- pair k = (c[2k], c[2k+1]) of node n at packed[k * 2 NN + 2 n], lane stride 32 bytes -- was
// 0.4 % slower in the single-wave three-iteration kernel (profiles/r03_fused3_levers.md).  Round 4: in the four-wave
// pipeline all four waves of a CU read these coefficients, 36 instructions per march step that each touch 48 cache lines in the
// node-major form against 16; pair planes measured 0.9349-0.9385 ms per pass against 0.9450-0.9491 node-major and 0.9393-0.9394 for
// planes split by node parity (one contiguous kilobyte per access), alternating runs on one box -> pair planes are the default.
// All three are bit-identical in their results.
#ifndef NSDG_NODAL_LAYOUT
#define NSDG_NODAL_LAYOUT 1
#endif
// NSDG_NODAL_LAYOUT: 0 node-major; 1 three pair planes (lane stride 32 bytes); 2 three pair planes, each split by the parity of
// the node index -- the marching kernels read nodes n0 + 2 lane, so the 64 lanes of every access touch ONE contiguous kilobyte.
// `plane` = doubles between two pair planes (what nodal_plane returns); nodal_off = where pair 0 of node n starts.
__host__ __device__ __forceinline__ long nodal_plane(long nnodes)
{
#if NSDG_NODAL_LAYOUT == 0
    return 2; // pair k at + 2 k
#elif NSDG_NODAL_LAYOUT == 1
    return 2 * nnodes;
#else
    return 4 * ((nnodes + 1) / 2); // [even nodes | odd nodes], two doubles each
#endif
}
__host__ __device__ __forceinline__ long nodal_off(long n, long plane)
{
#if NSDG_NODAL_LAYOUT == 0
    return n * 6;
#elif NSDG_NODAL_LAYOUT == 1
    return 2 * n;
#else
    return (n & 1) * (plane >> 1) + 2 * (n >> 1);
#endif
}

struct NodalConsts {
    double k1, k2, k3;
    double rdt; // rho_ice / dt (the adaptive form: K1 = rdt beta_n, K2 = rdt (1 + beta_n) per node)
};

__device__ __forceinline__ void node_update_packed(const NodalConsts& K, const double (&c)[6], double uu, double vv, double divx,
    double divy, double ilumped, double& un, double& vn)
{
    const double du = c[4] - uu, dv = c[5] - vv;
    const double drag = c[1] * fast_sqrt(du * du + dv * dv);
    const double denom = fast_rcp(K.k2 * c[0] + drag);
    const double c1 = K.k1 * c[0], cor = K.k3 * c[0];
    un = denom * (c1 * uu + c[2] + drag * c[4] + cor * vv + divx * ilumped);
    vn = denom * (c1 * vv + c[3] + drag * c[5] - cor * uu + divy * ilumped);
}

// the same with the node's own beta (adaptive form): qmax = the largest offer alpha_e h'_c of the adjacent elements, beta_n h'_n = max(alpha_min
// h'_n, qmax) -- at an ice-free node c[0] is h'_n scaled by 2^100 and the first argument wins: beta_n = alpha_min
__device__ __forceinline__ void node_update_packed_adaptive(const NodalConsts& K, const double (&c)[6], double uu, double vv, double divx,
    double divy, double ilumped, double qmax, double amin, double& un, double& vn)
{
    const double du = c[4] - uu, dv = c[5] - vv;
    const double drag = c[1] * fast_sqrt(du * du + dv * dv);
    const double bh = __builtin_fmax(amin * c[0], qmax); // beta_n h'_n
    const double denom = fast_rcp(__builtin_fma(K.rdt, bh + c[0], drag));
    const double c1 = K.rdt * bh, cor = K.k3 * c[0];
    un = denom * (c1 * uu + c[2] + drag * c[4] + cor * vv + divx * ilumped);
    vn = denom * (c1 * vv + c[3] + drag * c[5] - cor * uu + divy * ilumped);
}

// launch constants of the velocity update / of the adaptive alpha from the context's parameters and the packing's time step
static inline NodalConsts nsdg_nodal_consts(const nsdg_ctx* ctx)
{
    const nsdg_mevp_params& P = ctx->mevp;
    return NodalConsts { P.rho_ice * P.beta / ctx->pack_dt, P.rho_ice * (1. + P.beta) / ctx->pack_dt, P.rho_ice * P.fc, P.rho_ice / ctx->pack_dt };
}
static inline bool nsdg_adaptive(const nsdg_ctx* ctx) { return ctx->mevp.aevp_c > 0.; }
static inline AdaptConsts nsdg_adapt_consts(const nsdg_ctx* ctx)
{
    const nsdg_mevp_params& P = ctx->mevp;
    return AdaptConsts { P.aevp_c * ctx->pack_dt / (P.rho_ice * ctx->hx * ctx->hy), P.aevp_alpha_min * P.aevp_alpha_min, P.aevp_alpha_min };
}

// plane = nodal_plane(number of nodes of the local array)
__device__ __forceinline__ void load_nodal(const double* __restrict__ packed, long plane, long n, double (&c)[6])
{
    const double* p = packed + nodal_off(n, plane);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double2 t = *reinterpret_cast<const double2*>(p + k * plane);
        c[2 * k] = t.x;
        c[2 * k + 1] = t.y;
    }
}

__device__ __forceinline__ void store_nodal(double* __restrict__ packed, long plane, long n, const double (&c)[6])
{
    double* p = packed + nodal_off(n, plane);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        *reinterpret_cast<double2*>(p + k * plane) = make_double2(c[2 * k], c[2 * k + 1]);
}

} // namespace nsdg_mevp_detail
