// column_step.hip -- the reference's per-element column-physics step as one gfx950 kernel.
//
// What it replaces: the element loop of DevStep::iterate (core/src/DevStep.cpp:14-23) and everything
// it calls per element: IPhysics1d::updateDerivedData, NextsimPhysics::calculate
// (open-water / ice-atmosphere / ice-ocean fluxes, ThermoIce0, new-ice formation, Hibler lateral
// growth, min-c/h cut-off) and PrognosticData::updateAndIntegrate.  Formulae are cited inline
// (paths relative to /root/reference).
//
// Shape: two adjacent elements per lane (16-byte accesses), SoA planes, 15 coalesced loads + 5 stores per element
// (160 B / element-step algorithmic traffic); all intermediates (the reference's
// PhysicsData scratch and the NextsimPhysics members) live in registers.  The virtual plugin calls
// of the reference (albedo / freezing point) become wave-uniform switches on the params struct,
// which is passed by value as a kernel argument (SGPRs).  HBM-bound: ~0.5 kflop incl. 4 exp per
// 160 B.
#include "nsdg_internal.h"

namespace {

// core/src/include/constants.hpp:21-120
constexpr double SIGMA = 5.670374419e-8;
constexpr double ICE_EPSILON = 0.996;
constexpr double ICE_KAPPA = 2.0334;
constexpr double ICE_LF = 333.55e3;
constexpr double ICE_RHO = 917.;
constexpr double ICE_RHOSNOW = 330.;
constexpr double ICE_S = 5.;
constexpr double AIR_CP = 1004.64;
constexpr double AIR_RA = 287.058;
constexpr double VAP_CP = 1860.;
constexpr double VAP_LV0 = 2500.79e3;
constexpr double VAP_RA = 461.5;
constexpr double WATER_CP = 4186.84;
constexpr double WATER_MU = 0.055;
constexpr double WATER_RHOOCEAN = 1025.;
constexpr double WATER_TF = 273.15;

__device__ __forceinline__ double kelvin(double c) { return c + WATER_TF; } // constants.hpp:128

// Divisions.  The kernel is bound by fp64 issue, not by HBM, as long as its ~35 divisions per element go through
// the IEEE sequence (v_div_scale x2, v_rcp, 6 fma, v_div_fmas, v_div_fixup = 13 instructions each, more than half
// of the instruction stream).  They are replaced by
//   qdiv(a, b)      hardware reciprocal seed + two Newton steps + one residual correction of the quotient (8 instr.)
//   rdiv(a, b, 1/b) the same correction with a reciprocal that is known already: b is a compile-time constant or
//                   a launch parameter such as dt (3 instructions)
// Both return the quotient to within one ulp (the residual step makes it the correctly rounded one in all but
// a few per mille of the cases), far inside the 1e-13 parity tolerance, and end in the IEEE sequence's own last
// instruction, v_div_fixup_f64, which restores the special classes: x/0 = Inf, 0/0 = Inf/Inf = NaN, Inf/x = Inf,
// x/Inf = 0, NaN in -> NaN out.  (Round 2 left the fix-up out: with qlw = Inf the snow melt rate -Inf / L became NaN
// in the residual step, fmin / fmax swallowed the NaN, and the element kept its ice where the reference's -Inf
// thickness zeroes it -- the 1.88-against-0 of profiles/r03_column_cutoff_cause.md.)  What they still drop is the
// range SCALING: a subnormal denominator, or one above 2^1022, whose reciprocal is not representable, yields NaN.
// They are therefore used only where the denominator is a normal number for every input of the documented domain
// (include/nsdg.h): saturation-pressure and density formulae, the albedo weights, physical constants, and
// conc + del_c >= min_conc, which the reference tests itself.  The divisions whose denominator is a FREE input
// or a difference of data -- the concentration and the true thickness, the mixed-layer heat capacity (mld), deltaTml,
// the slab conductance and the surface-temperature Newton step, and dt -- are IEEE divisions, so that mld == 0, dt == 0 or a vanishing flux
// give the reference's Inf / 0 / NaN, bit for bit (tests: zero-denominator cases against the oracle).
#ifdef NSDG_COLUMN_NO_FIXUP // A/B builds only: the round-2 form without the special-case fix-up (what it costs)
#define NSDG_DIV_FIXUP(q, b, a) (q)
#else
#define NSDG_DIV_FIXUP(q, b, a) __builtin_amdgcn_div_fixup((q), (b), (a))
#endif
__device__ __forceinline__ double qdiv(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return NSDG_DIV_FIXUP(__builtin_fma(__builtin_fma(-b, q, a), r, q), b, a);
}
__device__ __forceinline__ double rdiv(double a, double b, double rb)
{
    const double q = a * rb;
    return NSDG_DIV_FIXUP(__builtin_fma(__builtin_fma(-b, q, a), rb, q), b, a);
}
#define CDIV(a, c) rdiv((a), (c), 1.0 / (c)) /* c is a compile-time constant: 1/c is folded */

// NextsimPhysics::SpecificHumidity parameter sets, NextsimPhysics.cpp:310,346,324-325
struct SpHum {
    double a, b, c, d, A, B, C;
};
constexpr SpHum SH_WATER = { 6.1121e2, 18.729, 257.87, 227.3, 7.2e-4, 3.20e-6, 5.9e-10 };
constexpr SpHum SH_ICE = { 6.1115e2, 23.036, 279.82, 333.7, 2.2e-4, 3.83e-6, 6.4e-10 };
constexpr double SH_ALPHA = 0.62197;
constexpr double SH_BETA = 1 - 0.62197;

__device__ __forceinline__ double sh_f(const SpHum& s, double t, double pPa) // :371-375
{
    return 1 + s.A + (pPa * 0.01) * (s.B + s.C * t * t);
}
__device__ __forceinline__ double sh_est(const SpHum& s, double t, double sal) // :377-381
{
    return s.a * exp(qdiv((s.b - CDIV(t, s.d)) * t, t + s.c)) * (1 - 5.37e-4 * sal);
}
__device__ __forceinline__ double sh_q(double est, double f, double p) // :335-343
{
    return qdiv(SH_ALPHA * f * est, p - SH_BETA * f * est);
}

__device__ __forceinline__ double freezing_point(int kind, double sss)
{
    // core/src/modules/include/UnescoFreezing.hpp:28-38 / LinearFreezing.hpp:30-34
    if (kind == NSDG_FREEZING_UNESCO)
        return sss * (-0.0575 + 1.710523e-3 * sqrt(sss) + -2.154996e-4 * sss);
    return -WATER_MU * sss;
}

__device__ __forceinline__ double ice_albedo(const nsdg_column_params& P, double temperature, double hs)
{
    constexpr double ICE_ALBEDO = 0.64, SNOW_ALBEDO = 0.85;
    if (P.albedo_kind == NSDG_ALBEDO_CCSM) { // physics/src/modules/CCSMIceAlbedo.cpp:28-36
        const double iceAlbedoT = P.ccsm_ice_albedo - fmax(0., 0.075 * (temperature + 1.));
        const double snowAlbedoT = P.ccsm_snow_albedo - fmax(0., 0.124 * (temperature + 1.));
        const double f = qdiv(hs, hs + 0.02);
        return f * snowAlbedoT + (1 - f) * iceAlbedoT;
    }
    const double bare = ICE_ALBEDO + 0.4 * (1 - ICE_ALBEDO) * P.i0;
    if (P.albedo_kind == NSDG_ALBEDO_SMU2) // SMU2IceAlbedo.cpp:21-29
        return (hs > 0.) ? fmin(SNOW_ALBEDO, ICE_ALBEDO + CDIV((SNOW_ALBEDO - ICE_ALBEDO) * hs, 0.2)) : bare;
    return (hs > 0.) ? SNOW_ALBEDO : bare; // SMUIceAlbedo.cpp:19-26
}

struct ColumnIn {
    double thick, conc, snow, tice, sst, sss, tair, tdew, slp, qsw, qlw, mld, snowfall, wind, newice;
};
struct ColumnOut {
    double hice, cice, hsnow, tice0, newice;
};

// The column physics of ONE element (everything DevStep::iterate does to it), on register values.
template <bool DIAG>
__device__ __forceinline__ ColumnOut column_element(const nsdg_column_params& P, double dt, const ColumnIn& in, double* d /* NSDG_NDIAG or unused */)
{
    const double thick = in.thick, conc = in.conc, snow = in.snow, tice = in.tice;
    const double sst = in.sst, sss = in.sss, tair = in.tair, tdew = in.tdew, slp = in.slp;
    const double qsw = in.qsw, qlw = in.qlw, mld = in.mld, snowfall = in.snowfall, wind = in.wind;
    double newice = in.newice;

    // PrognosticData.hpp:56,75,78; ExternalData.hpp:60
    // true thicknesses: IEEE divisions (conc is a free input; a subnormal concentration must not turn into NaN)
    const double h_true = (conc != 0) ? thick / conc : 0;
    const double hs_true = (conc != 0) ? snow / conc : 0;
    const double idt = 1.0 / dt; // launch parameter: one IEEE division per lane
    // x / dt: the 3-instruction form when dt is an ordinary number (wave-uniform test), IEEE otherwise (dt == 0 -> +-Inf)
    const bool dt_regular = isnormal(dt) && isnormal(idt);
    auto div_dt = [&](double a) { return dt_regular ? rdiv(a, dt, idt) : a / dt; };
    const double tf = freezing_point(P.freezing_kind, sss);
    const double mlbhc = mld * WATER_RHOOCEAN * WATER_CP;

    // ---- updateDerivedData (IPhysics1d.hpp:33-45; NextsimPhysics.cpp:85-114)
    const double q_a = sh_q(sh_est(SH_WATER, tdew, 0.), sh_f(SH_WATER, tdew, slp), slp);
    const double q_w = sh_q(sh_est(SH_WATER, sst, sss), sh_f(SH_WATER, sst, slp), slp);
    const double est_i = sh_est(SH_ICE, tice, 0.);
    const double f_i = sh_f(SH_ICE, tice, slp);
    const double q_i = sh_q(est_i, f_i, slp);
    const double Ra_wet = qdiv(AIR_RA, 1 - q_a * (1 - VAP_RA / AIR_RA));
    const double rho = qdiv(slp, Ra_wet * kelvin(tair));
    const double cspec = AIR_CP + q_a * VAP_CP;
    double hs = hs_true;
    double hi = h_true;

    // ---- NextsimPhysics::calculate (:116-131)
    const double evap = P.drag_ocean_q * rho * wind * (q_w - q_a); // :133-137
    const double tau = rho * (1e-3 * fmax(1.0, fmin(2.0, 0.61 + 0.063 * wind))); // :139-142,291-295
    // open water heat flux :144-162
    const double Lw = VAP_LV0 + sst * (-2.36418e3 + sst * (1.58927 + sst * (-6.14342e-2))); // :297-302
    const double sstK = kelvin(sst), sstK2 = sstK * sstK;
    const double Qlhow = evap * Lw;
    const double Qshow = P.drag_ocean_t * rho * cspec * wind * (sst - tair);
    const double Qswow = -qsw * (1 - P.ocean_albedo);
    const double Qlwow = ICE_EPSILON * SIGMA * (sstK2 * sstK2) - qlw; // :383-386
    double Qow = Qlhow + Qshow + Qlwow + Qswow;
    // ice-atmosphere :164-198
    const double subl = P.drag_ice_t * rho * wind * (q_i - q_a);
    const double Li = VAP_LV0 + ICE_LF - 240. + tice * (-290. + tice * (-4.)); // :304-307
    const double Qlhi = subl * Li;
    double dq_dT;
    { // SpecificHumidityIce::dq_dT :356-368 (df_dT written exactly as in the reference)
        const double df_dT = 2 * SH_ICE.C * SH_ICE.B * tice;
        const double ct = SH_ICE.c + tice;
        const double dest_dT = qdiv(SH_ICE.b * SH_ICE.c * SH_ICE.d - tice * (2 * SH_ICE.c + tice), SH_ICE.d * (ct * ct)) * est_i;
        const double den = slp - SH_BETA * est_i * f_i;
        dq_dT = qdiv(SH_ALPHA * slp * (f_i * dest_dT + est_i * df_dT), den * den);
    }
    const double dQlh_dT = Li * (P.drag_ice_t * rho * wind * dq_dT);
    const double Qshi = P.drag_ice_t * rho * cspec * wind * (tice - tair);
    const double dQsh_dT = P.drag_ice_t * rho * cspec * wind;
    const double albedoValue = ice_albedo(P, tice, (conc > 0) ? hs_true : 0.); // :183 snow / conc for conc > 0
    const double Qswi = -qsw * (1. - P.i0) * (1 - albedoValue);
    const double ticeK = kelvin(tice), ticeK2 = ticeK * ticeK;
    const double sb_i = ICE_EPSILON * SIGMA * (ticeK2 * ticeK2);
    const double Qlwi = sb_i - qlw;
    const double dQlw_dT = qdiv(4, ticeK) * sb_i;
    const double Qia = Qlhi + Qshi + Qlwi + Qswi;
    const double dQ_dT = dQlh_dT + dQsh_dT + dQlw_dT;
    // ice-ocean :222-226 -> BasicIceOceanHeatFlux.cpp:16-25
    double Qio = div_dt((sst - tf) * mlbhc); // BasicIceOceanHeatFlux.cpp:24

    // ---- massFluxIceOcean :200-220
    double hifroms = 0;
    double Tnew;
    { // ThermoIce0::calculate, ThermoIce0.cpp:34-133
        constexpr double freezingPointIce = -WATER_MU * ICE_S;
        constexpr double bulkLHFusionSnow = ICE_LF * ICE_RHOSNOW;
        constexpr double bulkLHFusionIce = ICE_LF * ICE_RHO;
        if (thick == 0 || conc == 0) { // :45-51
            hi = 0;
            hs = 0;
            Tnew = freezingPointIce;
        } else {
            const double k_lSlab = (P.ks * ICE_KAPPA) / (P.ks * h_true + ICE_KAPPA * hs_true); // :58-59 (IEEE: data-dependent denominator)
            const double QIceConduction = k_lSlab * (tf - tice); // :60
            const double remainingFlux = QIceConduction - Qia; // :61
            Tnew = tice + remainingFlux / (k_lSlab + dQ_dT); // :62-63 (IEEE)
            Tnew = fmin((hs_true > 0.) ? 0. : freezingPointIce, Tnew); // :66-68
            const double snowMeltRate = CDIV(fmin(-remainingFlux, 0.), bulkLHFusionSnow); // :71
            const double snowSublRate = CDIV(subl, ICE_RHOSNOW); // :72
            hs += (snowMeltRate - snowSublRate) * dt; // :74
            const double excessIceMelt = CDIV(fmin(hs, 0.) * bulkLHFusionSnow, bulkLHFusionIce); // :76-77
            hs = fmax(hs, 0.); // :79
            hs += CDIV(snowfall * dt, ICE_RHOSNOW); // :81
            const double iceBottomChange = CDIV((QIceConduction - Qio) * dt, bulkLHFusionIce); // :84-85
            hi += excessIceMelt + iceBottomChange; // :87-88
            const double iceDraught = CDIV(hi * ICE_RHO + hs * ICE_RHOSNOW, WATER_RHOOCEAN); // :95-97
            if (P.flooding && iceDraught > hi) { // :98-106
                const double newIce = iceDraught - hi;
                hifroms += newIce;
                hi = iceDraught;
                hs -= CDIV(newIce * ICE_RHO, ICE_RHOSNOW);
            }
            if (hi < P.min_thick) { // :108-132
                hifroms = 0;
                Qio += div_dt(hi * bulkLHFusionIce) + div_dt(hs * bulkLHFusionSnow);
                hi = 0;
                hs = 0;
                Tnew = freezingPointIce;
            }
        }
    }
    { // newIceFormation :228-254; newice keeps its old value when the branch is not taken (A.7 quirk 1)
        const double coolingFlux = Qow;
        const double deltaTml = -coolingFlux / mlbhc * dt; // :236 (IEEE: mld == 0 gives -+Inf as in the reference)
        const double t1 = sst + deltaTml;
        if (t1 < tf) {
            const double sensibleFlux = (tf - sst) / deltaTml * coolingFlux; // :241 (IEEE)
            const double latentFlux = coolingFlux - sensibleFlux;
            Qow = sensibleFlux;
            newice = CDIV(latentFlux * dt * (1 - conc), ICE_LF * ICE_RHO);
        }
    }
    double c_new;
    { // lateralGrowth :262-289 with HiblerConcentration.cpp:32-47
        double del_c = newice * (1. / P.h0);
        if (hi < h_true && !(conc >= 1))
            del_c += (hi - h_true) * conc * P.phi_m / h_true; // HiblerConcentration.cpp:40-47 (IEEE: h is data)
        c_new = conc + del_c;
        if (c_new >= P.min_conc) {
            const double rc = qdiv(1.0, conc + del_c);
            hi += rdiv(newice - hi * del_c, conc + del_c, rc); // updateThickness :257-260
            if (del_c < 0)
                Qow -= div_dt(del_c * hs * ICE_LF * ICE_RHOSNOW);
            else
                hs += rdiv(0. - hs * del_c, conc + del_c, rc);
        }
    }
    if (c_new < P.min_conc || hi < P.min_thick) { // :211-219
        Qow += div_dt(c_new * ICE_LF * (hi * ICE_RHO + hs * ICE_RHOSNOW));
        c_new = 0;
        hi = 0;
        hs = 0;
    }

    if (DIAG) {
        d[NSDG_D_RHO] = rho;
        d[NSDG_D_QA] = q_a;
        d[NSDG_D_QW] = q_w;
        d[NSDG_D_QI] = q_i;
        d[NSDG_D_CSPEC] = cspec;
        d[NSDG_D_TAU] = tau;
        d[NSDG_D_HI] = hi;
        d[NSDG_D_HS] = hs;
        d[NSDG_D_CNEW] = c_new;
        d[NSDG_D_QIA] = Qia;
        d[NSDG_D_QIO] = Qio;
        d[NSDG_D_SUBL] = subl;
        d[NSDG_D_DQDT] = dQ_dT;
        d[NSDG_D_HIFROMS] = hifroms;
        d[NSDG_D_QOW] = Qow;
    }
    // ---- PrognosticData::updateAndIntegrate (core/src/PrognosticData.cpp:63-71; PhysicsData.hpp:61,66)
    return ColumnOut { hi * c_new, c_new, hs * c_new, Tnew, newice };
}

// Scalar kernel (one element per lane): used when diagnostics are requested and for the odd tail element.
template <bool DIAG>
__global__ __launch_bounds__(256) void column_step_kernel(nsdg_column_params P, long first, long n, long nplane, double dt,
    double* __restrict__ hice, double* __restrict__ cice, double* __restrict__ hsnow,
    double* __restrict__ tice0, const double* __restrict__ sst_, const double* __restrict__ sss_,
    const double* __restrict__ tair_, const double* __restrict__ tdew_, const double* __restrict__ slp_,
    const double* __restrict__ qsw_, const double* __restrict__ qlw_, const double* __restrict__ mld_,
    const double* __restrict__ snowfall_, const double* __restrict__ wind_, double* __restrict__ newice_,
    double* __restrict__ diag)
{
    const long e = first + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n)
        return;
    const ColumnIn in = { hice[e], cice[e], hsnow[e], tice0[e], sst_[e], sss_[e], tair_[e], tdew_[e], slp_[e], qsw_[e], qlw_[e], mld_[e],
        snowfall_[e], wind_[e], newice_[e] };
    double d[NSDG_NDIAG];
    const ColumnOut o = column_element<DIAG>(P, dt, in, d);
    hice[e] = o.hice;
    cice[e] = o.cice;
    hsnow[e] = o.hsnow;
    tice0[e] = o.tice0;
    newice_[e] = o.newice;
    if (DIAG) {
#pragma unroll
        for (int k = 0; k < NSDG_NDIAG; ++k)
            diag[(long)k * nplane + e] = d[k];
    }
}

// Production kernel: two adjacent elements per lane, every plane access a 16-byte load/store (the
// widest coalesced access; 8-byte accesses reach a lower fraction of the HBM rate).
#ifndef NSDG_COL_WAVES
#define NSDG_COL_WAVES 1
#endif
// Every plane is read once and five are written once: streaming (non-temporal) accesses, NSDG_COL_NT bits 1 loads, 2 stores.
// 4096^2 elements, three alternations on one box: 0 0.522-0.535 ms, 1 0.511-0.516, 2 0.521-0.524, 3 0.510-0.518: 3 is the default.
#ifndef NSDG_COL_NT
#define NSDG_COL_NT 3
#endif
typedef double col_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 col_load(const double2* p)
{
    if (NSDG_COL_NT & 1) {
        const col_pair v = __builtin_nontemporal_load(reinterpret_cast<const col_pair*>(p));
        return make_double2(v.x, v.y);
    }
    return *p;
}
__device__ __forceinline__ void col_store(double2* p, double a, double b)
{
    if (NSDG_COL_NT & 2) {
        col_pair v;
        v.x = a, v.y = b;
        __builtin_nontemporal_store(v, reinterpret_cast<col_pair*>(p));
    } else
        *p = make_double2(a, b);
}
__global__ __launch_bounds__(256, NSDG_COL_WAVES) void column_step_kernel_x2(nsdg_column_params P, long npairs, double dt,
    double2* __restrict__ hice, double2* __restrict__ cice, double2* __restrict__ hsnow, double2* __restrict__ tice0,
    const double2* __restrict__ sst_, const double2* __restrict__ sss_, const double2* __restrict__ tair_,
    const double2* __restrict__ tdew_, const double2* __restrict__ slp_, const double2* __restrict__ qsw_,
    const double2* __restrict__ qlw_, const double2* __restrict__ mld_, const double2* __restrict__ snowfall_,
    const double2* __restrict__ wind_, double2* __restrict__ newice_)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs)
        return;
    const double2 a0 = col_load(hice + i), a1 = col_load(cice + i), a2 = col_load(hsnow + i), a3 = col_load(tice0 + i), a4 = col_load(sst_ + i),
                  a5 = col_load(sss_ + i), a6 = col_load(tair_ + i), a7 = col_load(tdew_ + i), a8 = col_load(slp_ + i), a9 = col_load(qsw_ + i),
                  a10 = col_load(qlw_ + i), a11 = col_load(mld_ + i), a12 = col_load(snowfall_ + i), a13 = col_load(wind_ + i),
                  a14 = col_load(newice_ + i);
    const ColumnIn inx = { a0.x, a1.x, a2.x, a3.x, a4.x, a5.x, a6.x, a7.x, a8.x, a9.x, a10.x, a11.x, a12.x, a13.x, a14.x };
    const ColumnIn iny = { a0.y, a1.y, a2.y, a3.y, a4.y, a5.y, a6.y, a7.y, a8.y, a9.y, a10.y, a11.y, a12.y, a13.y, a14.y };
    const ColumnOut ox = column_element<false>(P, dt, inx, nullptr);
    const ColumnOut oy = column_element<false>(P, dt, iny, nullptr);
    col_store(hice + i, ox.hice, oy.hice);
    col_store(cice + i, ox.cice, oy.cice);
    col_store(hsnow + i, ox.hsnow, oy.hsnow);
    col_store(tice0 + i, ox.tice0, oy.tice0);
    col_store(newice_ + i, ox.newice, oy.newice);
}

} // namespace

extern "C" int nsdg_column_step(nsdg_ctx* ctx, int64_t n, double dt, double* hice, double* cice, double* hsnow,
    double* tice0, const double* sst, const double* sss, const double* tair, const double* tdew,
    const double* slp, const double* qsw, const double* qlw, const double* mld, const double* snowfall,
    const double* wind, double* newice, double* diag)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(n >= 0, "negative element count");
    if (n == 0)
        return NSDG_OK;
    NSDG_CHECK_ARG(hice && cice && hsnow && tice0 && sst && sss && tair && tdew && slp && qsw && qlw && mld
            && snowfall && wind && newice,
        "null field pointer");
    NSDG_CHECK_ARG(n < (1LL << 31) * 256, "element count too large for one launch");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(256);
    if (diag) {
        hipLaunchKernelGGL(column_step_kernel<true>, dim3(nsdg_div_up(n, 256)), block, 0, ctx->stream, ctx->column, 0L, (long)n, (long)n, dt,
            hice, cice, hsnow, tice0, sst, sss, tair, tdew, slp, qsw, qlw, mld, snowfall, wind, newice, diag);
    } else {
        // 16-byte path for the even part when every plane is 16-byte aligned, scalar kernel for the rest
        const double* planes[15] = { hice, cice, hsnow, tice0, sst, sss, tair, tdew, slp, qsw, qlw, mld, snowfall, wind, newice };
        bool aligned = true;
        for (const double* p : planes)
            aligned = aligned && (((uintptr_t)p & 15) == 0);
        const long npairs = aligned ? n / 2 : 0;
        if (npairs > 0)
            hipLaunchKernelGGL(column_step_kernel_x2, dim3(nsdg_div_up(npairs, 256)), block, 0, ctx->stream, ctx->column, npairs, dt,
                (double2*)hice, (double2*)cice, (double2*)hsnow, (double2*)tice0, (const double2*)sst, (const double2*)sss,
                (const double2*)tair, (const double2*)tdew, (const double2*)slp, (const double2*)qsw, (const double2*)qlw,
                (const double2*)mld, (const double2*)snowfall, (const double2*)wind, (double2*)newice);
        const long done = 2 * npairs;
        if (done < n)
            hipLaunchKernelGGL(column_step_kernel<false>, dim3(nsdg_div_up(n - done, 256)), block, 0, ctx->stream, ctx->column, done, (long)n,
                (long)n, dt, hice, cice, hsnow, tice0, sst, sss, tair, tdew, slp, qsw, qlw, mld, snowfall, wind, newice, diag);
    }
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}
