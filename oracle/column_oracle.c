/*
 * column_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, one scalar loop) of the reference's per-element column-physics step,
 * i.e. the body of DevStep::iterate (core/src/DevStep.cpp:14-23):
 *     updateDerivedData -> NextsimPhysics::calculate -> PrognosticData::updateAndIntegrate.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (nextsimdg_amd/) never does.
 *
 * Pinning: checked by tests/test_oracle_column.py against every known-answer value the
 * reference's own tests hold for this path (physics/test/NextsimPhysics_test.cpp:73-77,123,
 * 160-172,229-240,298-309; core/test/ElementData_test.cpp:76-86) and against the 17-digit
 * probe values recorded in SURVEY.md Appendix C.  The reference itself is NOT buildable in this
 * image (needs Boost.program_options + generated .ipp files), see DESIGN.md section 4.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Operation order is kept as written in the reference so that a CPU run is as close to the
 * reference's own floating-point result as the same libm allows.
 */
#include <math.h>
#include <stddef.h>

#include "column_oracle.h"

#ifdef ORACLE_OMP
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#else
#define OMP_FOR
#endif

/* core/src/include/constants.hpp:21-120 */
static const double SIGMA = 5.670374419e-8; /* :21 */
static const double ICE_EPSILON = 0.996; /* :43 */
static const double ICE_KAPPA = 2.0334; /* :46 */
static const double ICE_LF = 333.55e3; /* :49 */
static const double ICE_RHO = 917.; /* :56 */
static const double ICE_RHOSNOW = 330.; /* :63 */
static const double ICE_S = 5.; /* :66 */
static const double AIR_CP = 1004.64; /* :76 */
static const double AIR_RA = 287.058; /* :79 */
static const double VAP_CP = 1860.; /* :89 */
static const double VAP_LV0 = 2500.79e3; /* :92 */
static const double VAP_RA = 461.5; /* :95 */
static const double WATER_CP = 4186.84; /* :102 */
static const double WATER_MU = 0.055; /* :111 */
static const double WATER_RHOOCEAN = 1025.; /* :117 */
static const double WATER_TF = 273.15; /* :120 (= Ice::Tm :69) */

static double kelvin(double c) { return c + WATER_TF; } /* constants.hpp:128 */

void oracle_column_default_params(oracle_column_params* p)
{
    /* physics/src/modules/NextsimPhysics.cpp:76-82 */
    p->drag_ocean_q = 1.5e-3;
    p->drag_ocean_t = 0.83e-3;
    p->drag_ice_t = 1.3e-3;
    p->ocean_albedo = 0.07;
    p->i0 = 0.17;
    p->min_conc = 1e-12;
    p->min_thick = 0.01;
    /* physics/src/modules/ThermoIce0.cpp:30-31 */
    p->ks = 0.3096;
    p->flooding = 1;
    /* physics/src/modules/HiblerConcentration.cpp:28-29 */
    p->h0 = 0.25;
    p->phi_m = 0.5;
    /* physics/src/modules/CCSMIceAlbedo.cpp:22-23,40-41 */
    p->ccsm_ice_albedo = 0.538;
    p->ccsm_snow_albedo = 0.8256;
    /* first implementation listed is the default: physics/src/modules/modules.json:4-8,
     * core/src/modules/modules.json:4-7, core/src/ModuleLoader.cpp:51-54 */
    p->albedo_kind = ORACLE_ALBEDO_SMU;
    p->freezing_kind = ORACLE_FREEZING_LINEAR;
}

/* core/src/modules/include/LinearFreezing.hpp:30-34, UnescoFreezing.hpp:28-38 */
double oracle_freezing_point(int kind, double sss)
{
    if (kind == ORACLE_FREEZING_UNESCO) {
        const double a0 = -0.0575, a1 = +1.710523e-3, a2 = -2.154996e-4, b = -7.53e-4, p0 = 0;
        return sss * (a0 + a1 * sqrt(sss) + a2 * sss) + b * p0;
    }
    return -WATER_MU * sss;
}

/* physics/src/modules/{SMUIceAlbedo.cpp:19-26, SMU2IceAlbedo.cpp:21-29, CCSMIceAlbedo.cpp:28-36} */
double oracle_albedo(const oracle_column_params* p, double temperature, double snow_thickness)
{
    const double ICE_ALBEDO = 0.64, SNOW_ALBEDO = 0.85;
    switch (p->albedo_kind) {
    case ORACLE_ALBEDO_CCSM: {
        const double tLimit = -1.;
        double iceAlbedoT = p->ccsm_ice_albedo - fmax(0., 0.075 * (temperature - tLimit));
        double snowAlbedoT = p->ccsm_snow_albedo - fmax(0., 0.124 * (temperature - tLimit));
        double snowCoverFraction = snow_thickness / (snow_thickness + 0.02);
        return snowCoverFraction * snowAlbedoT + (1 - snowCoverFraction) * iceAlbedoT;
    }
    case ORACLE_ALBEDO_SMU2:
        if (snow_thickness > 0.)
            return fmin(SNOW_ALBEDO, ICE_ALBEDO + (SNOW_ALBEDO - ICE_ALBEDO) * snow_thickness / 0.2);
        return ICE_ALBEDO + 0.4 * (1 - ICE_ALBEDO) * p->i0;
    default: /* SMU */
        if (snow_thickness > 0.)
            return SNOW_ALBEDO;
        return ICE_ALBEDO + 0.4 * (1 - ICE_ALBEDO) * p->i0;
    }
}

/* NextsimPhysics::SpecificHumidity, physics/src/modules/NextsimPhysics.cpp:309-381 */
typedef struct {
    double a, b, c, d, A, B, C, alpha, beta;
} sphum_t;
static const sphum_t SH_WATER = { 6.1121e2, 18.729, 257.87, 227.3, 7.2e-4, 3.20e-6, 5.9e-10, 0.62197, 1 - 0.62197 }; /* :310,:324-325 */
static const sphum_t SH_ICE = { 6.1115e2, 23.036, 279.82, 333.7, 2.2e-4, 3.83e-6, 6.4e-10, 0.62197, 1 - 0.62197 }; /* :346 */

static double sh_f(const sphum_t* s, double t, double pPa) /* :371-375 */
{
    double pressure_mb = pPa * 0.01;
    return 1 + s->A + pressure_mb * (s->B + s->C * t * t);
}
static double sh_est(const sphum_t* s, double t, double sal) /* :377-381 */
{
    double salFactor = 1 - 5.37e-4 * sal;
    return s->a * exp((s->b - t / s->d) * t / (t + s->c)) * salFactor;
}
static double sh_q(const sphum_t* s, double t, double p, double sal) /* :335-343 */
{
    double estCalc = sh_est(s, t, sal);
    double fCalc = sh_f(s, t, p);
    return s->alpha * fCalc * estCalc / (p - s->beta * fCalc * estCalc);
}
static double sh_dq_dT(const sphum_t* s, double t, double p) /* :356-368, written as in the reference */
{
    double df_dT = 2 * s->C * s->B * t;
    double numerator = s->b * s->c * s->d - t * (2 * s->c + t);
    double denominator = s->d * pow(s->c + t, 2);
    double estCalc = sh_est(s, t, 0);
    double fCalc = sh_f(s, t, p);
    double dest_dT = numerator / denominator * estCalc;
    numerator = s->alpha * p * (fCalc * dest_dT + estCalc * df_dT);
    denominator = pow(p - s->beta * estCalc * fCalc, 2);
    return numerator / denominator;
}

static double drag_ocean_m(double w) { return 1e-3 * fmax(1.0, fmin(2.0, 0.61 + 0.063 * w)); } /* :291-295 */
static double latent_heat_water(double t) /* :297-302 */
{
    return VAP_LV0 + t * (-2.36418e3 + t * (1.58927 + t * (-6.14342e-2)));
}
static double latent_heat_ice(double t) { return VAP_LV0 + ICE_LF - 240. + t * (-290. + t * (-4.)); } /* :304-307 */
static double stefan_boltzmann(double tC) { return ICE_EPSILON * SIGMA * pow(kelvin(tC), 4); } /* :383-386 */

/* One element, one step.  State in/out: *H (hice, cell mean), *c (cice), *Hs (hsnow, cell mean),
 * *T (tice[0]); *newice is the per-element persistent m_newice. */
void oracle_column_element(const oracle_column_params* P, double dt, double* H, double* c, double* Hs,
    double* T, double sst, double sss, double tair, double tdew, double slp, double qsw, double qlw,
    double mld, double snowfall, double wind, double* newice, double* diag /* ORACLE_NDIAG or NULL */)
{
    const double thick = *H, conc = *c, snow = *Hs, tice = *T;
    /* PrognosticData.hpp:56,75,78 */
    const double h_true = (conc != 0) ? thick / conc : 0;
    const double hs_true = (conc != 0) ? snow / conc : 0;
    const double tf = oracle_freezing_point(P->freezing_kind, sss);
    const double mlbhc = mld * WATER_RHOOCEAN * WATER_CP; /* ExternalData.hpp:60 */

    /* ---- IPhysics1d::updateDerivedData, physics/src/modules/include/IPhysics1d.hpp:33-45 */
    const double q_a = sh_q(&SH_WATER, tdew, slp, 0); /* NextsimPhysics.cpp:85-88 */
    const double q_w = sh_q(&SH_WATER, sst, slp, sss); /* :90-96 */
    const double q_i = sh_q(&SH_ICE, tice, slp, 0); /* :98-102 */
    const double Ra_wet = AIR_RA / (1 - q_a * (1 - VAP_RA / AIR_RA)); /* :106 */
    const double rho = slp / (Ra_wet * kelvin(tair)); /* :107 */
    const double cspec = AIR_CP + q_a * VAP_CP; /* :113 */
    double hs = hs_true; /* IPhysics1d.hpp:43 */
    double hi = h_true; /* :44 */

    /* ---- NextsimPhysics::calculate, NextsimPhysics.cpp:116-131 */
    const double evap = P->drag_ocean_q * rho * wind * (q_w - q_a); /* :133-137 */
    const double tau = rho * drag_ocean_m(wind); /* :139-142 */
    /* heatFluxOpenWater :144-162 */
    const double Qlhow = evap * latent_heat_water(sst);
    const double Qshow = P->drag_ocean_t * rho * cspec * wind * (sst - tair);
    const double Qswow = -qsw * (1 - P->ocean_albedo);
    const double Qlwow = stefan_boltzmann(sst) - qlw;
    double Qow = Qlhow + Qshow + Qlwow + Qswow;
    /* massFluxIceAtmosphere :164-168 */
    const double subl = P->drag_ice_t * rho * wind * (q_i - q_a);
    /* heatFluxIceAtmosphere :170-198 */
    const double Qlhi = subl * latent_heat_ice(tice);
    const double dmdot_dT = P->drag_ice_t * rho * wind * sh_dq_dT(&SH_ICE, tice, slp);
    const double dQlh_dT = latent_heat_ice(tice) * dmdot_dT;
    const double Qshi = P->drag_ice_t * rho * cspec * wind * (tice - tair);
    const double dQsh_dT = P->drag_ice_t * rho * cspec * wind;
    const double albedoValue = oracle_albedo(P, tice, (conc > 0) ? (snow / conc) : 0.);
    const double Qswi = -qsw * (1. - P->i0) * (1 - albedoValue);
    const double Qlwi = stefan_boltzmann(tice) - qlw;
    const double dQlw_dT = 4 / kelvin(tice) * stefan_boltzmann(tice);
    const double Qia = Qlhi + Qshi + Qlwi + Qswi;
    const double dQ_dT = dQlh_dT + dQsh_dT + dQlw_dT;
    /* heatFluxIceOcean :222-226 -> BasicIceOceanHeatFlux.cpp:16-25 */
    double Qio = (sst - tf) * mlbhc / dt;

    /* ---- massFluxIceOcean :200-220 */
    double hifroms = 0;
    double Tnew;
    double c_new;
    { /* ThermoIce0::calculate, physics/src/modules/ThermoIce0.cpp:34-133 */
        const double freezingPointIce = -WATER_MU * ICE_S;
        const double bulkLHFusionSnow = ICE_LF * ICE_RHOSNOW;
        const double bulkLHFusionIce = ICE_LF * ICE_RHO;
        if (thick == 0 || conc == 0) { /* :45-51 */
            hi = 0;
            hs = 0;
            Tnew = freezingPointIce;
        } else {
            const double k_lSlab = P->ks * ICE_KAPPA / (P->ks * h_true + ICE_KAPPA * hs_true); /* :58-59 */
            const double QIceConduction = k_lSlab * (tf - tice); /* :60 */
            const double remainingFlux = QIceConduction - Qia; /* :61 */
            Tnew = tice + remainingFlux / (k_lSlab + dQ_dT); /* :62-63 */
            const double meltingLimit = (hs_true > 0.) ? 0 : freezingPointIce; /* :66 */
            Tnew = fmin(meltingLimit, Tnew); /* :67-68 */
            const double snowMeltRate = fmin(-remainingFlux, 0.) / bulkLHFusionSnow; /* :71 */
            const double snowSublRate = subl / ICE_RHOSNOW; /* :72 */
            hs += (snowMeltRate - snowSublRate) * dt; /* :74 */
            const double excessIceMelt = fmin(hs, 0.) * bulkLHFusionSnow / bulkLHFusionIce; /* :76-77 */
            hs = fmax(hs, 0.); /* :79 */
            hs += snowfall * dt / ICE_RHOSNOW; /* :81 */
            const double iceBottomChange = (QIceConduction - Qio) * dt / bulkLHFusionIce; /* :84-85 */
            const double iceThicknessChange = excessIceMelt + iceBottomChange; /* :87 */
            hi += iceThicknessChange; /* :88 */
            const double iceDraught = (hi * ICE_RHO + hs * ICE_RHOSNOW) / WATER_RHOOCEAN; /* :95-97 */
            if (P->flooding && iceDraught > hi) { /* :98-106 */
                const double newIce = iceDraught - hi;
                hifroms += newIce;
                hi = iceDraught;
                hs -= newIce * ICE_RHO / ICE_RHOSNOW;
            }
            if (hi < P->min_thick) { /* :108-132 (topMelt/botMelt scaling has no side effect) */
                hifroms = 0;
                const double deltaQio = hi * bulkLHFusionIce / dt + hs * bulkLHFusionSnow / dt;
                Qio += deltaQio;
                hi = 0;
                hs = 0;
                Tnew = freezingPointIce;
            }
        }
    }
    { /* newIceFormation :228-254 */
        const double coolingFlux = Qow;
        const double deltaTml = -coolingFlux / mlbhc * dt;
        const double t0 = sst;
        const double t1 = t0 + deltaTml;
        if (t1 < tf) {
            const double sensibleFlux = (tf - t0) / deltaTml * coolingFlux;
            const double latentFlux = coolingFlux - sensibleFlux;
            Qow = sensibleFlux;
            *newice = latentFlux * dt * (1 - conc) / (ICE_LF * ICE_RHO);
        }
    }
    { /* lateralGrowth :262-289 */
        double del_c = 0;
        const double ooh0 = 1. / P->h0; /* HiblerConcentration.cpp:36 (latched static there) */
        del_c += (*newice) * ooh0; /* freeze, HiblerConcentration.cpp:32-38 */
        if (hi < h_true) { /* melt, HiblerConcentration.cpp:40-47 */
            if (!(conc >= 1)) {
                const double del_hi = hi - h_true;
                del_c += del_hi * conc * P->phi_m / h_true;
            }
        }
        c_new = conc + del_c; /* :274 */
        if (c_new >= P->min_conc) { /* :276-288 */
            hi += (*newice - hi * del_c) / (conc + del_c); /* updateThickness :257-260,278 */
            if (del_c < 0) {
                Qow -= del_c * hs * ICE_LF * ICE_RHOSNOW / dt; /* :282-283 */
            } else {
                hs += (0. - hs * del_c) / (conc + del_c); /* :286 */
            }
        }
    }
    if (c_new < P->min_conc || hi < P->min_thick) { /* :211-219 */
        Qow += c_new * ICE_LF * (hi * ICE_RHO + hs * ICE_RHOSNOW) / dt;
        c_new = 0;
        hi = 0;
        hs = 0;
    }

    /* ---- PrognosticData::updateAndIntegrate, core/src/PrognosticData.cpp:63-71 with
     *      PhysicsData.hpp:61,66,69,76.  sst/sss are not touched. */
    *H = hi * c_new;
    *c = c_new;
    *Hs = hs * c_new;
    *T = Tnew;

    if (diag) {
        diag[ORACLE_D_RHO] = rho;
        diag[ORACLE_D_QA] = q_a;
        diag[ORACLE_D_QW] = q_w;
        diag[ORACLE_D_QI] = q_i;
        diag[ORACLE_D_CSPEC] = cspec;
        diag[ORACLE_D_TAU] = tau;
        diag[ORACLE_D_HI] = hi;
        diag[ORACLE_D_HS] = hs;
        diag[ORACLE_D_CNEW] = c_new;
        diag[ORACLE_D_QIA] = Qia;
        diag[ORACLE_D_QIO] = Qio;
        diag[ORACLE_D_SUBL] = subl;
        diag[ORACLE_D_DQDT] = dQ_dT;
        diag[ORACLE_D_HIFROMS] = hifroms;
        diag[ORACLE_D_QOW] = Qow;
    }
}

/* The element loop of DevStep::iterate (core/src/DevStep.cpp:17-22) over flat SoA arrays.
 * diag, when not NULL, is ORACLE_NDIAG planes of n doubles: diag[k*n + e]. */
void oracle_column_step(const oracle_column_params* P, long n, double dt, double* hice, double* cice,
    double* hsnow, double* tice0, const double* sst, const double* sss, const double* tair,
    const double* tdew, const double* slp, const double* qsw, const double* qlw, const double* mld,
    const double* snowfall, const double* wind, double* newice, double* diag)
{
    OMP_FOR
    for (long e = 0; e < n; ++e) {
        double d[ORACLE_NDIAG];
        oracle_column_element(P, dt, &hice[e], &cice[e], &hsnow[e], &tice0[e], sst[e], sss[e], tair[e],
            tdew[e], slp[e], qsw[e], qlw[e], mld[e], snowfall[e], wind[e], &newice[e], diag ? d : NULL);
        if (diag)
            for (int k = 0; k < ORACLE_NDIAG; ++k)
                diag[(size_t)k * n + e] = d[k];
    }
}
