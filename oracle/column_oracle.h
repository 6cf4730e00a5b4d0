/* column_oracle.h -- TEST INFRASTRUCTURE (see column_oracle.c). */
#ifndef ORACLE_COLUMN_H
#define ORACLE_COLUMN_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_ALBEDO_SMU = 0, ORACLE_ALBEDO_SMU2 = 1, ORACLE_ALBEDO_CCSM = 2 };
enum { ORACLE_FREEZING_LINEAR = 0, ORACLE_FREEZING_UNESCO = 1 };

/* Same field order as nsdg_column_params in include/nsdg.h (tests rely on it). */
typedef struct {
    double drag_ocean_q, drag_ocean_t, drag_ice_t, ocean_albedo, i0, min_conc, min_thick;
    double ks;
    double h0, phi_m;
    double ccsm_ice_albedo, ccsm_snow_albedo;
    int flooding;
    int albedo_kind;
    int freezing_kind;
    int reserved;
} oracle_column_params;

enum {
    ORACLE_D_RHO = 0,
    ORACLE_D_QA,
    ORACLE_D_QW,
    ORACLE_D_QI,
    ORACLE_D_CSPEC,
    ORACLE_D_TAU,
    ORACLE_D_HI,
    ORACLE_D_HS,
    ORACLE_D_CNEW,
    ORACLE_D_QIA,
    ORACLE_D_QIO,
    ORACLE_D_SUBL,
    ORACLE_D_DQDT,
    ORACLE_D_HIFROMS,
    ORACLE_D_QOW,
    ORACLE_NDIAG
};

void oracle_column_default_params(oracle_column_params* p);
double oracle_freezing_point(int kind, double sss);
double oracle_albedo(const oracle_column_params* p, double temperature, double snow_thickness);
void oracle_column_element(const oracle_column_params* P, double dt, double* H, double* c, double* Hs,
    double* T, double sst, double sss, double tair, double tdew, double slp, double qsw, double qlw,
    double mld, double snowfall, double wind, double* newice, double* diag);
void oracle_column_step(const oracle_column_params* P, long n, double dt, double* hice, double* cice,
    double* hsnow, double* tice0, const double* sst, const double* sss, const double* tair,
    const double* tdew, const double* slp, const double* qsw, const double* qlw, const double* mld,
    const double* snowfall, const double* wind, double* newice, double* diag);

#ifdef __cplusplus
}
#endif
#endif
