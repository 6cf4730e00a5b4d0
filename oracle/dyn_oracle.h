/* dyn_oracle.h -- TEST INFRASTRUCTURE (see dyn_oracle.c).  PARITY UNPINNED: the reference snapshot
 * holds no dynamics code; this restates the published mathematics only. */
#ifndef ORACLE_DYN_H
#define ORACLE_DYN_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as nsdg_mevp_params in include/nsdg.h. */
typedef struct {
    double rho_ice, rho_atm, rho_ocean;
    double c_atm, c_ocean;
    double pstar, compaction;
    double delta_min;
    double fc;
    double alpha, beta;
    double h_min; /* lower limit applied to the nodal mean thickness in the momentum equation */
    /* ice-free-node rule (round 5; shape of the column model's cut-off, physics/src/modules/NextsimPhysics.cpp:210-219:
     * c_new < minc || hi < minh): a node whose mean concentration is below min_conc, whose TRUE thickness cgH / cgA is below
     * min_thick or whose mean thickness is at the mass floor h_min (its mass would be made up) is in free drift -- full exposure (a = 1) to wind and ocean drag, Coriolis, its floor mass -- and the stress
     * divergence of the neighbouring elements is weighted by 2^-100 there.  Both 0: rule off. */
    double min_conc, min_thick;
    /* local, solution-adaptive alpha and beta (round 6; after Kimmritz, Danilov & Losch 2016): aevp_c > 0 replaces the uniform alpha,
     * beta: per sub-iteration and element  zeta_e = max over its 9 Gauss points of P / (2 Delta),
     * alpha_e = sqrt(max(aevp_alpha_min^2, aevp_c zeta_e dt / (rho_ice max(cgH_c, h_min) hx hy)))  with cgH_c the nodal mean thickness
     * at the element's centre node (alpha_e = aevp_alpha_min if that node is ice-free by the rule above),
     * S <- (1 - 1/alpha_e) S + (1/alpha_e) Proj sigma, and per node, with h'_n = max(cgH_n, h_min),
     * beta_n = max(aevp_alpha_min, max over the adjacent elements of alpha_e h'_c(e) / h'_n)  -- alpha_e scaled by the ratio of the element's
     * mass to the node's, so that alpha_e beta_n satisfies the stability bound of EVERY element-node pair (a light node beside a strong
     * element: the edge of a lead) -- and beta_n = aevp_alpha_min at an ice-free node. */
    double aevp_c, aevp_alpha_min;
} oracle_mevp_params;

void oracle_dyn_init(void);
void oracle_mevp_default_params(oracle_mevp_params* p);

/* number of DG coefficients for order 0/1/2 */
int oracle_dg_ncoef(int order);

void oracle_prepare_advection(int nx, int ny, int order, const double* u, const double* v,
    double* vx_dg, double* vy_dg, double* un_x, double* un_y);

void oracle_transport_stage(int nx, int ny, int j0, int j1, double hx, double hy, int order, double dt,
    double a, double b, const double* phi0, const double* phis, double* out, const double* vx_dg,
    const double* vy_dg, const double* un_x, const double* un_y);

/* full SSP-RK(order+1) step on all rows; scratch = 2 * ncoef * nx * ny doubles */
void oracle_transport_step(int nx, int ny, double hx, double hy, int order, double dt, double* phi,
    const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y, double* scratch);

/* Closure of the transport (round 5): after a step, on element rows [j0, j1), in place.  cap != 0: a cell mean above hi is set
 * to hi (for the concentration with hi = 1: convergence beyond a closed cover turns into thickness -- the mean thickness, the
 * conserved volume, is a field of its own and is not touched).  Then the Zhang-Shu scaling limiter: the higher coefficients are
 * scaled by the largest theta in [0, 1] for which the values at the scheme's quadrature points -- (order+1)^2 volume Gauss
 * points, the order+1 Gauss points of each of the four edges and the four corners -- lie in [lo, hi]; the cell mean is not changed by it.
 * hi = +infinity: no upper bound. */
void oracle_transport_limit(int nx, int ny, int j0, int j1, int order, double* phi, double lo, double hi, int cap);

void oracle_dg_to_cg(int nx, int ny, int ncoef, const double* f_dg, double* f_cg);

void oracle_ice_strength(int nx, int ny, int j0, int j1, const oracle_mevp_params* p, const double* H,
    const double* A, double* pg);

/* stress update on element rows [k0,k1).  Adaptive form (p->aevp_c > 0): dt, cgh, cga are read and alpha_e[iy * nx + ix] receives the
 * element's alpha of this sub-iteration; uniform form: they are ignored (alpha_e may be NULL). */
void oracle_mevp_stress(int nx, int ny, int k0, int k1, double hx, double hy, const oracle_mevp_params* p,
    const double* u, const double* v, const double* pg, double* s11, double* s12, double* s22, double dt, const double* cgh,
    const double* cga, double* alpha_e);

/* velocity update on the nodes owned by element rows [j0,j1) (bottom-left ownership) */
void oracle_mevp_velocity(int nx, int ny, int j0, int j1, double hx, double hy, double dt,
    const oracle_mevp_params* p, const double* s11, const double* s12, const double* s22,
    const double* u_old, const double* v_old, double* u_new, double* v_new, const double* u0,
    const double* v0, const double* tax, const double* tay, const double* uo, const double* vo,
    const double* cgh, const double* cga, const double* alpha_e);

/* nsub full-domain sub-iterations, result left in u,v (scratch = 2 * nnodes doubles) */
void oracle_mevp_subcycle(int nx, int ny, double hx, double hy, double dt, int nsub,
    const oracle_mevp_params* p, double* s11, double* s12, double* s22, double* u, double* v,
    const double* u0, const double* v0, const double* tax, const double* tay, const double* uo,
    const double* vo, const double* cgh, const double* cga, const double* pg, double* scratch);

void oracle_wind_stress(long nnodes, const oracle_mevp_params* p, const double* ua, const double* va,
    double* tax, double* tay);

#ifdef __cplusplus
}
#endif
#endif
