/*
 * nsdg.h -- C ABI of the MI355X-native sea-ice dynamics / column-physics core (libnsdg.so).
 *
 * This is the drop-in boundary of SURVEY.md section 8(b): a host program that keeps the reference's
 * IModelStep / IStructure / ModuleLoader surfaces (see nextsimdg_amd/host/ and INTEGRATION.md)
 * calls these entry points once per model step; everything behind them is hand-written HIP for
 * gfx950.  No C++ types, exceptions or torch types cross the boundary.
 *
 * Conventions
 *  - Every function returns NSDG_OK (0) or a negative nsdg_status; nsdg_last_error() returns a
 *    thread-local description of the last failure.
 *  - All array arguments are DEVICE pointers to fp64 owned by the caller; the library never frees
 *    or reallocates them.  Calls are asynchronous on the context's stream: a buffer may be reused by
 *    the host after the stream has been synchronised (nsdg_ctx_synchronize or the caller's own sync).
 *  - One context per GPU/stream.  Calls on one context must not be issued concurrently from two
 *    host threads; different contexts are independent.  (The reference is single-threaded and
 *    non-re-entrant: static m_dt / m_freezer, core/src/PrognosticData.cpp:12-13.)
 *  - No clamping or input checking is added to the physics (SURVEY.md section 8b "Errors").  The results follow the
 *    reference's arithmetic, including its Inf / NaN / 0 for mld == 0, dt == 0, vanishing fluxes, zero pressure and
 *    non-finite forcing values: every division either is an IEEE division or ends in the IEEE sequence's special-case
 *    fix-up (csrc/column_step.hip).  The one exclusion: a SUBNORMAL or > 2^1022 value of slp, of an absolute
 *    temperature or of conc + del_c makes the reciprocal-based divisions return NaN where IEEE division scales.
 *
 * Data layout
 *  - element (ix, iy) of an nx x ny local array, ix fastest:  e = iy*nx + ix.  (The reference's
 *    restart layout is the same x-major linear index i*nx + j, core/src/DevGridIO.cpp:107-109.)
 *  - DG field, nc coefficients: coefficient-major planes  f[c*nx*ny + e]   (nc = 1/3/6 for DG0/1/2,
 *    8 for the stress space).
 *  - CG2 nodal field: (2nx+1) x (2ny+1) lattice, node (gx, gy) at  n = gy*(2nx+1) + gx.
 *  - edge-normal velocities at the ng = order+1 edge Gauss points:
 *        un_x[g*(nx+1)*ny + iy*(nx+1) + ex]   (vertical edges, normal = +x)
 *        un_y[g*nx*(ny+1) + ey*nx + ix]       (horizontal edges, normal = +y)
 *  - arrays private to the mEVP sub-cycle -- the stress coefficients s11/s12/s22 (nc = 8) and the ice
 *    strength at the 3x3 Gauss points pg (nc = 9, q = 3*qy + qx) -- use a TILED layout: tiles of 64
 *    consecutive elements of a row with all coefficients of the tile together, the coefficients in pairs
 *    interleaved by element (so that two coefficients travel in one 16-byte access): with the tile base
 *    T = ((iy*ntx + ix/64)*nc)*64, ntx = ceil(nx/64), and l = ix%64,
 *        coefficient c < 2*(nc/2):  a[T + (c/2)*128 + 2*l + c%2]
 *        odd last coefficient (nc = 9, c = 8):  a[T + (nc/2)*128 + l]
 *    nsdg_tiled_len(nx, ny, nc) doubles, 16-byte aligned.  Element rows stay contiguous, so row ranges / ghost
 *    rows work as for the plane layout.  (nextsimdg_amd/abi.py: tile() / untile() convert from / to planes.)
 *  - Row ranges [j0, j1) are ELEMENT rows of the local array; the edges of the local array are the
 *    physical boundary (zero inflow for transport, v = 0 for momentum).  A rank of a row-block
 *    decomposition passes arrays that include its ghost rows and the range of rows it owns.
 */
#ifndef NSDG_H
#define NSDG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSDG_ABI_VERSION 6

typedef enum {
    NSDG_OK = 0,
    NSDG_ERR_ARG = -1, /* bad argument (null pointer, bad size / order / range) */
    NSDG_ERR_HIP = -2, /* a HIP runtime call or kernel launch failed */
    NSDG_ERR_STATE = -3, /* call sequence error (e.g. grid not set) */
    NSDG_ERR_NODEVICE = -4, /* no usable HIP device */
    NSDG_ERR_COMM = -5 /* RCCL missing or an RCCL call failed; a rank of a local group failed */
} nsdg_status;

typedef struct nsdg_ctx nsdg_ctx;

/* ---- context ------------------------------------------------------------------------------
 * Replaces the process-wide static state of the reference (ModuleLoader singleton,
 * core/src/include/ModuleLoader.hpp:22-27; static physics parameters,
 * physics/src/modules/NextsimPhysics.cpp:26-39) by an explicit per-GPU object.
 * `stream` is a hipStream_t (NULL = the device's default stream). */
int nsdg_abi_version(void);
const char* nsdg_last_error(void);
int nsdg_ctx_create(int device_id, void* stream, nsdg_ctx** out);
int nsdg_ctx_destroy(nsdg_ctx* ctx);
/* Waits for everything enqueued on the context's stream (and its communication stream).  On a context with a
 * communicator the wait is BOUNDED by the communicator's deadline (nsdg_comm_deadline_set): if the streams have not
 * drained by then -- a neighbour rank that died leaves ncclRecv waiting for ever -- it returns NSDG_ERR_COMM, the
 * communicator is marked broken (nsdg_comm_finalize then aborts it instead of draining it) and the caller should
 * leave the process with a non-zero status without synchronising the device again. */
int nsdg_ctx_synchronize(nsdg_ctx* ctx);

/* Measurement aid: streaming device copy of n doubles, 16 bytes per lane and access -- the "device-copy peak" that
 * SURVEY.md section 8(d) asks the roofline fraction to be quoted against, measured on the box the bench runs on
 * (bench.py: roofline.copy_peak_GBs).  dst and src 16-byte aligned, not overlapping. */
int nsdg_copy_f64(nsdg_ctx* ctx, double* dst, const double* src, int64_t n);

/* ---- column physics (the reference's per-element step) ---------------------------------------- */
enum { NSDG_ALBEDO_SMU = 0, NSDG_ALBEDO_SMU2 = 1, NSDG_ALBEDO_CCSM = 2 }; /* physics/src/modules/modules.json:4-8 */
enum { NSDG_FREEZING_LINEAR = 0, NSDG_FREEZING_UNESCO = 1 }; /* core/src/modules/modules.json:4-7 */

/* Every configuration value of the column path:
 * nextsim_thermo.* (physics/src/modules/NextsimPhysics.cpp:50-58,76-82), thermoice0.*
 * (ThermoIce0.cpp:23-31), Hibler.* (HiblerConcentration.cpp:21-29), CCSMIceAlbedo.*
 * (CCSMIceAlbedo.cpp:40-41) and the [Modules] choices of IIceAlbedo / IFreezingPoint. */
typedef struct {
    double drag_ocean_q, drag_ocean_t, drag_ice_t, ocean_albedo, i0, min_conc, min_thick;
    double ks;
    double h0, phi_m;
    double ccsm_ice_albedo, ccsm_snow_albedo;
    int32_t flooding;
    int32_t albedo_kind;
    int32_t freezing_kind;
    int32_t reserved;
} nsdg_column_params;

void nsdg_column_default_params(nsdg_column_params* p);
int nsdg_column_params_set(nsdg_ctx* ctx, const nsdg_column_params* p);

/* optional diagnostics: NSDG_NDIAG planes of n doubles, diag[k*n + e] */
enum {
    NSDG_D_RHO = 0, NSDG_D_QA, NSDG_D_QW, NSDG_D_QI, NSDG_D_CSPEC, NSDG_D_TAU, NSDG_D_HI, NSDG_D_HS,
    NSDG_D_CNEW, NSDG_D_QIA, NSDG_D_QIO, NSDG_D_SUBL, NSDG_D_DQDT, NSDG_D_HIFROMS, NSDG_D_QOW, NSDG_NDIAG
};

/* One model step of the column physics on n independent elements; replaces the element loop of
 * DevStep::iterate (core/src/DevStep.cpp:14-23), i.e. per element
 *   IPhysics1d::updateDerivedData (physics/src/modules/include/IPhysics1d.hpp:33-45),
 *   NextsimPhysics::calculate (physics/src/modules/NextsimPhysics.cpp:116-131),
 *   PrognosticData::updateAndIntegrate (core/src/PrognosticData.cpp:63-71).
 * hice/cice/hsnow/tice0 are updated in place; sst/sss and the forcing are read-only (the reference
 * never updates sst/sss); newice is the per-element persistent NextsimPhysics::m_newice
 * (NextsimPhysics.cpp:244-253) and must be carried by the caller between steps; wind is
 * PhysicsData::windSpeed (physics/src/include/PhysicsData.hpp:44).  diag may be NULL. */
int nsdg_column_step(nsdg_ctx* ctx, int64_t n, double dt, double* hice, double* cice, double* hsnow,
    double* tice0, const double* sst, const double* sss, const double* tair, const double* tdew,
    const double* slp, const double* qsw, const double* qlw, const double* mld, const double* snowfall,
    const double* wind, double* newice, double* diag);

/* ---- dynamics: DG transport + mEVP (no counterpart in the reference snapshot: CMakeLists.txt:43-46
 *      comments the dynamics component out; built from the published formulation)
 * INPUT DOMAIN AND CLOSURE (round 5; the failure it answers: profiles/r04_soak_divergence_cause.md, the runs that now complete:
 * profiles/r05_closure.md).  Without a closure the scheme leaves the physical range once ice converging against a closed wall
 * reaches A = 1 (a one-element band piles up, the unlimited DG2 transport undershoots beside it, a node of floor mass between full
 * elements runs away).  Three pieces close it; none of them has a counterpart in the reference, whose concentration model has no
 * cap either (physics/src/modules/HiblerConcentration.cpp:32-47) -- the shape followed is the column model's own cut-off rule
 * (physics/src/modules/NextsimPhysics.cpp:210-219):
 *   1. ridging: the cell mean of a field with cap_mean != 0 (the concentration, hi = 1) is capped at hi at the end of a transport
 *      step; the mean thickness -- the conserved volume -- is a field of its own and is not touched, so convergence beyond a
 *      closed cover turns into (true) thickness;
 *   2. a Zhang-Shu scaling limiter at the end of a transport step keeps the values of a bounded field at the scheme's quadrature
 *      points (volume and edge Gauss points) and at its corners -- together: at every CG2 node of the element too -- inside [lo, hi]
 *      by scaling its higher coefficients; cell means are never changed;
 *   3. ice-free nodes (nsdg_mevp_params.min_conc / min_thick) are in free drift and do not feel their neighbours' stress.
 * 1 and 2 are per-field properties the host states with nsdg_transport_bounds_set (they are OFF until it does: the library does
 * not know which field is a concentration); 3 is ON by default.
 * What the closure does NOT replace is a sub-cycle that can follow the forcing (profiles/r05_closure.md): a UNIFORM alpha = beta must satisfy
 * alpha beta >= (2.4 pi)^2 P* dt / (2 delta_min rho_ice h^2) (linear stability, with the margin a one-day run needs), and with the
 * literature's delta_min = 2e-9 on a mesh finer than ~2 km that is an alpha of 10^4 ... 10^5, under which 120 sub-iterations
 * move stress and velocity less than 1 % of the way per model step -- a compressible cover (A < 1) then leaves the physical range
 * within a model day or two with or without the closure.  Round 5's hosts kept alpha = beta = 1500 and raised delta_min to the smallest
 * value for which that is stable on the mesh (1.9e-7 / 7.4e-7 / 3.0e-6 1/s at 500 / 250 / 125 m: a viscosity capped 90 to 1500 times
 * lower than the literature's).  Since round 6 the hosts run LOCAL, SOLUTION-ADAPTIVE alpha and beta (nsdg_mevp_params.aevp_c) at
 * delta_min = 2e-9: a uniform cover A = 0.9 then completes 53 model hours at 1024 x 1024 and the model day at 4096 x 4096
 * (profiles/r06_soak_1024_A09_adaptive_1600_steps.txt, r06_config5_one_day_A09_adaptive_host.txt).  nsdg_mevp_stable_params holds the
 * rule and its three forms; the defaults of nsdg_mevp_default_params -- uniform alpha = beta = 1500, delta_min = 2e-9 -- are the
 * literature's and are UNSTABLE below ~2 km: call nsdg_mevp_stable_params.  The calls do not check any of this; both hosts stop
 * loudly on non-finite fields. */
typedef struct {
    double rho_ice, rho_atm, rho_ocean;
    double c_atm, c_ocean;
    double pstar, compaction;
    double delta_min;
    double fc;
    double alpha, beta;
    double h_min; /* floor of the nodal mean thickness in the nodal mass */
    /* ice-free-node rule: a node with mean concentration < min_conc, true thickness cgH / cgA < min_thick or a mean thickness at the
     * mass floor (cgH <= h_min: its mass would be made up) is in free drift (full
     * exposure to wind and ocean drag, Coriolis, floor mass) and the stress divergence of the neighbouring elements is weighted by
     * 2^-100 there.  Defaults: the column model's cut-off values 1e-12 and 0.01 m (nextsim_thermo.min_conc / min_thick,
     * physics/src/modules/NextsimPhysics.cpp:81-82).  Both 0: rule off. */
    double min_conc, min_thick;
    /* LOCAL, SOLUTION-ADAPTIVE alpha and beta (round 6; after Kimmritz, Danilov & Losch 2016, "The adaptive EVP method for solving
     * the sea ice momentum equation", Ocean Modelling 101 -- parity unpinned like the rest of the dynamics).  aevp_c > 0 replaces the
     * uniform alpha, beta above: in every sub-iteration every element takes the alpha its own viscosity asks for,
     *     zeta_e = max over its 9 Gauss points of P / (2 Delta),
     *     alpha_e = sqrt(max(aevp_alpha_min^2, aevp_c zeta_e dt / (rho_ice h'_c hx hy))),
     * h'_c = max(nodal mean thickness at the element's centre node, h_min) (alpha_e = aevp_alpha_min where that node is ice-free), the
     * stress relaxes with 1 / alpha_e, and every node takes beta_n = max(aevp_alpha_min, max over its adjacent elements of alpha_e h'_c(e) / h'_n)
     * (alpha_e weighted by the ratio of the element's mass to the node's: the bound of every element-node pair; alpha_min at an ice-free node).  aevp_c is the
     * stability bound's constant: (2.4 pi)^2 = 56.85 is the bound DESIGN.md section 3.4 states with its margin; aevp_alpha_min must not be
     * chosen small on a fine mesh (nsdg_mevp_stable_params sets both).  Where the ice
     * deforms alpha is small and the stress follows the strain rate within a few sub-iterations, where it is rigid alpha is what the
     * uniform form needs everywhere; the converged sub-cycle solves the same implicit step.  aevp_c = 0 (the default of
     * nsdg_mevp_default_params): uniform alpha, beta -- bit-identical to ABI 5.  The adaptive form exists in the marching kernels
     * (nsdg_mevp_iterate*, nsdg_mevp_subcycle, nsdg_rb_mevp_run); nsdg_mevp_stress / nsdg_mevp_velocity return NSDG_ERR_STATE. */
    double aevp_c, aevp_alpha_min;
} nsdg_mevp_params;

void nsdg_mevp_default_params(nsdg_mevp_params* p);
int nsdg_mevp_params_set(nsdg_ctx* ctx, const nsdg_mevp_params* p);

/* The stability rule of the explicit sub-cycle, stated ONCE (round 5 had a copy in each host): on cells of size h = min(hx, hy) and
 * for a model time step dt the sub-cycle is linearly stable where  alpha beta >= (2.4 pi)^2 zeta dt / (m h^2)  (2.4: the margin a
 * one-day run needs, profiles/r02_alpha_margin.txt), zeta / m <= pstar / (2 delta_min rho_ice).  Three ways to satisfy it, chosen by
 * `mode`; the other members of *p are read (pstar, rho_ice, and what the mode keeps) and left alone:
 *   NSDG_SUBCYCLE_ADAPTIVE       aevp_c = (2.4 pi)^2: alpha_e, beta_n follow the local viscosity of every sub-iteration; delta_min stays (the
 *                                literature's 2e-9 by default); aevp_alpha_min = 50, the constant Kimmritz et al. (2016) publish.  The
 *                                hosts' default since round 6.  On meshes finer than 1 km every deforming element then sits AT its
 *                                stability limit and 120 sub-iterations do not converge: the velocity is noisy at element scale
 *                                (profiles/r06_adaptive_noise.md) -- the runs complete, the noise acts as a viscosity.
 *   NSDG_SUBCYCLE_ADAPTIVE_CONVERGED  the same with aevp_alpha_min = the bound's alpha for the reference strain rate NSDG_AEVP_DELTA_REF
 *                                (500 / 1000 / 2000 at 500 / 250 / 125 m, dt = 120 s; never below 50): the sub-cycle converges (0.1 % /
 *                                2 % of the maximum speed left after 120 sub-iterations at 1024^2 / 2048^2).  A converged plastic
 *                                solution exposes the time-step limit of the explicit strength / transport splitting (Lipscomb et al.
 *                                2007): dt = 120 s is too long for it at 250 m and below (2048^2 leaves the physical range after 11 model
 *                                hours, with dt = 60 s or 30 s it does not: profiles/r06_adaptive_noise.md) -- choose dt with the mesh.
 *   NSDG_SUBCYCLE_KEEP_ALPHA     uniform alpha = beta = p->alpha; delta_min is raised to the smallest value for which that is stable
 *                                (never lowered): the viscosity is capped -- below a strain rate of delta_min the ice creeps
 *                                (nsdg_mevp_creep_percent_per_day).  The hosts' default of round 5.
 *   NSDG_SUBCYCLE_KEEP_DELTA_MIN uniform alpha = beta = the bound's value for p->delta_min (at least 1500): rounds 1-4. */
enum { NSDG_SUBCYCLE_ADAPTIVE = 0, NSDG_SUBCYCLE_KEEP_ALPHA = 1, NSDG_SUBCYCLE_KEEP_DELTA_MIN = 2, NSDG_SUBCYCLE_ADAPTIVE_CONVERGED = 3 };
#define NSDG_AEVP_DELTA_REF 1.67e-6 /* 1/s: 14 % per day */
int nsdg_mevp_stable_params(nsdg_mevp_params* p, int32_t mode, double h, double dt);
/* strain rate below which the ice creeps instead of staying rigid (= delta_min), in percent per day */
double nsdg_mevp_creep_percent_per_day(const nsdg_mevp_params* p);

/* number of doubles of a tiled array (see "Data layout") */
int64_t nsdg_tiled_len(int32_t nx, int32_t ny, int32_t nc);

/* shape and cell size of the local element array all following calls refer to */
int nsdg_grid_set(nsdg_ctx* ctx, int32_t nx, int32_t ny, double hx, double hy);

/* kernel variant of the mEVP sub-cycle = the largest number of sub-iterations a kernel pass may perform:
 * 0 = two kernels per sub-iteration (element-wise stress, node-gather velocity); 1 = fused marching kernel, one launch per
 * sub-iteration (the bitwise reference of the multi-iteration passes); 2, 3, 4 = up to that many sub-iterations per pass on the
 * stage-per-wave pipeline (csrc/mevp_fused4.hip: one pipeline stage per wave of a four-wave workgroup, hand-over point to point
 * through LDS; nsdg_mevp_iterate2 / 3 / 4, nsdg_mevp_subcycle; what is left of a sub-cycle whose length is no multiple of the
 * variant runs as a shorter pass of the same kernel, a single sub-iteration on the kernel of variant 1).
 * NSDG_MEVP_DEFAULT_VARIANT (4) is the default of a new context.  Variants 1, 2, 3 and 4 agree bit for bit, variant 0 to fp64
 * round-off.  (Round 6 built passes of EIGHT -- two sub-iterations per stage wave -- bit-exact and slower: profiles/r06_fused8.md.)  (Rounds 1-4 had kernels of their own for 2 and 3 sub-iterations per pass -- two / three stages in ONE wave --
 * and a four-stage kernel with one workgroup barrier per march step: superseded, see the history of csrc/.) */
#define NSDG_MEVP_DEFAULT_VARIANT 4
int nsdg_mevp_variant_set(nsdg_ctx* ctx, int32_t variant);

/* CG2 velocity -> DG(order) velocity and edge-normal velocities used by the transport */
int nsdg_prepare_advection(nsdg_ctx* ctx, int32_t order, const double* u, const double* v, double* vx_dg,
    double* vy_dg, double* un_x, double* un_y);

/* transport stage kernel: 0 = one lane per element gathering all neighbours from memory, 2 (default) = two elements per
 * lane (16-byte accesses, the inner neighbour from registers; falls back to 0 for an odd nx or arrays that are not
 * 16-byte aligned).  Bit-identical results.  (1 was the marching kernel of rounds 1-2: never faster, removed.)
 * strip_rows: rows per workgroup (<= 4), 0 = default. */
int nsdg_transport_variant_set(nsdg_ctx* ctx, int32_t variant, int32_t strip_rows);

/* one Runge-Kutta stage on element rows [j0, j1) for nfields fields advected by the same velocity:
 *   out[f] = a*phi0[f] + b*(phis[f] + dt*L(phis[f]))
 * phi0/phis/out are HOST arrays of nfields DEVICE pointers. */
int nsdg_transport_stage(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, double dt, double a, double b,
    int32_t nfields, const double* const* phi0, const double* const* phis, double* const* out,
    const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y);

/* full SSP-RK(order+1) step over all rows (single-domain convenience); scratch: 2*nfields*nc*nx*ny */
int nsdg_transport_step(nsdg_ctx* ctx, int32_t order, double dt, int32_t nfields, double* const* phi,
    const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y, double* scratch);

/* The same step OUT OF PLACE in ONE launch: all Runge-Kutta stages of a step fused as a march -- a wave owns a window of
 * 64 - 2 (order + 1) element columns and a strip of rows and walks up the rows, stage k running k rows behind the newest row
 * read, the neighbours' edge traces taken from the adjacent lanes and from the rows the lane holds: every value is read once per
 * step, nothing goes through LDS, and the columns / rows at the edges of a window / strip are recomputed instead of exchanged.
 * For meshes whose step is bound by launch latency (BASELINE config 2: 512 x 512 DG1) and for callers that ping-pong their
 * fields.  phi_out must not alias phi_in.  Bit-identical to nsdg_transport_step. */
int nsdg_transport_step_oop(nsdg_ctx* ctx, int32_t order, double dt, int32_t nfields, const double* const* phi_in,
    double* const* phi_out, const double* vx_dg, const double* vy_dg, const double* un_x, const double* un_y);

/* The fused step on the rows [j0, j1) of the local array only (a row block's own rows): phi_in must be valid on order + 2 rows
 * below j0 and above j1 where the array has them (the ghost rows of a block whose ghost zone is at least that deep; the array
 * boundary needs none: nothing flows in).  Rows outside [j0, j1) of phi_out are not written. */
int nsdg_transport_step_oop_rows(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, double dt, int32_t nfields,
    const double* const* phi_in, double* const* phi_out, const double* vx_dg, const double* vy_dg, const double* un_x,
    const double* un_y);

/* Closure of a transport step (see "INPUT DOMAIN AND CLOSURE" above): bounds of the advected fields, in the order in which the
 * step entry points receive them.  nfields = 0 (default): no closure.  Once set, nsdg_transport_step, nsdg_transport_step_oop[_rows]
 * and nsdg_rb_transport_run apply cap + limiter to the new state before they return it (the marching launch in its epilogue, at
 * no extra memory traffic); they then require the same number of fields.  nsdg_transport_stage never limits.
 * The bounds are STATE OF THE CONTEXT: they apply to every later step call on it, whatever fields that call advances, until they are
 * set again (nfields = 0 clears them).  A row-block plan can carry its own instead (nsdg_rb_transport_desc.own_bounds), which then
 * neither reads nor changes the context's. */
typedef struct {
    double lo, hi; /* bounds at the quadrature points; hi = +infinity (HUGE_VAL): no upper bound */
    int32_t cap_mean; /* != 0: a cell mean above hi is set to hi (needs a finite hi) */
    int32_t reserved;
} nsdg_field_bounds;
int nsdg_transport_bounds_set(nsdg_ctx* ctx, int32_t nfields, const nsdg_field_bounds* bounds);

/* cap + limiter as a pass of its own, in place on the element rows [j0, j1) -- for callers that compose a step from
 * nsdg_transport_stage calls; bit-identical to what the step entry points apply.  NSDG_ERR_STATE without bounds. */
int nsdg_transport_limit(nsdg_ctx* ctx, int32_t order, int32_t j0, int32_t j1, int32_t nfields, double* const* phi);

/* nodal average of a DG field on the CG2 lattice (mean thickness / concentration at the nodes) */
int nsdg_dg_to_cg(nsdg_ctx* ctx, int32_t ncoef, const double* f_dg, double* f_cg);

/* P = pstar * max(H,0) * exp(-C (1 - clamp(A,0,1))) at the 3x3 Gauss points of rows [j0, j1) */
int nsdg_ice_strength(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* H, const double* A, double* pg);

/* ---- external forcing providers, evaluated on the device at the model time of each step (csrc/forcing.hip) ----
 * They replace DummyExternalData::setAll (core/src/include/DummyExternalData.hpp:22-34: the same eight constants in
 * every element, set once) and fill PhysicsData::windSpeed (physics/src/include/PhysicsData.hpp:26,44), which a run
 * of the reference never sets.
 *
 * Placement of the local array in the global domain for these providers: local element row 0 is global row `row0`
 * of `ny_global` rows (defaults 0 and the local ny: a single domain).  A row block then evaluates the very
 * expressions the single domain evaluates, bit for bit. */
int nsdg_block_set(nsdg_ctx* ctx, int32_t row0, int32_t ny_global);

/* Square box test of side domain_size [m] at model time t [s] on the CG2 lattice of the local array: cyclone wind
 * (ua, va), its centre drifting along the diagonal, and circular ocean current (uo, vo); either pair may be NULL. */
int nsdg_boxtest_forcing(nsdg_ctx* ctx, double domain_size, double t, double* ua, double* va, double* uo, double* vo);

/* Thermodynamic forcing planes of the column step (nx*ny doubles each) at model time t [s]:
 * NSDG_FORCING_DUMMY = the reference's constants (tair -1, tdew -4, slp 1e5, qsw 0, qlw 311, mld 10, snowfall 0);
 * NSDG_FORCING_WINTER = smooth winter fields in the ranges of SURVEY.md section 8(d): a synoptic pattern drifting eastwards
 * with a 5-day period and a diurnal cycle of the short-wave flux. */
enum { NSDG_FORCING_DUMMY = 0, NSDG_FORCING_WINTER = 1 };
int nsdg_column_forcing(nsdg_ctx* ctx, int32_t kind, double t, double* tair, double* tdew, double* slp, double* qsw, double* qlw,
    double* mld, double* snowfall);

/* wind speed of the column step = |u_a| at the element centre (CG2 node (2 ix + 1, 2 iy + 1)) */
int nsdg_column_wind(nsdg_ctx* ctx, const double* ua, const double* va, double* wind);

/* tau_a = c_atm * rho_atm * |u_a| u_a at nnodes nodes */
int nsdg_wind_stress(nsdg_ctx* ctx, int64_t nnodes, const double* ua, const double* va, double* tax, double* tay);

/* mEVP stress update (in place) on element rows [k0, k1):  S <- (1-1/alpha) S + (1/alpha) Proj sigma(v) */
int nsdg_mevp_stress(nsdg_ctx* ctx, int32_t k0, int32_t k1, const double* u, const double* v, const double* pg,
    double* s11, double* s12, double* s22);

/* Per-step coefficients of the momentum update at every CG2 node in an internal packed layout
 * (6 doubles per node, csrc/mevp_common.h), valid for the whole sub-cycle of one time step because
 * u0, v0, the wind stress, the ocean current and the nodal H, A do not change inside it.  The context
 * remembers dt and the mEVP parameters of the packing for the following iterate/velocity calls.
 * `packed` must hold 8*(2nx+1)*(2ny+1) doubles (6 used) and be 16-byte aligned. */
int nsdg_mevp_pack_nodal(nsdg_ctx* ctx, double dt, const double* u0, const double* v0, const double* tax,
    const double* tay, const double* uo, const double* vo, const double* cgh, const double* cga, double* packed);

/* The whole per-step nodal preparation in one launch: nodal means of the DG2 fields H and A, wind stress of
 * (ua, va) and coefficient packing -- equivalent to nsdg_dg_to_cg x2 + nsdg_wind_stress + nsdg_mevp_pack_nodal
 * (bit-identical `packed`) without materialising cgH, cgA and tau_a.  u0, v0: velocity at the start of the step. */
int nsdg_mevp_prepare(nsdg_ctx* ctx, double dt, const double* H, const double* A, const double* ua, const double* va,
    const double* uo, const double* vo, const double* u0, const double* v0, double* packed);

/* mEVP velocity update of the nodes owned (bottom-left) by element rows [j0, j1) */
int nsdg_mevp_velocity(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11, const double* s12,
    const double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed);

/* one complete sub-iteration: stress on element rows [k0, j1) (S_out <- relax(S_in, u_old)), then the
 * velocity of the nodes owned by rows [j0, j1) from S_out.  Out-of-place in both the stress and the
 * velocity, so a row may be updated redundantly by two owners (a rank and its neighbour, or two row
 * strips of the fused kernel) with bit-identical results.  k0 == j0 - 1 (one redundant ghost row below)
 * or k0 == j0 == 0 (rows start at the physical boundary).  Used by multi-rank drivers that exchange
 * velocity halos between sub-iterations. */
int nsdg_mevp_iterate(nsdg_ctx* ctx, int32_t k0, int32_t j0, int32_t j1, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old,
    const double* v_old, double* u_new, double* v_new, const double* packed, const double* pg);

/* TWO complete sub-iterations in one pass on the owned element rows [j0, j1): reads S_in, u_old on rows
 * j0-2 .. j1 (node rows 2(j0-2) .. 2*j1+2) and writes S_out = S^{p+2} on rows [j0, j1) and u_new = u^{p+2} on
 * the nodes they own; the intermediate stress and velocity never leave the registers.  j0 == 0 (physical
 * boundary) or j0 >= 2 (two ghost rows below); above, one ghost row or the physical boundary (j1 == ny).
 * A multi-rank driver refreshes the ghost rows of S_out and u_new after every pass (or runs k passes on row ranges
 * shrinking by 2 rows per side and pass on a (2k, 2k-1)-row ghost zone).  Requires variant >= 2. */
int nsdg_mevp_iterate2(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old, const double* v_old,
    double* u_new, double* v_new, const double* packed, const double* pg);

/* THREE complete sub-iterations in one pass on the owned element rows [j0, j1): reads S_in, u_old on rows
 * j0-3 .. j1+1 and writes S_out = S^{p+3} on rows [j0, j1) and u_new = u^{p+3} on the nodes they own.  j0 == 0
 * or j0 >= 3 (three ghost rows below); j1 == ny or j1 + 2 <= ny (two ghost rows above).  Requires variant >= 3. */
int nsdg_mevp_iterate3(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old, const double* v_old,
    double* u_new, double* v_new, const double* packed, const double* pg);

/* The same on TWO disjoint row ranges [j0a, j1a) and [j0b, j1b) in ONE launch -- the two bands of rows whose results a row
 * block sends to its neighbours: as one launch they share the resident wave slots instead of paying two pipeline fills
 * one after the other.  Each range obeys the ghost-row conditions of nsdg_mevp_iterate3; bit-identical to two calls. */
int nsdg_mevp_iterate3_pair(nsdg_ctx* ctx, int32_t j0a, int32_t j1a, int32_t j0b, int32_t j1b, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old, const double* v_old, double* u_new,
    double* v_new, const double* packed, const double* pg);

/* FOUR complete sub-iterations in one pass on the owned element rows [j0, j1): reads S_in, u_old on rows
 * j0-4 .. j1+2 and writes S_out = S^{p+4} on rows [j0, j1) and u_new = u^{p+4} on the nodes they own.  j0 == 0
 * or j0 >= 4 (four ghost rows below); j1 == ny or j1 + 3 <= ny (three ghost rows above).  Requires variant 4. */
int nsdg_mevp_iterate4(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old, const double* v_old,
    double* u_new, double* v_new, const double* packed, const double* pg);

/* The same on TWO disjoint row ranges in ONE launch (see nsdg_mevp_iterate3_pair); each range obeys the ghost-row
 * conditions of nsdg_mevp_iterate4; bit-identical to two calls. */
int nsdg_mevp_iterate4_pair(nsdg_ctx* ctx, int32_t j0a, int32_t j1a, int32_t j0b, int32_t j1b, const double* s11_in, const double* s12_in,
    const double* s22_in, double* s11_out, double* s12_out, double* s22_out, const double* u_old, const double* v_old, double* u_new,
    double* v_new, const double* packed, const double* pg);

/* nsub sub-iterations over the whole local array (packs the nodal coefficients, then iterates);
 * result in s11/s12/s22 and u, v (u0/v0 may be the same arrays as u/v).  scratch: 10*(2nx+1)*(2ny+1) + 24*nx*ny doubles, 16-byte aligned
 * (packed coefficients + ping-pong copies of the velocity and the stress). */
int nsdg_mevp_subcycle(nsdg_ctx* ctx, double dt, int32_t nsub, double* s11, double* s12, double* s22, double* u,
    double* v, const double* u0, const double* v0, const double* tax, const double* tay, const double* uo,
    const double* vo, const double* cgh, const double* cga, const double* pg, double* scratch);

/* The waits of the stage-per-wave pipeline (csrc/mevp_fused4.hip, csrc/mevp_p2p.h) are bounded: a wave that waits for
 * a hand-over longer than ~0.2 s gives up, releases the other waits of its workgroup and reports the event -- the launch then finishes
 * with WRONG results instead of hanging the GPU.  The report reaches the host without being asked for: the kernel sets a flag in host
 * memory that belongs to the context, and from then on nsdg_ctx_synchronize (after the launch has completed), nsdg_mevp_subcycle and
 * nsdg_rb_mevp_run (at their next call) return NSDG_ERR_HIP -- sticky, until this call takes the events.  This call waits for the
 * context's stream, returns the number of events since the last call (0 in a correct program), resets the count and clears the
 * error status.  Counter and flag are per context. */
int nsdg_mevp_pipeline_health(nsdg_ctx* ctx, uint32_t* waits_given_up);

/* rows per strip of the fused marching kernel (performance knob; results do not depend on it);
 * 0 (default) = chosen per launch from the row count and the number of resident wave slots */
int nsdg_mevp_strip_rows_set(nsdg_ctx* ctx, int32_t rows);
/* register budget of the fused kernel: 1 or 2 resident waves per SIMD (performance knob) */
int nsdg_mevp_occupancy_set(nsdg_ctx* ctx, int32_t waves_per_simd);

/* ---- row-block decomposition: ghost-row exchange (SURVEY.md section 8(b) "nsdg_halo_exchange", 8(e)) ----------
 * The reference is a single-process, single-thread program; these entry points have no
 * counterpart in it.  They sit behind the same seam as everything else, the batched model step
 * IModelStep::iterate (core/src/include/IModelStep.hpp:16-34): a multi-rank step owns one context per GPU and
 * exchanges ghost rows with its <= 2 neighbours between kernel launches.  No collective is involved.
 *
 * Communicator: one per context.  nsdg_comm_init is the RCCL transport (one process per GPU; `id` is the
 * NSDG_COMM_ID_BYTES-byte ncclUniqueId obtained by rank 0 from nsdg_comm_unique_id and distributed by the caller --
 * torch.distributed in the Python driver, a TCP rendezvous in the C++ host); collective over the ranks.
 * librccl.so.1 is loaded on first use.  nsdg_comm_init_local is the in-process transport: the ranks of `group`
 * are contexts of ONE process driven by one host thread each (device-to-device copies, host hand-shake); used
 * by the one-GPU tests of the multi-rank driver and usable as a single-process multi-GPU mode. */
#define NSDG_COMM_ID_BYTES 128
int nsdg_comm_unique_id(void* id);
int nsdg_comm_init(nsdg_ctx* ctx, int32_t rank, int32_t world, const void* id);
int nsdg_comm_init_local(nsdg_ctx* ctx, int64_t group, int32_t rank, int32_t world);
int nsdg_comm_finalize(nsdg_ctx* ctx);
int nsdg_comm_rank(nsdg_ctx* ctx, int32_t* rank, int32_t* world);
/* Upper bound in seconds on any wait for a neighbour: the host-side hand-shake of the local transport and the drain in
 * nsdg_ctx_synchronize (RCCL: a dead peer never answers).  0 = wait for ever.  Default: the environment variable
 * NSDG_COMM_TIMEOUT_S, else 300.  May be called before or after nsdg_comm_init*. */
int nsdg_comm_deadline_set(nsdg_ctx* ctx, double seconds);

/* Rehearsal aid for a box with ONE GPU (tools/rank_share_timing.py, bench.py's loopback rehearsal): a loopback exchange
 * has no wire time, so a kernel that spins for delay_us + bytes of the larger direction / gbs (GB/s per direction) is put on
 * the communication stream between pack and transport of every exchange.  0, 0 = off (the default unless the environment
 * variables NSDG_HALO_DELAY_US / NSDG_HALO_SIM_GBS are set when the communicator is created: they are read once, there). */
int nsdg_comm_simulate_wire(nsdg_ctx* ctx, double delay_us, double gbs);

/* A plan fixes what ONE kind of exchange moves: for each of the four directions a list of contiguous blocks of
 * doubles in the caller's arrays (row blocks: ghost rows are contiguous in every layout of this ABI).  up_send
 * travels to rank_above and is received there as from_below, down_send travels to rank_below and arrives as
 * from_above; the packed sizes of matching directions must agree between neighbours.  rank_* = -1: physical
 * boundary, no traffic.  rank_below == rank_above == own rank: loopback (what goes up arrives from below), the
 * one-GPU rehearsal of an interior block.  Every rank of a communicator creates its plans in the same order. */
typedef struct nsdg_halo nsdg_halo;
typedef struct {
    double* ptr;
    int64_t count;
} nsdg_halo_seg;
int nsdg_halo_plan_create(nsdg_ctx* ctx, int32_t rank_below, int32_t rank_above, int32_t n_up, const nsdg_halo_seg* up_send,
    int32_t n_down, const nsdg_halo_seg* down_send, int32_t n_above, const nsdg_halo_seg* from_above, int32_t n_below,
    const nsdg_halo_seg* from_below, nsdg_halo** out);
int nsdg_halo_plan_destroy(nsdg_halo* plan);
/* packed doubles per direction (segments are padded to 16 bytes) */
int nsdg_halo_counts(const nsdg_halo* plan, int64_t* up, int64_t* down, int64_t* above, int64_t* below);

/* One exchange.  start: everything enqueued on the context's stream so far is visible to the exchange; the send
 * blocks are packed (one launch) and the transport is posted on the context's communication stream.  finish: the
 * received blocks are scattered into the ghost rows (one launch) and the context's stream waits for that.  Neither
 * call blocks the host (the local transport blocks until the neighbour thread has posted).  Kernels launched between
 * the two calls overlap with the exchange; they must not write the send blocks nor touch the receive blocks. */
int nsdg_halo_start(nsdg_ctx* ctx, nsdg_halo* plan);
int nsdg_halo_finish(nsdg_ctx* ctx, nsdg_halo* plan);

/* What the exchanges of a plan (or of all plans of a row-block driver) cost, measured with events on the communication
 * stream: `ms` is the time from "the data to send is ready" (the compute stream has reached nsdg_halo_start) to "the
 * ghost rows are written" (the unpack kernel has finished), summed over `exchanges` exchanges -- it contains the pack
 * and unpack kernels, the transfer and any wait for a neighbour that is late, and it overlaps with whatever the
 * compute stream does between start and finish.  `untimed` exchanges could not be timed (events recycled before they
 * completed).  The call waits for the last exchange of the plan(s) to finish; reset != 0 zeroes the counters. */
typedef struct {
    int64_t exchanges, untimed;
    double ms;
    int64_t bytes_sent, bytes_received; /* per exchange */
} nsdg_halo_stats;
int nsdg_halo_stats_get(nsdg_ctx* ctx, nsdg_halo* plan, nsdg_halo_stats* out, int32_t reset);

/* ---- row-block drivers: one call per model step for the sub-cycle and for the transport of a rank's block ------
 * The sequence of kernel passes and ghost exchanges that a multi-rank model step (IModelStep::iterate,
 * core/src/include/IModelStep.hpp:16-34) issues, built once from the block geometry and the device arrays:
 *   local array nx x ny (ghost rows included), owned element rows [j0, j1), nominal ghost depths
 *   (depth_below, depth_above) -- the same on every rank -- and the neighbour ranks (-1 = physical boundary; a
 *   rank keeps ghost rows only towards existing neighbours: j0 = depth_below or 0, j1 = ny - depth_above or ny).
 * Kernels with v = 4 / 3 / 2 sub-iterations per pass (nsdg_mevp_variant_set) are used when the depths are (v k, v k - 1)
 * -- k passes run between two exchanges, the ghost rows are advanced redundantly -- or when the block has no
 * neighbours; otherwise one sub-iteration per pass with (1, 1).  Both calls are asynchronous on the context's
 * stream.  Blocks with neighbours need a communicator (nsdg_comm_init*) before the plan is created. */
typedef struct nsdg_rb_mevp nsdg_rb_mevp;
typedef struct {
    int32_t nx, ny, j0, j1;
    int32_t depth_below, depth_above;
    int32_t rank_below, rank_above;
    int32_t nsub; /* sub-iterations per model step */
    int32_t overlap; /* launch the rows whose results travel first, post the exchange, then the interior */
    int32_t use_graph; /* replay the launches between two exchanges as one hipGraph */
    int32_t reserved;
    double *s11[2], *s12[2], *s22[2]; /* ping-pong: tiled stress */
    double *u[2], *v[2]; /* ping-pong: CG2 velocity */
    const double* packed; /* nsdg_mevp_prepare / nsdg_mevp_pack_nodal output of this step */
    const double* pg; /* nsdg_ice_strength output of this step */
} nsdg_rb_mevp_desc;
int nsdg_rb_mevp_create(nsdg_ctx* ctx, const nsdg_rb_mevp_desc* desc, nsdg_rb_mevp** out);
int nsdg_rb_mevp_destroy(nsdg_rb_mevp* plan);
/* sub-iterations per kernel pass and passes between two exchanges the plan settled on */
int nsdg_rb_mevp_info(const nsdg_rb_mevp* plan, int32_t* per_pass, int32_t* group_passes);
/* nsub sub-iterations; `parity` = which of the ping-pong buffers holds the current iterate (ghost rows valid),
 * *parity_out = which one holds the result (ghost rows refreshed) */
int nsdg_rb_mevp_run(nsdg_ctx* ctx, nsdg_rb_mevp* plan, int32_t parity, int32_t* parity_out);
/* exchange statistics summed over the plan's ghost exchanges (bytes: of its largest exchange) */
int nsdg_rb_mevp_stats(nsdg_ctx* ctx, nsdg_rb_mevp* plan, nsdg_halo_stats* out, int32_t reset);

#define NSDG_RB_MAX_FIELDS 4
typedef struct nsdg_rb_transport nsdg_rb_transport;
typedef struct {
    int32_t nx, ny, j0, j1;
    int32_t depth_below, depth_above;
    int32_t rank_below, rank_above;
    int32_t order; /* 2 */
    int32_t nfields;
    double* phi[NSDG_RB_MAX_FIELDS]; /* DG2 coefficient planes [6][ny][nx] of each advected field */
    double* t1[NSDG_RB_MAX_FIELDS]; /* same size: stage buffer; receives the new state */
    double* t2[NSDG_RB_MAX_FIELDS]; /* same size: stage buffer */
    const double *vx_dg, *vy_dg, *un_x, *un_y; /* nsdg_prepare_advection output of this step */
    /* closure of the step (nsdg_field_bounds below "INPUT DOMAIN AND CLOSURE").  own_bounds = 0: whatever nsdg_transport_bounds_set has
     * stated on the context when the plan RUNS (ABI 5); own_bounds != 0: the plan's own -- nbounds of them (0 = none, or nfields) --
     * whatever the context says: a context that also steps other fields (a snow layer, a damage field) does not leak its bounds into
     * this plan, nor this plan's into those calls. */
    int32_t own_bounds, nbounds;
    nsdg_field_bounds bounds[NSDG_RB_MAX_FIELDS];
} nsdg_rb_transport_desc;
int nsdg_rb_transport_create(nsdg_ctx* ctx, const nsdg_rb_transport_desc* desc, nsdg_rb_transport** out);
int nsdg_rb_transport_destroy(nsdg_rb_transport* plan);
/* one SSP-RK3 step.  parity 0: the state is in phi[], the new state (ghost rows refreshed) is written to t1[];
 * parity 1: the other way round; *parity_out = 1 - parity */
int nsdg_rb_transport_run(nsdg_ctx* ctx, nsdg_rb_transport* plan, double dt, int32_t parity, int32_t* parity_out);
int nsdg_rb_transport_stats(nsdg_ctx* ctx, nsdg_rb_transport* plan, nsdg_halo_stats* out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* NSDG_H */
