// nsdg_ctx.hip -- context, parameters and error reporting of libnsdg.so.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "nsdg_internal.h"

static thread_local char g_err[512] = "";

void nsdg_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int nsdg_p2p_check(nsdg_ctx* ctx, const char* where)
{
    if (*(volatile unsigned*)ctx->p2p_flag_host)
        ctx->p2p_given_up = 1;
    if (!ctx->p2p_given_up)
        return NSDG_OK;
    nsdg_set_error("%s: a bounded wait of the mEVP pipeline gave up in an earlier launch on this context: its fields are wrong "
                   "(nsdg_mevp_pipeline_health returns the count and clears this status)", where);
    return NSDG_ERR_HIP;
}

extern "C" {

int nsdg_abi_version(void) { return NSDG_ABI_VERSION; }

const char* nsdg_last_error(void) { return g_err; }

void nsdg_column_default_params(nsdg_column_params* p)
{
    // defaults of the reference: physics/src/modules/NextsimPhysics.cpp:76-82, ThermoIce0.cpp:30-31,
    // HiblerConcentration.cpp:28-29, CCSMIceAlbedo.cpp:22-23; default module = first listed
    // (core/src/ModuleLoader.cpp:51-54): SMUIceAlbedo, LinearFreezing.
    p->drag_ocean_q = 1.5e-3;
    p->drag_ocean_t = 0.83e-3;
    p->drag_ice_t = 1.3e-3;
    p->ocean_albedo = 0.07;
    p->i0 = 0.17;
    p->min_conc = 1e-12;
    p->min_thick = 0.01;
    p->ks = 0.3096;
    p->h0 = 0.25;
    p->phi_m = 0.5;
    p->ccsm_ice_albedo = 0.538;
    p->ccsm_snow_albedo = 0.8256;
    p->flooding = 1;
    p->albedo_kind = NSDG_ALBEDO_SMU;
    p->freezing_kind = NSDG_FREEZING_LINEAR;
    p->reserved = 0;
}

void nsdg_mevp_default_params(nsdg_mevp_params* p)
{
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
    // ice-free-node rule ON, with the column model's own cut-off values (nextsim_thermo.min_conc / min_thick,
    // physics/src/modules/NextsimPhysics.cpp:81-82, applied there as  c_new < minc || hi < minh,  :210-219)
    p->min_conc = 1e-12;
    p->min_thick = 0.01;
    // uniform alpha, beta (bit-identical to ABI 5); nsdg_mevp_stable_params(.., NSDG_SUBCYCLE_ADAPTIVE, ..) turns the adaptive form on
    p->aevp_c = 0.;
    p->aevp_alpha_min = 50.;
}

int nsdg_mevp_stable_params(nsdg_mevp_params* p, int32_t mode, double h, double dt)
{
    NSDG_CHECK_ARG(p != nullptr, "null parameters");
    NSDG_CHECK_ARG(h > 0 && dt > 0, "cell size and time step must be positive");
    NSDG_CHECK_ARG(p->pstar > 0 && p->rho_ice > 0, "pstar and rho_ice must be positive");
    const double pi = 3.14159265358979323846, margin = 2.4;
    // alpha beta >= c zeta dt / (m h^2) with c = (margin pi)^2 and zeta / m <= pstar / (2 delta_min rho_ice)
    const double c = margin * margin * pi * pi;
    switch (mode) {
    case NSDG_SUBCYCLE_ADAPTIVE: // the published constant
        p->aevp_c = c;
        p->aevp_alpha_min = 50.;
        return NSDG_OK;
    case NSDG_SUBCYCLE_ADAPTIVE_CONVERGED:
        // The lower bound of alpha_e = the uniform bound's alpha for a reference strain rate of active deformation (NSDG_AEVP_DELTA_REF,
        // 1.67e-6 1/s = 14 % per day): 500 / 1000 / 2000 at 500 / 250 / 125 m with dt = 120 s, never below 50.  Measured
        // (profiles/r06_adaptive_noise.md): with 50 every deforming element sits AT its stability limit and the iteration is noisy at
        // element scale on meshes finer than 1 km (a tenth of the nodes change by > 1 mm/s per model step, the largest by a third of the
        // maximum speed); from this floor on the same momentum problem converges to a steady state -- and the coupled model then needs
        // a time step that fits the mesh (include/nsdg.h).
        p->aevp_c = c;
        p->aevp_alpha_min = std::max(50., std::sqrt(c * p->pstar * dt / (2. * NSDG_AEVP_DELTA_REF * p->rho_ice * h * h)));
        return NSDG_OK;
    case NSDG_SUBCYCLE_KEEP_ALPHA:
        NSDG_CHECK_ARG(p->alpha > 0, "alpha must be positive");
        p->aevp_c = 0.;
        p->beta = p->alpha;
        p->delta_min = std::max(p->delta_min, c * p->pstar * dt / (2. * p->rho_ice * h * h * p->alpha * p->alpha));
        return NSDG_OK;
    case NSDG_SUBCYCLE_KEEP_DELTA_MIN:
        NSDG_CHECK_ARG(p->delta_min > 0, "delta_min must be positive");
        p->aevp_c = 0.;
        p->alpha = p->beta = std::max(1500., std::sqrt(c * p->pstar * dt / (2. * p->delta_min * p->rho_ice * h * h)));
        return NSDG_OK;
    }
    nsdg_set_error("nsdg_mevp_stable_params: unknown mode %d", (int)mode);
    return NSDG_ERR_ARG;
}

double nsdg_mevp_creep_percent_per_day(const nsdg_mevp_params* p) { return p ? p->delta_min * 86400. * 100. : 0.; }

int nsdg_ctx_create(int device_id, void* stream, nsdg_ctx** out)
{
    NSDG_CHECK_ARG(out != nullptr, "null output pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        nsdg_set_error("nsdg_ctx_create: no HIP device available");
        return NSDG_ERR_NODEVICE;
    }
    NSDG_CHECK_ARG(device_id >= 0 && device_id < ndev, "device id out of range");
    NSDG_CHECK_HIP(hipSetDevice(device_id));
    nsdg_ctx* c = new nsdg_ctx();
    c->device = device_id;
    c->stream = (hipStream_t)stream;
    nsdg_column_default_params(&c->column);
    nsdg_mevp_default_params(&c->mevp);
    c->nx = c->ny = 0;
    c->row0 = c->ny_global = 0;
    c->hx = c->hy = 0.;
    c->mevp_variant = NSDG_MEVP_DEFAULT_VARIANT;
    c->strip_rows = 0;
    {
        hipDeviceProp_t prop;
        c->num_cus = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    c->fused_min_waves = 1;
    // measured (2048^2 DG2): gather 0.802-0.806 ms per RK3 step of one field, two elements per lane 0.770-0.786 ms
    // (profiles/r03_transport_ab.log; bit-identical); march 1.51-1.76 against 1.55 ms for two fields in round 1
    c->transport_variant = 2;
    c->transport_rows = 0;
    c->nbounds = 0;
    c->pack_dt = 0.;
    c->d_ptrs = nullptr;
    c->comm = nullptr;
    c->comm_group = 0;
    {
        const char* v = std::getenv("NSDG_COMM_TIMEOUT_S");
        c->comm_deadline_s = (v && *v) ? std::atof(v) : 300.;
        if (!(c->comm_deadline_s >= 0.))
            c->comm_deadline_s = 300.;
    }
    // the report channel of the pipelines' bounded waits: a wait that gives up must reach the host without anybody asking
    c->p2p_count_dev = nullptr, c->p2p_flag_host = nullptr, c->p2p_flag_dev = nullptr, c->p2p_given_up = 0;
    hipError_t e = hipMalloc((void**)&c->p2p_count_dev, sizeof(unsigned));
    if (e == hipSuccess)
        e = hipMemset(c->p2p_count_dev, 0, sizeof(unsigned));
    if (e == hipSuccess)
        e = hipHostMalloc((void**)&c->p2p_flag_host, sizeof(unsigned), hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->p2p_flag_host = 0;
        e = hipHostGetDevicePointer((void**)&c->p2p_flag_dev, c->p2p_flag_host, 0);
    }
    if (e != hipSuccess) {
        nsdg_set_error("nsdg_ctx_create: allocating the pipeline report channel failed: %s", hipGetErrorString(e));
        nsdg_ctx_destroy(c);
        return NSDG_ERR_HIP;
    }
    *out = c;
    return NSDG_OK;
}

int nsdg_ctx_destroy(nsdg_ctx* ctx)
{
    if (!ctx)
        return NSDG_OK;
    nsdg_comm_finalize(ctx);
    if (ctx->p2p_count_dev || ctx->p2p_flag_host) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream); // no launch may still hold the addresses
        if (ctx->p2p_count_dev)
            (void)hipFree(ctx->p2p_count_dev);
        if (ctx->p2p_flag_host)
            (void)hipHostFree(ctx->p2p_flag_host);
    }
    delete ctx;
    return NSDG_OK;
}

int nsdg_ctx_synchronize(nsdg_ctx* ctx)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    if (ctx->comm) { // a dead neighbour must not block this rank for ever
        const int rc = nsdg_comm_bounded_drain(ctx);
        return rc ? rc : nsdg_p2p_check(ctx, __func__);
    }
    NSDG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return nsdg_p2p_check(ctx, __func__);
}

// waits of the pipelines that gave up since the last call (0 in a correct program); takes the events: the context's status is OK again
int nsdg_mevp_pipeline_health(nsdg_ctx* ctx, uint32_t* waits_given_up)
{
    NSDG_CHECK_ARG(ctx && waits_given_up, "null argument");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    NSDG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); // no launch of this context is running: read and reset cannot lose an event
    unsigned n = 0;
    NSDG_CHECK_HIP(hipMemcpy(&n, ctx->p2p_count_dev, sizeof(unsigned), hipMemcpyDeviceToHost));
    NSDG_CHECK_HIP(hipMemset(ctx->p2p_count_dev, 0, sizeof(unsigned)));
    *(volatile unsigned*)ctx->p2p_flag_host = 0;
    ctx->p2p_given_up = 0;
    *waits_given_up = n;
    return NSDG_OK;
}

namespace {
// one 16-byte access per lane, one workgroup per 4 KB, no loop: the fastest of the copies compared in
// tools/microbench/copy_peak.hip on the MI355X (profiles/r03_copy_peak.txt: 6.23 TB/s read + write; two / four / eight accesses
// in flight per lane 6.0 / 5.8 / 3.7, grid-stride loops 5.4-5.6, hipMemcpyAsync 5.0-5.5, non-temporal x 4 6.1-6.3)
__global__ __launch_bounds__(256) void copy16_kernel(double2* __restrict__ dst, const double2* __restrict__ src, long n2)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n2)
        dst[i] = src[i];
}
} // namespace

int nsdg_copy_f64(nsdg_ctx* ctx, double* dst, const double* src, int64_t n)
{
    NSDG_CHECK_ARG(ctx && dst && src && n >= 0, "null pointer or negative count");
    NSDG_CHECK_ARG(((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0, "dst and src must be 16-byte aligned");
    NSDG_CHECK_ARG(dst + n <= src || src + n <= dst, "dst and src must not overlap");
    if (n == 0)
        return NSDG_OK;
    const long n2 = n >> 1;
    NSDG_CHECK_ARG(n2 < (1L << 39), "count too large");
    const unsigned blocks = (unsigned)((n2 + 255) / 256);
    if (n2)
        hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(dst), reinterpret_cast<const double2*>(src), n2);
    if (n & 1)
        NSDG_CHECK_HIP(hipMemcpyAsync(dst + n - 1, src + n - 1, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_column_params_set(nsdg_ctx* ctx, const nsdg_column_params* p)
{
    NSDG_CHECK_ARG(ctx && p, "null argument");
    NSDG_CHECK_ARG(p->albedo_kind >= 0 && p->albedo_kind <= 2, "albedo_kind must be 0 (SMU), 1 (SMU2) or 2 (CCSM)");
    NSDG_CHECK_ARG(p->freezing_kind >= 0 && p->freezing_kind <= 1, "freezing_kind must be 0 (linear) or 1 (UNESCO)");
    ctx->column = *p;
    return NSDG_OK;
}

int nsdg_mevp_params_set(nsdg_ctx* ctx, const nsdg_mevp_params* p)
{
    NSDG_CHECK_ARG(ctx && p, "null argument");
    NSDG_CHECK_ARG(p->alpha > 0 && p->beta > 0, "alpha and beta must be positive");
    NSDG_CHECK_ARG(p->aevp_c >= 0 && (p->aevp_c == 0 || p->aevp_alpha_min > 0), "aevp_c must be >= 0 and aevp_alpha_min positive");
    ctx->mevp = *p;
    ctx->pack_dt = 0.; // the packed nodal coefficients were built from the old parameters: repack before iterating
    return NSDG_OK;
}

int nsdg_grid_set(nsdg_ctx* ctx, int32_t nx, int32_t ny, double hx, double hy)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(nx > 0 && ny > 0, "nx and ny must be positive");
    NSDG_CHECK_ARG(hx > 0 && hy > 0, "cell sizes must be positive");
    NSDG_CHECK_ARG((long)(2 * (long)nx + 1) * (2 * (long)ny + 1) < (1L << 31), "grid too large for 32-bit node indices");
    if (ctx->nx != nx || ctx->ny != ny || ctx->hx != hx || ctx->hy != hy)
        ctx->pack_dt = 0.; // packed nodal coefficients belong to the old grid
    ctx->nx = nx;
    ctx->ny = ny;
    ctx->hx = hx;
    ctx->hy = hy;
    return NSDG_OK;
}

int nsdg_mevp_strip_rows_set(nsdg_ctx* ctx, int32_t rows)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(rows >= 0 && rows <= 4096, "rows per strip must be in 0..4096 (0 = automatic)");
    ctx->strip_rows = rows;
    return NSDG_OK;
}

int nsdg_mevp_occupancy_set(nsdg_ctx* ctx, int32_t waves_per_simd)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(waves_per_simd == 1 || waves_per_simd == 2, "waves per SIMD must be 1 or 2");
    ctx->fused_min_waves = waves_per_simd;
    return NSDG_OK;
}

int nsdg_mevp_variant_set(nsdg_ctx* ctx, int32_t variant)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(variant >= 0 && variant <= 4, "variant must be 0, 1, 2, 3 or 4");
    ctx->mevp_variant = variant;
    return NSDG_OK;
}

} // extern "C"
