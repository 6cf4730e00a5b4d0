// nsdg_internal.h -- shared by the translation units of libnsdg.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/nsdg.h"

struct nsdg_comm; // halo.hip

struct nsdg_ctx {
    int device;
    hipStream_t stream;
    nsdg_column_params column;
    nsdg_mevp_params mevp;
    int nx, ny; // local element array
    int row0, ny_global; // its placement in the global domain (analytic forcing providers); ny_global 0 = single domain
    double hx, hy;
    int mevp_variant;
    int strip_rows; // rows per strip of the fused marching kernel (0 = chosen per launch)
    int num_cus;
    double pack_dt; // time step the packed nodal coefficients were built for (0 = never packed)
    int transport_variant, transport_rows; // transport stage kernel: 0 one element per lane / 2 two elements per lane; rows per workgroup
    int fused_min_waves; // register budget of the fused kernel: 1 or 2 waves per SIMD
    int nbounds; // closure of a transport step: bounds of the advected fields (0 = none), nsdg_transport_bounds_set
    nsdg_field_bounds bounds[4];
    // device scratch for small host->device tables (field pointer lists of the transport stage)
    double** d_ptrs;
    nsdg_comm* comm; // row-block communicator (halo.hip), null until nsdg_comm_init*
    int64_t comm_group; // id of the local group the communicator belongs to
    double comm_deadline_s; // upper bound on any wait for a neighbour (0 = for ever)
    // report channel of the mEVP pipelines' bounded waits (mevp_p2p.h): a device counter and a flag in mapped host memory
    unsigned* p2p_count_dev;
    unsigned* p2p_flag_host; // hipHostMalloc'ed; p2p_flag_dev is its device address
    unsigned* p2p_flag_dev;
    unsigned p2p_given_up; // sticky: events seen so far and not yet taken by nsdg_mevp_pipeline_health
};

void nsdg_set_error(const char* fmt, ...);
// NSDG_ERR_HIP (sticky until nsdg_mevp_pipeline_health) if a bounded wait of an mEVP pipeline launched on this context has given up
// in a launch that has completed; does not synchronise
int nsdg_p2p_check(nsdg_ctx* ctx, const char* where);
int nsdg_comm_bounded_drain(nsdg_ctx* ctx); // halo.hip: drain the context's streams within the communicator's deadline

#define NSDG_CHECK_ARG(cond, msg)                                        \
    do {                                                                 \
        if (!(cond)) {                                                   \
            nsdg_set_error("%s: %s", __func__, msg);                    \
            return NSDG_ERR_ARG;                                         \
        }                                                                \
    } while (0)

#define NSDG_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            nsdg_set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
            return NSDG_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define NSDG_CHECK_LAUNCH() NSDG_CHECK_HIP(hipGetLastError())

#define NSDG_NEED_GRID(ctx)                                             \
    do {                                                                \
        NSDG_CHECK_ARG(ctx != nullptr, "null context");                 \
        if ((ctx)->nx <= 0) {                                           \
            nsdg_set_error("%s: nsdg_grid_set was not called", __func__); \
            return NSDG_ERR_STATE;                                      \
        }                                                               \
    } while (0)

static inline int nsdg_div_up(long a, long b) { return (int)((a + b - 1) / b); }
