// mevp.hip -- mEVP stress / velocity sub-cycle on a uniform rectangular mesh (CG2 velocity,
// 8-coefficient DG stress, 3x3 Gauss points) and the per-step helper kernels around it.
//
// No counterpart in the reference snapshot (CMakeLists.txt:43-46); the scheme is stated in
// DESIGN.md section 3.2 and restated on the CPU in oracle/dyn_oracle.c (parity unpinned).
//
// Variant 0 ("two-kernel"): per sub-iteration
//   mevp_stress_kernel   one lane per element: gathers its 9 nodal velocities, forms the strain-rate
//                        coefficients (sparse 8x9 constant operators folded into the instruction
//                        stream), evaluates the VP law at the 9 Gauss points, relaxes the 3x8 stress
//                        coefficients in place.
//   mevp_velocity_kernel one lane per element = its 4 bottom-left-owned nodes: gathers the stress
//                        of the (up to) 4 adjacent elements -- a node-centred gather, no atomics,
//                        deterministic and independent of the decomposition -- and applies the
//                        momentum update.
// Variant 1 ("fused march") lives in mevp_fused.hip.
#include <initializer_list>

#include "mevp_common.h"

namespace nsdg_mevp_detail {

__global__ __launch_bounds__(256) void mevp_stress_kernel(int nx, int ny, int k0, int k1, double ihx, double ihy,
    double ialpha, double dmin2, const double* __restrict__ u, const double* __restrict__ v,
    const double* __restrict__ pg, const double* S11i, const double* S12i, const double* S22i, double* S11, double* S12,
    double* S22)
{
    // S??i may alias S?? (in-place update): every lane reads only its own element before writing it
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = k0 + blockIdx.y * 4 + threadIdx.y;
    if (ix >= nx || iy >= k1)
        return;
    const int ntx = tiles_per_row(nx);
    const long ts = tile_off(ix, iy, ntx, 8), tp = tile_off(ix, iy, ntx, 9);
    const int nn = 2 * nx + 1;
    double ul[9], vl[9], P[9], s11[8], s12[8], s22[8];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const long n = (long)(2 * iy + a / 3) * nn + 2 * ix + a % 3;
        ul[a] = u[n];
        vl[a] = v[n];
    }
    tile_load9(pg, tp, ix & 63, P);
    tile_load8(S11i, ts, s11);
    tile_load8(S12i, ts, s12);
    tile_load8(S22i, ts, s22);
    stress_update(ul, vl, P, ihx, ihy, ialpha, dmin2, s11, s12, s22);
    tile_store8(S11, ts, s11);
    tile_store8(S12, ts, s12);
    tile_store8(S22, ts, s22);
}

__device__ __forceinline__ void load_stress(const double* __restrict__ S11, const double* __restrict__ S12,
    const double* __restrict__ S22, long ts, double (&s11)[8], double (&s12)[8], double (&s22)[8])
{
    tile_load8(S11, ts, s11);
    tile_load8(S12, ts, s12);
    tile_load8(S22, ts, s22);
}

__global__ __launch_bounds__(256) void mevp_velocity_kernel(NodalConsts K, int nx, int ny, int j0, int j1, double hx, double hy,
    const double* __restrict__ S11, const double* __restrict__ S12, const double* __restrict__ S22,
    const double* __restrict__ u_old, const double* __restrict__ v_old, const double* __restrict__ packed,
    double* __restrict__ u_new, double* __restrict__ v_new)
{
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = j0 + blockIdx.y * 4 + threadIdx.y;
    if (ix >= nx || iy >= j1)
        return;
    const int ntx = tiles_per_row(nx);
    const int nn = 2 * nx + 1;
    const long nplane = nodal_plane((long)nn * (2 * ny + 1));
    const double iarea = 1. / (hx * hy);
    double s11[8], s12[8], s22[8];
    double cx, cy;
    // node sums, accumulated in the order below-left, below, left, own (same as the oracle)
    double vx_ = 0., vy_ = 0., exx = 0., exy = 0., eyx = 0., eyy = 0., ccx, ccy;
    const bool hasL = ix > 0, hasB = iy > 0;
    if (hasL && hasB) {
        load_stress(S11, S12, S22, tile_off(ix - 1, iy - 1, ntx, 8), s11, s12, s22);
        node_contrib<8>(s11, s12, s22, hx, hy, cx, cy);
        vx_ += cx, vy_ += cy;
    }
    if (hasB) {
        load_stress(S11, S12, S22, tile_off(ix, iy - 1, ntx, 8), s11, s12, s22);
        node_contrib<6>(s11, s12, s22, hx, hy, cx, cy);
        vx_ += cx, vy_ += cy;
        node_contrib<7>(s11, s12, s22, hx, hy, cx, cy);
        exx += cx, exy += cy;
    }
    if (hasL) {
        load_stress(S11, S12, S22, tile_off(ix - 1, iy, ntx, 8), s11, s12, s22);
        node_contrib<2>(s11, s12, s22, hx, hy, cx, cy);
        vx_ += cx, vy_ += cy;
        node_contrib<5>(s11, s12, s22, hx, hy, cx, cy);
        eyx += cx, eyy += cy;
    }
    load_stress(S11, S12, S22, tile_off(ix, iy, ntx, 8), s11, s12, s22);
    node_contrib<0>(s11, s12, s22, hx, hy, cx, cy);
    vx_ += cx, vy_ += cy;
    node_contrib<1>(s11, s12, s22, hx, hy, cx, cy);
    exx += cx, exy += cy;
    node_contrib<3>(s11, s12, s22, hx, hy, cx, cy);
    eyx += cx, eyy += cy;
    node_contrib<4>(s11, s12, s22, hx, hy, ccx, ccy);

    const long nV = (long)(2 * iy) * nn + 2 * ix; // vertex; EX = nV+1; EY = nV+nn; C = nV+nn+1
    double un, vn, c[6];
    // inverse lumped masses: 4, 2, 2, 1 adjacent elements times LUMP = 1/36, 1/9, 1/9, 4/9 of the cell area
    if (hasL && hasB) { // vertex
        load_nodal(packed, nplane, nV, c);
        node_update_packed(K, c, u_old[nV], v_old[nV], vx_, vy_, 9. * iarea, un, vn);
    } else
        un = vn = 0.;
    u_new[nV] = un, v_new[nV] = vn;
    if (hasB) { // bottom edge-mid
        load_nodal(packed, nplane, nV + 1, c);
        node_update_packed(K, c, u_old[nV + 1], v_old[nV + 1], exx, exy, 4.5 * iarea, un, vn);
    } else
        un = vn = 0.;
    u_new[nV + 1] = un, v_new[nV + 1] = vn;
    if (hasL) { // left edge-mid
        load_nodal(packed, nplane, nV + nn, c);
        node_update_packed(K, c, u_old[nV + nn], v_old[nV + nn], eyx, eyy, 4.5 * iarea, un, vn);
    } else
        un = vn = 0.;
    u_new[nV + nn] = un, v_new[nV + nn] = vn;
    load_nodal(packed, nplane, nV + nn + 1, c); // centre
    node_update_packed(K, c, u_old[nV + nn + 1], v_old[nV + nn + 1], ccx, ccy, 2.25 * iarea, un, vn);
    u_new[nV + nn + 1] = un, v_new[nV + nn + 1] = vn;
    // right column / top row of the local lattice are boundary nodes (v = 0)
    if (ix == nx - 1) {
        u_new[nV + 2] = 0., v_new[nV + 2] = 0.;
        u_new[nV + nn + 2] = 0., v_new[nV + nn + 2] = 0.;
    }
    if (iy == ny - 1) {
        u_new[nV + 2 * nn] = 0., v_new[nV + 2 * nn] = 0.;
        u_new[nV + 2 * nn + 1] = 0., v_new[nV + 2 * nn + 1] = 0.;
        if (ix == nx - 1)
            u_new[nV + 2 * nn + 2] = 0., v_new[nV + 2 * nn + 2] = 0.;
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double psi_rt(int c, double x, double y)
{
    switch (c) {
    case 0: return 1.;
    case 1: return x;
    case 2: return y;
    case 3: return x * x - 1. / 12.;
    case 4: return y * y - 1. / 12.;
    default: return x * y;
    }
}

// nodal average of a DG field at CG2 node (gx, gy): mean over the 1/2/4 adjacent elements of the DG value there
__device__ __forceinline__ double node_average(int nx, int ny, int nc, const double* __restrict__ f, int gx, int gy)
{
    const long N = (long)nx * ny;
    const int ix_hi = gx / 2, ix_lo = (gx % 2 == 0) ? gx / 2 - 1 : gx / 2;
    const int iy_hi = gy / 2, iy_lo = (gy % 2 == 0) ? gy / 2 - 1 : gy / 2;
    double s = 0.;
    int cnt = 0;
    for (int iy = iy_lo; iy <= iy_hi; ++iy)
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
            if (ix < 0 || ix >= nx || iy < 0 || iy >= ny)
                continue;
            const double x = -0.5 + 0.5 * (gx - 2 * ix), y = -0.5 + 0.5 * (gy - 2 * iy);
            const long e = (long)iy * nx + ix;
            double val = 0.;
            for (int c = 0; c < nc; ++c)
                val += f[c * N + e] * psi_rt(c, x, y);
            s += val;
            ++cnt;
        }
    return s / cnt;
}

// per-step momentum coefficients of one node, 6 doubles (layout: mevp_common.h)
__device__ __forceinline__ void pack_node(const nsdg_mevp_params& P, double dt, double u0, double v0, double tax, double tay,
    double uoc, double voc, double cgh, double cga, double* __restrict__ packed, long plane, long n)
{
    const double h = fmax(cgh, P.h_min);
    // Ice-free-node rule (round 5; the shape of the column model's cut-off  c_new < minc || hi < minh,
    // physics/src/modules/NextsimPhysics.cpp:210-219): a node whose mean concentration is below min_conc, whose TRUE thickness
    // cgH / cgA is below min_thick or whose mean thickness is at the mass floor h_min (a made-up mass) is in FREE DRIFT -- full exposure (a = 1) to wind stress and ocean drag, Coriolis, its floor mass
    // -- and does not feel the stress divergence of the neighbouring elements.  That costs the sub-cycle NOTHING: the update is
    //   u' = (K1 h' u + c2 + drag u_o + K3 h' v + div_x / M) / (K2 h' + drag),  drag = cd |v_o - v|,
    // homogeneous of degree 0 in (h', cd, c2, c3, div): scaling the four packed coefficients by 2^100 (exact: a power of two)
    // leaves every other term as it is and weights the divergence by 2^-100 -- below the last bit of the sum.  Both 0: rule off.
    const bool ice_free = (P.min_conc > 0. || P.min_thick > 0.) && (cga < P.min_conc || cgh < P.min_thick * cga || cgh <= P.h_min);
    const double a = ice_free ? 1. : fmin(fmax(cga, 0.), 1.);
    const double mdt = P.rho_ice * h / dt;
    const double cor = P.rho_ice * h * P.fc;
    const double w = ice_free ? 0x1p100 : 1.;
    const double c[6] = { w * h, w * (a * (P.c_ocean * P.rho_ocean)), w * (mdt * u0 + a * tax - cor * voc), w * (mdt * v0 + a * tay + cor * uoc), uoc, voc };
    store_nodal(packed, plane, n, c);
}

__device__ __forceinline__ void wind_tau(double f_atm, double ua, double va, double& tax, double& tay)
{
    const double m = sqrt(ua * ua + va * va);
    tax = f_atm * m * ua;
    tay = f_atm * m * va;
}

__global__ __launch_bounds__(256) void mevp_pack_nodal_kernel(nsdg_mevp_params P, long nnodes, double dt,
    const double* __restrict__ u0, const double* __restrict__ v0, const double* __restrict__ tax,
    const double* __restrict__ tay, const double* __restrict__ uo, const double* __restrict__ vo,
    const double* __restrict__ cgh, const double* __restrict__ cga, double* __restrict__ packed)
{
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes)
        return;
    pack_node(P, dt, u0[n], v0[n], tax[n], tay[n], uo[n], vo[n], cgh[n], cga[n], packed, nodal_plane(nnodes), n);
}

// node_average() on a tile of DG coefficients staged in LDS: tile[c][iy - ey0][ix - ex0]; same loops, same order of
// summation, so the result is bit-identical to the global-memory form
constexpr int PREP_TW = 33, PREP_TH = 3; // element columns / rows under a 64 x 4 block of nodes
__device__ __forceinline__ double node_average_tile(int nx, int ny, const double (*tile)[PREP_TH][PREP_TW], int ex0, int ey0, int gx, int gy)
{
    const int ix_hi = gx / 2, ix_lo = (gx % 2 == 0) ? gx / 2 - 1 : gx / 2;
    const int iy_hi = gy / 2, iy_lo = (gy % 2 == 0) ? gy / 2 - 1 : gy / 2;
    double s = 0.;
    int cnt = 0;
    for (int iy = iy_lo; iy <= iy_hi; ++iy)
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
            if (ix < 0 || ix >= nx || iy < 0 || iy >= ny)
                continue;
            const double x = -0.5 + 0.5 * (gx - 2 * ix), y = -0.5 + 0.5 * (gy - 2 * iy);
            double val = 0.;
            for (int c = 0; c < 6; ++c)
                val += tile[c][iy - ey0][ix - ex0] * psi_rt(c, x, y);
            s += val;
            ++cnt;
        }
    return s / cnt;
}

// All per-step nodal preparation of the momentum equation in one pass over the CG2 lattice: nodal means of
// H and A (dg_to_cg), wind stress, coefficient packing -- without materialising cgH, cgA, tau_a.  A workgroup
// handles 64 x 4 nodes; the DG coefficients of the 33 x 3 elements under them are staged in LDS once (9.5 KB)
// instead of being gathered by every node lane from memory (a vertex node reads 4 elements x 6 coefficients x 2
// fields: the gather form issued ~57 loads per lane and was bound by vector-memory issue, 0.67 ms at 2048^2).
__global__ __launch_bounds__(256) void mevp_prepare_kernel(nsdg_mevp_params P, int nx, int ny, double dt,
    const double* __restrict__ H, const double* __restrict__ A, const double* __restrict__ ua, const double* __restrict__ va,
    const double* __restrict__ uo, const double* __restrict__ vo, const double* __restrict__ u0, const double* __restrict__ v0,
    double* __restrict__ packed)
{
    __shared__ double tile[2][6][PREP_TH][PREP_TW];
    const int gx0 = blockIdx.x * 64, gy0 = blockIdx.y * 4;
    const int ex0 = gx0 / 2 - 1, ey0 = gy0 / 2 - 1;
    const long N = (long)nx * ny;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int k = tid; k < 2 * 6 * PREP_TH * PREP_TW; k += 256) {
        const int col = k % PREP_TW, r = (k / PREP_TW) % PREP_TH, c = (k / (PREP_TW * PREP_TH)) % 6, fld = k / (PREP_TW * PREP_TH * 6);
        const int ix = ex0 + col, iy = ey0 + r;
        double val = 0.;
        if (ix >= 0 && ix < nx && iy >= 0 && iy < ny)
            val = (fld == 0 ? H : A)[c * N + (long)iy * nx + ix];
        tile[fld][c][r][col] = val;
    }
    __syncthreads();
    const int gx = gx0 + threadIdx.x;
    const int gy = gy0 + threadIdx.y;
    const int nn = 2 * nx + 1, nm = 2 * ny + 1;
    if (gx >= nn || gy >= nm)
        return;
    const long n = (long)gy * nn + gx;
    const double cgh = node_average_tile(nx, ny, tile[0], ex0, ey0, gx, gy), cga = node_average_tile(nx, ny, tile[1], ex0, ey0, gx, gy);
    double tax, tay;
    wind_tau(P.c_atm * P.rho_atm, ua[n], va[n], tax, tay);
    pack_node(P, dt, u0[n], v0[n], tax, tay, uo[n], vo[n], cgh, cga, packed, nodal_plane((long)nn * nm), n);
}

// one lane per CG2 node
__global__ __launch_bounds__(256) void dg_to_cg_kernel(int nx, int ny, int nc, const double* __restrict__ f, double* __restrict__ g)
{
    const int gx = blockIdx.x * 64 + threadIdx.x;
    const int gy = blockIdx.y * 4 + threadIdx.y;
    const int nn = 2 * nx + 1, nm = 2 * ny + 1;
    if (gx >= nn || gy >= nm)
        return;
    g[(long)gy * nn + gx] = node_average(nx, ny, nc, f, gx, gy);
}

__global__ __launch_bounds__(256) void ice_strength_kernel(int nx, int ny, int j0, int j1, double pstar, double compaction,
    const double* __restrict__ H, const double* __restrict__ A, double* __restrict__ pg)
{
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = j0 + blockIdx.y * 4 + threadIdx.y;
    if (ix >= nx || iy >= j1)
        return;
    const long N = (long)nx * ny;
    const long e = (long)iy * nx + ix;
    const long tp = tile_off(ix, iy, tiles_per_row(nx), 9);
    double hc[6], ac[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        hc[c] = H[c * N + e];
        ac[c] = A[c * N + e];
    }
    double pq[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        double h = 0., a = 0.;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            FMA_TAB(h, PSI_G3[q][c], hc[c]);
            FMA_TAB(a, PSI_G3[q][c], ac[c]);
        }
        h = fmax(h, 0.);
        a = fmin(fmax(a, 0.), 1.);
        pq[q] = pstar * h * exp(-compaction * (1. - a));
    }
    tile_store9(pg, tp, ix & 63, pq);
}

__global__ __launch_bounds__(256) void wind_stress_kernel(long n, double f_atm, const double* __restrict__ ua,
    const double* __restrict__ va, double* __restrict__ tax, double* __restrict__ tay)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    wind_tau(f_atm, ua[i], va[i], tax[i], tay[i]);
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

// defined in mevp_fused.hip
int nsdg_launch_mevp_fused(nsdg_ctx* ctx, int k0, int j0, int j1, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg);

// the tiled arrays are accessed 16 bytes at a time
static inline bool aligned16(std::initializer_list<const void*> ptrs)
{
    for (const void* p : ptrs)
        if ((uintptr_t)p & 15)
            return false;
    return true;
}
#define NSDG_CHECK_TILED(...) NSDG_CHECK_ARG(aligned16({ __VA_ARGS__ }), "tiled arrays (stress, ice strength) must be 16-byte aligned")

// defined in mevp_fused4.hip: a pass of nst = 2, 3 or 4 sub-iterations on the rows [j0, j1) and, if j0b < j1b, on a second disjoint range
int nsdg_launch_mevp_fused4_ranges(nsdg_ctx* ctx, int nst, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg);

static NodalConsts nodal_consts(const nsdg_ctx* ctx) { return nsdg_nodal_consts(ctx); }

// the adaptive form of alpha, beta (mevp_common.h) exists in the marching kernels only
#define NSDG_NOT_ADAPTIVE(ctx)                                                                                                  \
    do {                                                                                                                        \
        if (nsdg_adaptive(ctx)) {                                                                                               \
            nsdg_set_error("%s: the two-kernel form has no adaptive alpha / beta (nsdg_mevp_params.aevp_c > 0): use nsdg_mevp_iterate*", __func__); \
            return NSDG_ERR_STATE;                                                                                              \
        }                                                                                                                       \
    } while (0)

extern "C" {

int64_t nsdg_tiled_len(int32_t nx, int32_t ny, int32_t nc) { return (int64_t)ny * tiles_per_row(nx) * nc * 64; }

int nsdg_dg_to_cg(nsdg_ctx* ctx, int32_t ncoef, const double* f_dg, double* f_cg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(ncoef == 1 || ncoef == 3 || ncoef == 6, "ncoef must be 1, 3 or 6");
    NSDG_CHECK_ARG(f_dg && f_cg, "null field pointer");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(2 * ctx->nx + 1, 64), nsdg_div_up(2 * ctx->ny + 1, 4));
    hipLaunchKernelGGL(dg_to_cg_kernel, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, ncoef, f_dg, f_cg);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_ice_strength(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* H, const double* A, double* pg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(H && A && pg, "null field pointer");
    NSDG_CHECK_TILED(pg);
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(j1 - j0, 4));
    hipLaunchKernelGGL(ice_strength_kernel, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, j0, j1, ctx->mevp.pstar,
        ctx->mevp.compaction, H, A, pg);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_wind_stress(nsdg_ctx* ctx, int64_t nnodes, const double* ua, const double* va, double* tax, double* tay)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(nnodes >= 0, "negative node count");
    NSDG_CHECK_ARG(ua && va && tax && tay, "null field pointer");
    if (nnodes == 0)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(wind_stress_kernel, dim3(nsdg_div_up(nnodes, 256)), dim3(256), 0, ctx->stream, (long)nnodes,
        ctx->mevp.c_atm * ctx->mevp.rho_atm, ua, va, tax, tay);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

static int launch_stress(nsdg_ctx* ctx, int k0, int k1, const double* u, const double* v, const double* pg, const double* s11i,
    const double* s12i, const double* s22i, double* s11, double* s12, double* s22)
{
    const dim3 block(64, 4), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(k1 - k0, 4));
    hipLaunchKernelGGL(mevp_stress_kernel, grid, block, 0, ctx->stream, ctx->nx, ctx->ny, k0, k1, 1. / ctx->hx, 1. / ctx->hy,
        1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, u, v, pg, s11i, s12i, s22i, s11, s12, s22);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_mevp_stress(nsdg_ctx* ctx, int32_t k0, int32_t k1, const double* u, const double* v, const double* pg, double* s11,
    double* s12, double* s22)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= k0 && k0 <= k1 && k1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(u && v && pg && s11 && s12 && s22, "null field pointer");
    NSDG_CHECK_TILED(pg, s11, s12, s22);
    NSDG_NOT_ADAPTIVE(ctx);
    if (k0 == k1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    return launch_stress(ctx, k0, k1, u, v, pg, s11, s12, s22, s11, s12, s22);
}

int nsdg_mevp_pack_nodal(nsdg_ctx* ctx, double dt, const double* u0, const double* v0, const double* tax, const double* tay,
    const double* uo, const double* vo, const double* cgh, const double* cga, double* packed)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(u0 && v0 && tax && tay && uo && vo && cgh && cga && packed, "null field pointer");
    NSDG_CHECK_ARG(dt > 0, "dt must be positive");
    NSDG_CHECK_ARG(((uintptr_t)packed & 15) == 0, "packed must be 16-byte aligned");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const long nnodes = (long)(2 * ctx->nx + 1) * (2 * ctx->ny + 1);
    hipLaunchKernelGGL(mevp_pack_nodal_kernel, dim3(nsdg_div_up(nnodes, 256)), dim3(256), 0, ctx->stream, ctx->mevp, nnodes, dt, u0,
        v0, tax, tay, uo, vo, cgh, cga, packed);
    NSDG_CHECK_LAUNCH();
    ctx->pack_dt = dt; // the launch constants K1, K2 of the velocity update belong to this packing
    return NSDG_OK;
}

int nsdg_mevp_prepare(nsdg_ctx* ctx, double dt, const double* H, const double* A, const double* ua, const double* va,
    const double* uo, const double* vo, const double* u0, const double* v0, double* packed)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(H && A && ua && va && uo && vo && u0 && v0 && packed, "null field pointer");
    NSDG_CHECK_ARG(dt > 0, "dt must be positive");
    NSDG_CHECK_ARG(((uintptr_t)packed & 15) == 0, "packed must be 16-byte aligned");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(2 * ctx->nx + 1, 64), nsdg_div_up(2 * ctx->ny + 1, 4));
    hipLaunchKernelGGL(mevp_prepare_kernel, grid, block, 0, ctx->stream, ctx->mevp, ctx->nx, ctx->ny, dt, H, A, ua, va, uo, vo, u0, v0,
        packed);
    NSDG_CHECK_LAUNCH();
    ctx->pack_dt = dt;
    return NSDG_OK;
}

int nsdg_mevp_velocity(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11, const double* s12, const double* s22,
    const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed, "null field pointer");
    NSDG_CHECK_TILED(s11, s12, s22);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_NOT_ADAPTIVE(ctx);
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    const dim3 block(64, 4), grid(nsdg_div_up(ctx->nx, 64), nsdg_div_up(j1 - j0, 4));
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_velocity: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    hipLaunchKernelGGL(mevp_velocity_kernel, grid, block, 0, ctx->stream, nodal_consts(ctx), ctx->nx, ctx->ny, j0, j1, ctx->hx, ctx->hy, s11, s12, s22,
        u_old, v_old, packed, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_mevp_iterate(nsdg_ctx* ctx, int32_t k0, int32_t j0, int32_t j1, const double* s11i, const double* s12i,
    const double* s22i, double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new,
    double* v_new, const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= k0 && k0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "need 0 <= k0 <= j0 <= j1 <= ny");
    NSDG_CHECK_ARG(k0 == j0 - 1 || (k0 == 0 && j0 == 0), "need k0 == j0 - 1 (one ghost row below) or k0 == j0 == 0");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg,
        "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (k0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (ctx->mevp_variant >= 1 || nsdg_adaptive(ctx)) // variants 2-4 use the single-iteration fused kernel for one sub-iteration; so does variant 0 in the adaptive form
        return nsdg_launch_mevp_fused(ctx, k0, j0, j1, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
    int rc = launch_stress(ctx, k0, j1, u_old, v_old, pg, s11i, s12i, s22i, s11, s12, s22);
    if (rc)
        return rc;
    return nsdg_mevp_velocity(ctx, j0, j1, s11, s12, s22, u_old, v_old, u_new, v_new, packed);
}

int nsdg_mevp_iterate2(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(j0 == 0 || j0 >= 2, "need two ghost rows below the owned rows (or j0 == 0 at the physical boundary)");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg,
        "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate2: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (ctx->mevp_variant >= 2) // a pass of the stage-per-wave pipeline with two stages
        return nsdg_launch_mevp_fused4_ranges(ctx, 2, j0, j1, 0, 0, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
    nsdg_set_error("nsdg_mevp_iterate2: select variant 2, 3 or 4 (nsdg_mevp_variant_set) or call nsdg_mevp_iterate twice");
    return NSDG_ERR_STATE;
}

int nsdg_mevp_iterate3(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(j0 == 0 || j0 >= 3, "need three ghost rows below the owned rows (or j0 == 0 at the physical boundary)");
    NSDG_CHECK_ARG(j1 == ctx->ny || j1 + 2 <= ctx->ny, "need two ghost rows above the owned rows (or j1 == ny at the physical boundary)");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg,
        "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate3: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (ctx->mevp_variant >= 3) // ... with three stages
        return nsdg_launch_mevp_fused4_ranges(ctx, 3, j0, j1, 0, 0, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
    nsdg_set_error("nsdg_mevp_iterate3: select variant 3 or 4 (nsdg_mevp_variant_set)");
    return NSDG_ERR_STATE;
}

int nsdg_mevp_iterate3_pair(nsdg_ctx* ctx, int32_t j0a, int32_t j1a, int32_t j0b, int32_t j1b, const double* s11i, const double* s12i,
    const double* s22i, double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    for (int k = 0; k < 2; ++k) {
        const int j0 = k ? j0b : j0a, j1 = k ? j1b : j1a;
        NSDG_CHECK_ARG(0 <= j0 && j0 < j1 && j1 <= ctx->ny, "row range outside the local array (or empty)");
        NSDG_CHECK_ARG(j0 == 0 || j0 >= 3, "need three ghost rows below the rows of a range (or j0 == 0 at the physical boundary)");
        NSDG_CHECK_ARG(j1 == ctx->ny || j1 + 2 <= ctx->ny, "need two ghost rows above the rows of a range (or j1 == ny at the physical boundary)");
    }
    NSDG_CHECK_ARG(j1a <= j0b || j1b <= j0a, "the two row ranges must be disjoint");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg, "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate3_pair: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (ctx->mevp_variant < 3) {
        nsdg_set_error("nsdg_mevp_iterate3_pair: select variant 3 or 4 (nsdg_mevp_variant_set)");
        return NSDG_ERR_STATE;
    }
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    return nsdg_launch_mevp_fused4_ranges(ctx, 3, j0a, j1a, j0b, j1b, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
}

int nsdg_mevp_iterate4(nsdg_ctx* ctx, int32_t j0, int32_t j1, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(0 <= j0 && j0 <= j1 && j1 <= ctx->ny, "row range outside the local array");
    NSDG_CHECK_ARG(j0 == 0 || j0 >= 4, "need four ghost rows below the owned rows (or j0 == 0 at the physical boundary)");
    NSDG_CHECK_ARG(j1 == ctx->ny || j1 + 3 <= ctx->ny, "need three ghost rows above the owned rows (or j1 == ny at the physical boundary)");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg,
        "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate4: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (j0 == j1)
        return NSDG_OK;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (ctx->mevp_variant >= 4)
        return nsdg_launch_mevp_fused4_ranges(ctx, 4, j0, j1, 0, 0, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
    nsdg_set_error("nsdg_mevp_iterate4: select variant 4 (nsdg_mevp_variant_set)");
    return NSDG_ERR_STATE;
}

int nsdg_mevp_iterate4_pair(nsdg_ctx* ctx, int32_t j0a, int32_t j1a, int32_t j0b, int32_t j1b, const double* s11i, const double* s12i,
    const double* s22i, double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    NSDG_NEED_GRID(ctx);
    for (int k = 0; k < 2; ++k) {
        const int j0 = k ? j0b : j0a, j1 = k ? j1b : j1a;
        NSDG_CHECK_ARG(0 <= j0 && j0 < j1 && j1 <= ctx->ny, "row range outside the local array (or empty)");
        NSDG_CHECK_ARG(j0 == 0 || j0 >= 4, "need four ghost rows below the rows of a range (or j0 == 0 at the physical boundary)");
        NSDG_CHECK_ARG(j1 == ctx->ny || j1 + 3 <= ctx->ny, "need three ghost rows above the rows of a range (or j1 == ny at the physical boundary)");
    }
    NSDG_CHECK_ARG(j1a <= j0b || j1b <= j0a, "the two row ranges must be disjoint");
    NSDG_CHECK_ARG(s11i && s12i && s22i && s11 && s12 && s22 && u_old && v_old && u_new && v_new && packed && pg, "null field pointer");
    NSDG_CHECK_TILED(s11i, s12i, s22i, s11, s12, s22, pg);
    NSDG_CHECK_ARG(u_new != u_old && v_new != v_old, "u_new/v_new must not alias u_old/v_old");
    NSDG_CHECK_ARG(s11 != s11i && s12 != s12i && s22 != s22i, "the output stress must not alias the input stress");
    if (!(ctx->pack_dt > 0)) {
        nsdg_set_error("nsdg_mevp_iterate4_pair: nsdg_mevp_pack_nodal was not called on this context");
        return NSDG_ERR_STATE;
    }
    if (ctx->mevp_variant < 4) {
        nsdg_set_error("nsdg_mevp_iterate4_pair: select variant 4 (nsdg_mevp_variant_set)");
        return NSDG_ERR_STATE;
    }
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    return nsdg_launch_mevp_fused4_ranges(ctx, 4, j0a, j1a, j0b, j1b, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
}

int nsdg_mevp_subcycle(nsdg_ctx* ctx, double dt, int32_t nsub, double* s11, double* s12, double* s22, double* u, double* v,
    const double* u0, const double* v0, const double* tax, const double* tay, const double* uo, const double* vo,
    const double* cgh, const double* cga, const double* pg, double* scratch)
{
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(nsub >= 0, "negative sub-iteration count");
    NSDG_CHECK_ARG(u && v && s11 && s12 && s22 && scratch, "null field pointer");
    // u0/v0 may alias u/v: they are consumed by the coefficient packing before the first pass writes u, v
    NSDG_CHECK_ARG(u0 && v0, "null field pointer");
    NSDG_CHECK_ARG(((uintptr_t)scratch & 15) == 0, "scratch must be 16-byte aligned");
    const long nnodes = (long)(2 * ctx->nx + 1) * (2 * ctx->ny + 1);
    const long M = nsdg_tiled_len(ctx->nx, ctx->ny, 8);
    // scratch: [packed nodal 6*nnodes (8*nnodes reserved)][u, v ping-pong 2*nnodes][stress ping-pong 3*M]
    double* packed = scratch;
    double *ua = u, *va = v, *ub = scratch + 8 * nnodes, *vb = ub + nnodes;
    double *sa[3] = { s11, s12, s22 };
    double *sb[3] = { vb + nnodes, vb + nnodes + M, vb + nnodes + 2 * M };
    int rc = nsdg_p2p_check(ctx, __func__); // a pipeline wait that gave up in an earlier launch: the fields this call would start from are wrong
    if (rc)
        return rc;
    rc = nsdg_mevp_pack_nodal(ctx, dt, u0, v0, tax, tay, uo, vo, cgh, cga, packed);
    if (rc)
        return rc;
    for (int it = 0; it < nsub; ++it) {
        if (ctx->mevp_variant >= 4 && it + 3 < nsub) { // four sub-iterations per pass
            rc = nsdg_mevp_iterate4(ctx, 0, ctx->ny, sa[0], sa[1], sa[2], sb[0], sb[1], sb[2], ua, va, ub, vb, packed, pg);
            it += 3;
        } else if (ctx->mevp_variant >= 3 && it + 2 < nsub) { // three sub-iterations per pass
            rc = nsdg_mevp_iterate3(ctx, 0, ctx->ny, sa[0], sa[1], sa[2], sb[0], sb[1], sb[2], ua, va, ub, vb, packed, pg);
            it += 2;
        } else if (ctx->mevp_variant >= 2 && it + 1 < nsub) { // two sub-iterations per pass
            rc = nsdg_mevp_iterate2(ctx, 0, ctx->ny, sa[0], sa[1], sa[2], sb[0], sb[1], sb[2], ua, va, ub, vb, packed, pg);
            ++it;
        } else
            rc = nsdg_mevp_iterate(ctx, 0, 0, ctx->ny, sa[0], sa[1], sa[2], sb[0], sb[1], sb[2], ua, va, ub, vb, packed, pg);
        if (rc)
            return rc;
        double* t = ua; ua = ub; ub = t;
        t = va; va = vb; vb = t;
        for (int k = 0; k < 3; ++k) { t = sa[k]; sa[k] = sb[k]; sb[k] = t; }
    }
    if (ua != u) { // odd number of sub-iterations: the results sit in scratch
        NSDG_CHECK_HIP(hipMemcpyAsync(u, ua, nnodes * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        NSDG_CHECK_HIP(hipMemcpyAsync(v, va, nnodes * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        NSDG_CHECK_HIP(hipMemcpyAsync(s11, sa[0], M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        NSDG_CHECK_HIP(hipMemcpyAsync(s12, sa[1], M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        NSDG_CHECK_HIP(hipMemcpyAsync(s22, sa[2], M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return NSDG_OK;
}

} // extern "C"
