// mevp_fused.hip -- variant 1 of the mEVP sub-iteration: ONE kernel per sub-iteration.
//
// A wave owns a strip of 63 element columns x R element rows and marches through it bottom to top,
// one element per lane per row.  Per row each lane
//   1. gathers its 9 nodal velocities, the 9 Gauss-point strengths and the 24 old stress coefficients,
//   2. relaxes the stress (same inlined stress_update as the two-kernel variant) and stores it,
//   3. forms the element's 18 contributions -(sigma, grad phi_a) to its 9 nodes in registers,
//   4. assembles the divergence at its 4 bottom-left-owned nodes from
//        - its own contributions,
//        - the left neighbour's right-column contributions   -> one-lane wavefront shift (DPP/bpermute),
//        - the contributions of the row below to its top nodes -> carried in registers from the
//          previous iteration of the march (and shifted by one lane for the below-left element),
//      and applies the momentum update to those 4 nodes.
// Nothing but the final stress and velocity ever goes to memory: per element-sub-iteration the kernel
// moves 8 (u,v) + 9 (P) + 24 (S in) + 24 (nodal coefficients) loads and 24 + 8 stores = 776 B of
// unique data, below the 896 B algorithmic figure of SURVEY.md section 8(d) (which counts H, A instead of the
// pre-evaluated P), against 1152 B for the two-kernel variant.
//
// Redundancy instead of synchronisation: lane 0 of every wave recomputes the column left of the
// strip (so 63 of 64 lanes own a column) and every strip recomputes the row below it (R+1 rows of
// stress for R rows of output).  The recomputed values are bit-identical to their owner's, which is
// why the stress is updated out of place (S_in -> S_out): a recomputing strip must never read a value
// its owner has already overwritten.  No barrier, no atomics, no inter-workgroup communication: every
// wave is independent and runs to completion.
#include "mevp_common.h"


namespace nsdg_mevp_detail {

struct StressPtrs {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

__device__ __forceinline__ double shift_up(double x)
{
    // value of lane-1 (lane 0 keeps its own; it is a redundant column whose result is discarded)
    return lane_from_left(x);
}

template <int MINW, bool AD>
__global__ __launch_bounds__(256, MINW) void mevp_fused_kernel(NodalConsts K, AdaptConsts AC, int nx, int ny, int k0, int j0, int j1, int R, int ncw, double hx,
    double hy, double ialpha, double dmin2, StressPtrs S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = wave / ncw, cw = wave - strip * ncw;
    const int y0 = k0 + strip * R;
    if (y0 >= j1)
        return; // wave-uniform
    const int y1 = min(y0 + R, j1);
    const int ixr = cw * 63 - 1 + lane;
    const bool valid = ixr >= 0 && ixr < nx; // lanes outside the array load a clamped column and store nothing
    const bool own = valid && lane > 0;
    const int ix = min(max(ixr, 0), nx - 1);
    const bool hasL = ix > 0;
    const int ntx = tiles_per_row(nx);
    const int nn = 2 * nx + 1;
    const long nplane = nodal_plane((long)nn * (2 * ny + 1));
    const double ihx = 1. / hx, ihy = 1. / hy, iarea = ihx * ihy;

    // contributions of the row below to its top-row nodes: b6 (top-left), b7 (top-mid) of my column and
    // bl8 = top-right of the column to my left
    double b6x = 0., b6y = 0., b7x = 0., b7y = 0., bl8x = 0., bl8y = 0.;
    double qb = 0., qbl = 0.; // adaptive form: the offers q_e of the element below and below-left

    for (int iy = (y0 > k0 ? y0 - 1 : y0); iy < y1; ++iy) {
        const bool prologue = iy < y0; // recomputed row owned by the strip below: nothing is stored
        const long ts = tile_off(ix, iy, ntx, 8), tp = tile_off(ix, iy, ntx, 9);
        const long nV = (long)(2 * iy) * nn + 2 * ix;
        double ul[9], vl[9], Pq[9], s11[8], s12[8], s22[8];
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const long n = nV + (a / 3) * nn + a % 3;
            ul[a] = u_old[n];
            vl[a] = v_old[n];
        }
        tile_load9(pg, tp, ix & 63, Pq);
        double qe = 0.; // adaptive form: this element's offer q_e = alpha_e h'_c of this sub-iteration (mevp_common.h)
        if constexpr (AD) {
            // local, solution-adaptive alpha (mevp_common.h); h' of the element's centre node is its packed coefficient [0]
            const double hc = packed[nodal_off(nV + nn + 1, nplane)];
            double r11[8], r12[8], r22[8], ialpha;
            stress_projected_adaptive(ul, vl, Pq, ihx, ihy, dmin2, hc, AC, r11, r12, r22, qe, ialpha);
            tile_load8(S.i11, ts, s11);
            tile_load8(S.i12, ts, s12);
            tile_load8(S.i22, ts, s22);
            stress_relax_adaptive(ialpha, r11, r12, r22, s11, s12, s22);
        } else if constexpr (MINW >= 2) {
            // 2 waves/SIMD build: stage the loads so that the live set stays under 256 registers -- the old
            // stress is fetched only after the projected stress is formed, the partner wave covers the latency
            double r11[8], r12[8], r22[8];
            stress_projected(ul, vl, Pq, ihx, ihy, ialpha, dmin2, r11, r12, r22);
            asm volatile("" ::: "memory");
            tile_load8(S.i11, ts, s11);
            tile_load8(S.i12, ts, s12);
            tile_load8(S.i22, ts, s22);
            stress_relax(ialpha, r11, r12, r22, s11, s12, s22);
        } else {
            tile_load8(S.i11, ts, s11);
            tile_load8(S.i12, ts, s12);
            tile_load8(S.i22, ts, s22);
            stress_update(ul, vl, Pq, ihx, ihy, ialpha, dmin2, s11, s12, s22);
        }
        if (!prologue && own) {
            tile_store8(S.o11, ts, s11);
            tile_store8(S.o12, ts, s12);
            tile_store8(S.o22, ts, s22);
        }
        double cx[9], cy[9];
        node_contrib_all(s11, s12, s22, hx, hy, cx, cy);
        if constexpr (MINW >= 2)
            asm volatile("" ::: "memory"); // keep the nodal-coefficient loads below this point
        // wavefront-level edge exchange: right-column contributions of the element to my left
        const double l2x = shift_up(cx[2]), l2y = shift_up(cy[2]);
        const double l5x = shift_up(cx[5]), l5y = shift_up(cy[5]);
        const double ql = AD ? shift_up(qe) : 0.;

        if (!prologue && iy >= j0) { // wave-uniform
            const bool hasB = iy > 0;
            double un, vn, c[6];
            // vertex: below-left + below + left + own (the oracle's summation order)
            if (hasL && hasB) {
                load_nodal(packed, nplane, nV, c);
                if constexpr (AD)
                    node_update_packed_adaptive(K, c, ul[0], vl[0], ((bl8x + b6x) + l2x) + cx[0], ((bl8y + b6y) + l2y) + cy[0], 9. * iarea,
                        __builtin_fmax(__builtin_fmax(qbl, qb), __builtin_fmax(ql, qe)), AC.amin, un, vn);
                else
                    node_update_packed(K, c, ul[0], vl[0], ((bl8x + b6x) + l2x) + cx[0], ((bl8y + b6y) + l2y) + cy[0], 9. * iarea, un, vn);
            } else
                un = vn = 0.;
            if (own)
                u_new[nV] = un, v_new[nV] = vn;
            // bottom edge-mid: below + own
            if (hasB) {
                load_nodal(packed, nplane, nV + 1, c);
                if constexpr (AD)
                    node_update_packed_adaptive(K, c, ul[1], vl[1], b7x + cx[1], b7y + cy[1], 4.5 * iarea, __builtin_fmax(qb, qe), AC.amin, un, vn);
                else
                    node_update_packed(K, c, ul[1], vl[1], b7x + cx[1], b7y + cy[1], 4.5 * iarea, un, vn);
            } else
                un = vn = 0.;
            if (own)
                u_new[nV + 1] = un, v_new[nV + 1] = vn;
            // left edge-mid: left + own
            if (hasL) {
                load_nodal(packed, nplane, nV + nn, c);
                if constexpr (AD)
                    node_update_packed_adaptive(K, c, ul[3], vl[3], l5x + cx[3], l5y + cy[3], 4.5 * iarea, __builtin_fmax(ql, qe), AC.amin, un, vn);
                else
                    node_update_packed(K, c, ul[3], vl[3], l5x + cx[3], l5y + cy[3], 4.5 * iarea, un, vn);
            } else
                un = vn = 0.;
            if (own)
                u_new[nV + nn] = un, v_new[nV + nn] = vn;
            // centre: own
            load_nodal(packed, nplane, nV + nn + 1, c);
            if constexpr (AD)
                node_update_packed_adaptive(K, c, ul[4], vl[4], cx[4], cy[4], 2.25 * iarea, qe, AC.amin, un, vn);
            else
                node_update_packed(K, c, ul[4], vl[4], cx[4], cy[4], 2.25 * iarea, un, vn);
            if (own) {
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
        }
        // carry the top-row contributions to the next row of the march
        b6x = cx[6], b6y = cy[6], b7x = cx[7], b7y = cy[7];
        bl8x = shift_up(cx[8]), bl8y = shift_up(cy[8]);
        if constexpr (AD)
            qb = qe, qbl = ql;
    }
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

int nsdg_launch_mevp_fused(nsdg_ctx* ctx, int k0, int j0, int j1, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new,
    const double* packed, const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, 63); // 63 owned columns per wave
    int R = ctx->strip_rows;
    if (R <= 0) {
        // Automatic strip height.  Every wave marches R+1 rows (one redundant), and the launch runs in
        // ceil(waves / resident wave slots) rounds, so the time is ~ rounds * (R+1) row-times: pick the R
        // that minimises it (measured on 2048^2: R = 17 -> 2 full rounds, 8 % faster than R = 4 with its
        // 8.25 rounds and 25 % redundant rows).  A single round is charged 1.5 row-times because one
        // straggling wave then ends the launch alone.
        const long slots = 2L * 4 * ctx->num_cus; // 2 waves per SIMD at ~204 VGPRs
        const int rows = j1 - k0;
        double best = 1e30;
        R = 4;
        for (int r = 2; r <= 64; ++r) {
            const long waves = (long)nsdg_div_up(rows, r) * ncw;
            const long rounds = (waves + slots - 1) / slots;
            const double cost = rounds * (r + 1.0) + (rounds == 1 ? 1.5 : 0.0);
            if (cost < best) {
                best = cost;
                R = r;
            }
        }
    }
    const int nstrips = nsdg_div_up(j1 - k0, R);
    const long nwaves = (long)ncw * nstrips;
    const StressPtrs S = { s11i, s12i, s22i, s11, s12, s22 };
    const NodalConsts K = nsdg_nodal_consts(ctx);
    const AdaptConsts AC = nsdg_adapt_consts(ctx);
    const double ialpha = 1. / ctx->mevp.alpha, dmin2 = ctx->mevp.delta_min * ctx->mevp.delta_min;
    const dim3 grid(nsdg_div_up(nwaves, 4)), block(256);
    // two register budgets of the same kernel: 1 wave/SIMD (no spills) or 2 waves/SIMD (a few scratch spills); the adaptive form (local
    // alpha, beta: mevp_common.h) runs at 1 wave/SIMD
    if (nsdg_adaptive(ctx))
        hipLaunchKernelGGL((mevp_fused_kernel<1, true>), grid, block, 0, ctx->stream, K, AC, ctx->nx, ctx->ny, k0, j0, j1, R, ncw, ctx->hx, ctx->hy, ialpha, dmin2, S,
            u_old, v_old, packed, pg, u_new, v_new);
    else if (ctx->fused_min_waves >= 2)
        hipLaunchKernelGGL((mevp_fused_kernel<2, false>), grid, block, 0, ctx->stream, K, AC, ctx->nx, ctx->ny, k0, j0, j1, R, ncw, ctx->hx, ctx->hy, ialpha, dmin2, S,
            u_old, v_old, packed, pg, u_new, v_new);
    else
        hipLaunchKernelGGL((mevp_fused_kernel<1, false>), grid, block, 0, ctx->stream, K, AC, ctx->nx, ctx->ny, k0, j0, j1, R, ncw, ctx->hx, ctx->hy, ialpha, dmin2, S,
            u_old, v_old, packed, pg, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}
