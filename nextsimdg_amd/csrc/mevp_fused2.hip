// mevp_fused2.hip -- variant 2 of the mEVP sub-cycle: TWO sub-iterations per kernel pass.
//
// The single-iteration fused kernel (mevp_fused.hip) is HBM-bound at ~87 % of the copy ceiling, so
// the only way to go faster is to move fewer bytes.  Here the march is software-pipelined in pseudo
// time: at march step t a lane performs
//     A(t): sub-iteration p   on element row t      (stress S^p(t), velocity u^p at the row's owned nodes)
//     B(t): sub-iteration p+1 on element row t - 1  (stress S^{p+1}(t-1), velocity u^{p+1})
// B(t) needs u^p on the three node rows of element row t-1: two of them were produced by A(t-1) and are
// still in registers, the third was produced by A(t) a moment ago.  S^p and u^p therefore never touch
// memory: per TWO sub-iterations an element reads u,v (64 B), P (72), S (192) and the packed nodal
// coefficients (192) once and writes S (192) and u,v (64) once -- 388 B per element-sub-iteration
// instead of 776.
//
// Redundancy instead of synchronisation, one level deeper than in the single-iteration kernel: a wave
// owns 61 of its 64 columns (lanes 0,1 recompute the two columns to its left, lane 63 the column to
// its right) and a strip of R rows runs A on rows y0-2 .. y1 and B on rows y0-1 .. y1-1.  All
// recomputed values are bit-identical to their owners', the stress and the velocity are written out of
// place, and nothing is exchanged between waves.  The arithmetic is the same sequence of inlined
// functions as in the other variants, so two passes of variant 1 and one pass of variant 2 agree to
// the last bit.
//
// Row ranges: a launch updates the owned element rows [j0, j1) of the local array and reads two rows
// below and one row above them (S, P, u: rows j0-2 .. j1, node rows 2(j0-2) .. 2*j1+2).  Where those rows
// do not exist the edge of the local array is the physical boundary, so a rank of a row-block
// decomposition keeps 2 ghost element rows below and 1 above and refreshes them (stress and velocity)
// after every pass.
#include "mevp_common.h"

#ifdef NSDG_STAMPS
__device__ unsigned nsdg_stamp_acc[64 * 16];
#endif

namespace nsdg_mevp_detail {

struct StressPtrs2 {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

__device__ __forceinline__ double from_left(double x) { return lane_from_left(x); }
__device__ __forceinline__ double from_right(double x) { return lane_from_right(x); }

// Everything of one element row that sub-iteration p+1 needs from sub-iteration p.  Two such sets
// alternate between "being written by A(t)" and "being read by B(t+1)", so nothing is copied when the
// march advances.
struct RowCarry {
    double s11[8], s12[8], s22[8]; // S^p of the row (relaxed in place to S^{p+1} by B)
    double P[9]; // ice strength at the Gauss points
    double c[4][6]; // packed momentum coefficients of the 4 owned nodes (V, EX, EY, C)
    double u[4], v[4]; // u^p, v^p at those nodes
};

struct MarchConst {
    NodalConsts K;
    int nx, ny, y0, y1, tbeg, tend, ix, ntx, nn;
    long nplane; // doubles between two pair planes of the packed nodal coefficients
    bool own, hasL, lastcol;
    double hx, hy, ihx, ihy, iarea, ialpha, dmin2;
};

// contributions of a row to its top nodes, carried to the next row of the march
struct TopCarry {
    double x6 = 0., y6 = 0., x7 = 0., y7 = 0., xl8 = 0., yl8 = 0.; // 6: top-left, 7: top-mid of my column, 8 of the left column
};

// One march step: A(t) = sub-iteration p on row t into `cur`; B(t) = sub-iteration p+1 on row t-1 from `prev`.
__device__ __forceinline__ void march_step(const MarchConst& M, int t, RowCarry& cur, RowCarry& prev, TopCarry& ca, TopCarry& cb,
    const StressPtrs2& S, const double* __restrict__ u_old, const double* __restrict__ v_old, const double* __restrict__ packed,
    const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new NSDG_STAMP_ARGS)
{
    const int nn = M.nn, ix = M.ix;
    NSDG_STAMP(0);
    // ---------------------------------------------------------------------- A(t): sub-iteration p on row t
    if (t <= M.tend) { // the last march step only drains B
        const long ts = tile_off(ix, t, M.ntx, 8), tp = tile_off(ix, t, M.ntx, 9);
        const long nV = (long)(2 * t) * nn + 2 * ix;
        double ul[9], vl[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const long n = nV + (a / 3) * nn + a % 3;
            ul[a] = u_old[n];
            vl[a] = v_old[n];
        }
        tile_load9(pg, tp, ix & 63, cur.P);
        tile_load8(S.i11, ts, cur.s11);
        tile_load8(S.i12, ts, cur.s12);
        tile_load8(S.i22, ts, cur.s22);
        load_nodal(packed, M.nplane, nV, cur.c[0]);
        load_nodal(packed, M.nplane, nV + 1, cur.c[1]);
        load_nodal(packed, M.nplane, nV + nn, cur.c[2]);
        load_nodal(packed, M.nplane, nV + nn + 1, cur.c[3]);
        NSDG_STAMP(1);
        stress_update(ul, vl, cur.P, M.ihx, M.ihy, M.ialpha, M.dmin2, cur.s11, cur.s12, cur.s22);
        NSDG_STAMP(2);
        double cx[9], cy[9];
        node_contrib_all(cur.s11, cur.s12, cur.s22, M.hx, M.hy, cx, cy);
        NSDG_STAMP(3);
        const double l2x = from_left(cx[2]), l2y = from_left(cy[2]);
        const double l5x = from_left(cx[5]), l5y = from_left(cy[5]);
        const bool hasB = t > 0;
        // u^p at the 4 owned nodes of row t (kept in registers; Dirichlet nodes are zero)
        if (M.hasL && hasB)
            node_update_packed(M.K, cur.c[0], ul[0], vl[0], ((ca.xl8 + ca.x6) + l2x) + cx[0], ((ca.yl8 + ca.y6) + l2y) + cy[0], 9. * M.iarea,
                cur.u[0], cur.v[0]);
        else
            cur.u[0] = cur.v[0] = 0.;
        if (hasB)
            node_update_packed(M.K, cur.c[1], ul[1], vl[1], ca.x7 + cx[1], ca.y7 + cy[1], 4.5 * M.iarea, cur.u[1], cur.v[1]);
        else
            cur.u[1] = cur.v[1] = 0.;
        if (M.hasL)
            node_update_packed(M.K, cur.c[2], ul[3], vl[3], l5x + cx[3], l5y + cy[3], 4.5 * M.iarea, cur.u[2], cur.v[2]);
        else
            cur.u[2] = cur.v[2] = 0.;
        node_update_packed(M.K, cur.c[3], ul[4], vl[4], cx[4], cy[4], 2.25 * M.iarea, cur.u[3], cur.v[3]);
        ca.x6 = cx[6], ca.y6 = cy[6], ca.x7 = cx[7], ca.y7 = cy[7];
        ca.xl8 = from_left(cx[8]), ca.yl8 = from_left(cy[8]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cur.u[k] = cur.v[k] = 0.; // node row 2*ny is the top boundary
    }

    NSDG_STAMP(4);
    // ---------------------------------------------------------------------- B(t): sub-iteration p+1 on row r = t-1
    const int r = t - 1;
    if (r >= M.tbeg && r >= M.y0 - 1 && r < M.y1) { // wave-uniform
        // u^p at the 9 nodes of element (ix, r): own nodes of rows r and t, the right neighbour's V / EY
        double ul[9], vl[9];
        ul[0] = prev.u[0], vl[0] = prev.v[0];
        ul[1] = prev.u[1], vl[1] = prev.v[1];
        ul[3] = prev.u[2], vl[3] = prev.v[2];
        ul[4] = prev.u[3], vl[4] = prev.v[3];
        ul[6] = cur.u[0], vl[6] = cur.v[0];
        ul[7] = cur.u[1], vl[7] = cur.v[1];
        const double r2u = from_right(prev.u[0]), r2v = from_right(prev.v[0]);
        const double r5u = from_right(prev.u[2]), r5v = from_right(prev.v[2]);
        const double r8u = from_right(cur.u[0]), r8v = from_right(cur.v[0]);
        ul[2] = M.lastcol ? 0. : r2u, vl[2] = M.lastcol ? 0. : r2v; // node column 2*nx is the right boundary
        ul[5] = M.lastcol ? 0. : r5u, vl[5] = M.lastcol ? 0. : r5v;
        ul[8] = M.lastcol ? 0. : r8u, vl[8] = M.lastcol ? 0. : r8v;
        stress_update(ul, vl, prev.P, M.ihx, M.ihy, M.ialpha, M.dmin2, prev.s11, prev.s12, prev.s22);
        NSDG_STAMP(5);
        const bool store = M.own && r >= M.y0;
        if (store) {
            const long ts = tile_off(ix, r, M.ntx, 8);
            tile_store8(S.o11, ts, prev.s11);
            tile_store8(S.o12, ts, prev.s12);
            tile_store8(S.o22, ts, prev.s22);
        }
        double cx[9], cy[9];
        NSDG_STAMP(6);
        node_contrib_all(prev.s11, prev.s12, prev.s22, M.hx, M.hy, cx, cy);
        NSDG_STAMP(7);
        const double l2x = from_left(cx[2]), l2y = from_left(cy[2]);
        const double l5x = from_left(cx[5]), l5y = from_left(cy[5]);
        if (r >= M.y0) { // wave-uniform: rows below y0 only feed the carried contributions
            const bool hasB = r > 0;
            const long nV = (long)(2 * r) * nn + 2 * ix;
            double un, vn;
            if (M.hasL && hasB)
                node_update_packed(M.K, prev.c[0], prev.u[0], prev.v[0], ((cb.xl8 + cb.x6) + l2x) + cx[0], ((cb.yl8 + cb.y6) + l2y) + cy[0],
                    9. * M.iarea, un, vn);
            else
                un = vn = 0.;
            if (store)
                u_new[nV] = un, v_new[nV] = vn;
            if (hasB)
                node_update_packed(M.K, prev.c[1], prev.u[1], prev.v[1], cb.x7 + cx[1], cb.y7 + cy[1], 4.5 * M.iarea, un, vn);
            else
                un = vn = 0.;
            if (store)
                u_new[nV + 1] = un, v_new[nV + 1] = vn;
            if (M.hasL)
                node_update_packed(M.K, prev.c[2], prev.u[2], prev.v[2], l5x + cx[3], l5y + cy[3], 4.5 * M.iarea, un, vn);
            else
                un = vn = 0.;
            if (store)
                u_new[nV + nn] = un, v_new[nV + nn] = vn;
            node_update_packed(M.K, prev.c[3], prev.u[3], prev.v[3], cx[4], cy[4], 2.25 * M.iarea, un, vn);
            if (store) {
                u_new[nV + nn + 1] = un, v_new[nV + nn + 1] = vn;
                if (M.lastcol) {
                    u_new[nV + 2] = 0., v_new[nV + 2] = 0.;
                    u_new[nV + nn + 2] = 0., v_new[nV + nn + 2] = 0.;
                }
                if (r == M.ny - 1) {
                    u_new[nV + 2 * nn] = 0., v_new[nV + 2 * nn] = 0.;
                    u_new[nV + 2 * nn + 1] = 0., v_new[nV + 2 * nn + 1] = 0.;
                    if (M.lastcol)
                        u_new[nV + 2 * nn + 2] = 0., v_new[nV + 2 * nn + 2] = 0.;
                }
            }
        }
        cb.x6 = cx[6], cb.y6 = cy[6], cb.x7 = cx[7], cb.y7 = cy[7];
        cb.xl8 = from_left(cx[8]), cb.yl8 = from_left(cy[8]);
    }
    NSDG_STAMP(8);
}

__global__ __launch_bounds__(256) void mevp_fused2_kernel(NodalConsts K, int nx, int ny, int j0, int j1, int R, int ncw, double hx, double hy,
    double ialpha, double dmin2, StressPtrs2 S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = wave / ncw, cw = wave - strip * ncw;
    MarchConst M;
    M.y0 = j0 + strip * R;
    if (M.y0 >= j1)
        return; // wave-uniform
    M.y1 = min(M.y0 + R, j1);
    const int ixr = cw * 61 - 2 + lane;
    const bool valid = ixr >= 0 && ixr < nx;
    M.K = K;
    M.nx = nx, M.ny = ny;
    M.own = valid && lane >= 2 && lane <= 62;
    M.ix = min(max(ixr, 0), nx - 1);
    M.hasL = M.ix > 0, M.lastcol = M.ix == nx - 1;
    M.ntx = tiles_per_row(nx);
    M.nn = 2 * nx + 1;
    M.nplane = nodal_plane((long)M.nn * (2 * ny + 1));
    M.hx = hx, M.hy = hy, M.ihx = 1. / hx, M.ihy = 1. / hy, M.iarea = M.ihx * M.ihy;
    M.ialpha = ialpha, M.dmin2 = dmin2;
    M.tbeg = max(M.y0 - 2, 0), M.tend = min(M.y1, ny - 1); // A runs on rows tbeg .. tend

    RowCarry X, Y; // alternate between "written by A" and "read by B": no copies when the march advances
#pragma unroll
    for (int k = 0; k < 4; ++k)
        Y.u[k] = Y.v[k] = 0.;
    TopCarry ca, cb; // sub-iteration p (row t-1) and p+1 (row t-2)
#ifdef NSDG_STAMPS
    unsigned stamp_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned stamp_last = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned stamp_t0 = stamp_last, stamp_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime(); // 100 MHz reference
#endif
    for (int t = M.tbeg; t <= M.tend + 1; t += 2) {
        march_step(M, t, X, Y, ca, cb, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
        if (t + 1 <= M.tend + 1)
            march_step(M, t + 1, Y, X, ca, cb, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
    }
#ifdef NSDG_STAMPS
    stamp_acc[9] = M.tend + 2 - M.tbeg; // march steps
    if (lane == 0 && (wave & 63) == 0 && wave / 64 < 64) {
        for (int k = 0; k < 10; ++k)
            nsdg_stamp_acc[(wave / 64) * 16 + k] = stamp_acc[k];
        nsdg_stamp_acc[(wave / 64) * 16 + 10] = (unsigned)__builtin_amdgcn_s_memtime() - stamp_t0; // shader cycles of the march
        nsdg_stamp_acc[(wave / 64) * 16 + 11] = (unsigned)__builtin_amdgcn_s_memrealtime() - stamp_rt0; // the same in 10 ns ticks
    }
#endif
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

#ifdef NSDG_STAMPS
extern "C" int nsdg_debug_read_stamps(unsigned* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nsdg_stamp_acc), sizeof(unsigned) * 64 * 16);
}
#endif

// two sub-iterations on the owned rows [j0, j1) of the local array
int nsdg_launch_mevp_fused2(nsdg_ctx* ctx, int j0, int j1, const double* s11i, const double* s12i, const double* s22i, double* s11, double* s12,
    double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed, const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, 61); // 61 owned columns per wave
    int R = ctx->strip_rows;
    if (R <= 0) {
        // every wave marches R+3 rows of A and R+1 rows of B (~2R+4 stress updates for 2R useful ones);
        // same round-counting rule as in the single-iteration kernel, with one resident wave per SIMD
        const long slots = 1L * 4 * ctx->num_cus;
        double best = 1e30;
        R = 16;
        for (int r = 1; r <= 256; ++r) {
            const long waves = (long)nsdg_div_up(j1 - j0, r) * ncw;
            const long rounds = (waves + slots - 1) / slots;
            const double cost = rounds * (r + 2.0) + (rounds == 1 ? 2.0 : 0.0);
            if (cost < best) {
                best = cost;
                R = r;
            }
        }
    }
    const int nstrips = nsdg_div_up(j1 - j0, R);
    const long nwaves = (long)ncw * nstrips;
    const StressPtrs2 S = { s11i, s12i, s22i, s11, s12, s22 };
    const nsdg_mevp_params& P = ctx->mevp;
    const NodalConsts K = { P.rho_ice * P.beta / ctx->pack_dt, P.rho_ice * (1. + P.beta) / ctx->pack_dt, P.rho_ice * P.fc };
    hipLaunchKernelGGL(mevp_fused2_kernel, dim3(nsdg_div_up(nwaves, 4)), dim3(256), 0, ctx->stream, K, ctx->nx, ctx->ny, j0, j1, R, ncw, ctx->hx,
        ctx->hy, 1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, S, u_old, v_old, packed, pg, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}
