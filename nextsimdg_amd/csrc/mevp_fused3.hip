// mevp_fused3.hip -- variant 3 of the mEVP sub-cycle: THREE sub-iterations per kernel pass.
//
// The two-iteration kernel (mevp_fused2.hip) streams 776 B per element and pass at 81-90 % of the HBM copy
// ceiling, so a third sub-iteration per pass is the next cut in bytes per sub-iteration (388 -> 259 B).
// The march is pipelined one level deeper: at march step t a lane performs
//     A(t): sub-iteration p   on element row t      (S^p(t),       u^p     at the row's owned nodes)
//     B(t): sub-iteration p+1 on element row t - 1  (S^{p+1}(t-1), u^{p+1})
//     C(t): sub-iteration p+2 on element row t - 2  (S^{p+2}(t-2), u^{p+2} -> memory)
// A -> B hand-over stays in registers exactly as in variant 2 (two alternating RowCarry sets).  The B -> C
// hand-over -- 24 stress coefficients and 8 nodal velocities per lane -- does not fit into the 512
// registers any more and is parked in LDS: B(t) writes slot (t-1)&1, C(t+1) reads it one march step
// later (2 slots x 32 doubles x 64 lanes = 32 KB per wave; one wave per workgroup, four workgroups per
// CU).  C re-reads the ice strength and the packed nodal coefficients of its row from memory (they were
// streamed in by A two steps earlier: L2 / Infinity Cache hits, not HBM traffic).
//
// Redundancy instead of synchronisation, one more level: a wave owns 59 of its 64 columns (lanes 0-2
// recompute the three columns to its left, lanes 62, 63 the two to its right) and a strip of R rows runs
// A on rows y0-3 .. y1+1, B on y0-2 .. y1 and C on y0-1 .. y1-1.  All recomputed values are bit-identical
// to their owners'; the arithmetic is the same sequence of inlined functions as in the other variants, so
// three passes of variant 1 and one pass of variant 3 agree to the last bit.
//
// Row ranges: a launch updates the owned element rows [j0, j1) and reads three rows below and two above
// them.  Where those rows do not exist the edge of the local array is the physical boundary.
#include "mevp_pipeline.h"

#ifdef NSDG_STAMPS
__device__ unsigned nsdg_stamp_acc3[64 * 16];
#endif

namespace nsdg_mevp_detail {

struct StressPtrs3 {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

// everything of one element row that sub-iteration p+1 needs from sub-iteration p (registers)
struct RowCarry3 {
    double s11[8], s12[8], s22[8]; // S^p of the row (relaxed in place to S^{p+1} by B)
    double c[4][6]; // packed momentum coefficients of the 4 owned nodes (V, EX, EY, C)
    double u[4], v[4]; // u^p, v^p at those nodes
};

constexpr int PARK_VALUES = 32; // 24 stress coefficients + u, v at the 4 owned nodes
constexpr int PARK_SLOT = PARK_VALUES * 64; // doubles per slot

// One march step: A(t) into `cur`; B(t) on row t-1 from `prev` (parked in LDS afterwards); C(t) on row t-2 from LDS.
__device__ __forceinline__ void march_step3(const MarchConst3& M, int t, RowCarry3& cur, RowCarry3& prev, TopCarry3& ca, TopCarry3& cb,
    TopCarry3& cc, double* __restrict__ park, const StressPtrs3& S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new NSDG_STAMP_ARGS)
{
    const int nn = M.nn, ix = M.ix;
    NSDG_STAMP(0);
    // ---------------------------------------------------------------------- A(t): sub-iteration p on row t
    if (t <= M.tendA) {
        const long ts = tile_off(ix, t, M.ntx, 8), tp = tile_off(ix, t, M.ntx, 9);
        const long nV = (long)(2 * t) * nn + 2 * ix;
        double ul[9], vl[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const long n = nV + (a / 3) * nn + a % 3;
            ul[a] = u_old[n];
            vl[a] = v_old[n];
        }
        double PA[9];
        tile_load9(pg, tp, ix & 63, PA);
        tile_load8(S.i11, ts, cur.s11);
        tile_load8(S.i12, ts, cur.s12);
        tile_load8(S.i22, ts, cur.s22);
        load_nodal(packed, M.nplane, nV, cur.c[0]);
        load_nodal(packed, M.nplane, nV + 1, cur.c[1]);
        load_nodal(packed, M.nplane, nV + nn, cur.c[2]);
        load_nodal(packed, M.nplane, nV + nn + 1, cur.c[3]);
        NSDG_STAMP(1);
        stress_update(ul, vl, PA, M.ihx, M.ihy, M.ialpha, M.dmin2, cur.s11, cur.s12, cur.s22);
        NSDG_STAMP(2);
        double cx[9], cy[9];
        node_contrib_all(cur.s11, cur.s12, cur.s22, M.hx, M.hy, cx, cy);
        const double uu[4] = { ul[0], ul[1], ul[3], ul[4] }, vv[4] = { vl[0], vl[1], vl[3], vl[4] };
        owned_node_updates(M, t > 0, cur.c, uu, vv, ca, cx, cy, cur.u, cur.v); // u^p, kept in registers
        carry_top(ca, cx, cy);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cur.u[k] = cur.v[k] = 0.; // node row 2*ny is the top boundary
    }

    NSDG_STAMP(3);
    // ---------------------------------------------------------------------- B(t): sub-iteration p+1 on row r = t-1
    const int r = t - 1;
    double bu[4] = { 0., 0., 0., 0. }, bv[4] = { 0., 0., 0., 0. }; // u^{p+1} at the owned nodes of row r (zero above the top boundary)
    if (r >= M.tbeg && r >= M.y0 - 2 && r <= M.tendB) { // wave-uniform
        // the ice strength of the row is read again (an L2 hit: A streamed it in one step ago) rather than
        // carried: 18 registers per set that would push the kernel over the 512-register file
        double PB[9];
        tile_load9(pg, tile_off(ix, r, M.ntx, 9), ix & 63, PB);
        double ul[9], vl[9];
        gather_nodes(M, prev.u, cur.u[0], cur.u[1], ul);
        gather_nodes(M, prev.v, cur.v[0], cur.v[1], vl);
        NSDG_STAMP(4);
        stress_update(ul, vl, PB, M.ihx, M.ihy, M.ialpha, M.dmin2, prev.s11, prev.s12, prev.s22);
        NSDG_STAMP(5);
        double cx[9], cy[9];
        node_contrib_all(prev.s11, prev.s12, prev.s22, M.hx, M.hy, cx, cy);
        if (r >= M.y0 - 1) // wave-uniform: the row below only feeds the carried contributions
            owned_node_updates(M, r > 0, prev.c, prev.u, prev.v, cb, cx, cy, bu, bv);
        carry_top(cb, cx, cy);
        // park S^{p+1}(r) and u^{p+1}(r) for C(t+1)
        double* slot = park + (r & 1) * PARK_SLOT + M.lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            slot[i * 64] = prev.s11[i];
            slot[(8 + i) * 64] = prev.s12[i];
            slot[(16 + i) * 64] = prev.s22[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            slot[(24 + k) * 64] = bu[k];
            slot[(28 + k) * 64] = bv[k];
        }
    }

    NSDG_STAMP(6);
    // ---------------------------------------------------------------------- C(t): sub-iteration p+2 on row q = t-2
    const int q = t - 2;
    if (q >= M.tbeg && q >= M.y0 - 1 && q < M.y1) { // wave-uniform
        const double* slot = park + (q & 1) * PARK_SLOT + M.lane;
        double s11[8], s12[8], s22[8], qu[4], qv[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s11[i] = slot[i * 64];
            s12[i] = slot[(8 + i) * 64];
            s22[i] = slot[(16 + i) * 64];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qu[k] = slot[(24 + k) * 64];
            qv[k] = slot[(28 + k) * 64];
        }
        const long tp = tile_off(ix, q, M.ntx, 9);
        const long nV = (long)(2 * q) * nn + 2 * ix;
        double P[9], c[4][6];
        tile_load9(pg, tp, ix & 63, P);
        load_nodal(packed, M.nplane, nV, c[0]);
        load_nodal(packed, M.nplane, nV + 1, c[1]);
        load_nodal(packed, M.nplane, nV + nn, c[2]);
        load_nodal(packed, M.nplane, nV + nn + 1, c[3]);
        double ul[9], vl[9];
        gather_nodes(M, qu, bu[0], bu[1], ul);
        gather_nodes(M, qv, bv[0], bv[1], vl);
        NSDG_STAMP(7);
        stress_update(ul, vl, P, M.ihx, M.ihy, M.ialpha, M.dmin2, s11, s12, s22);
        NSDG_STAMP(8);
        const bool store = M.own && q >= M.y0;
        if (store) {
            const long ts = tile_off(ix, q, M.ntx, 8);
            tile_store8(S.o11, ts, s11);
            tile_store8(S.o12, ts, s12);
            tile_store8(S.o22, ts, s22);
        }
        double cx[9], cy[9];
        NSDG_STAMP(9);
        node_contrib_all(s11, s12, s22, M.hx, M.hy, cx, cy);
        if (q >= M.y0) { // wave-uniform
            double un[4], vn[4];
            owned_node_updates(M, q > 0, c, qu, qv, cc, cx, cy, un, vn);
            if (store) {
                u_new[nV] = un[0], v_new[nV] = vn[0];
                u_new[nV + 1] = un[1], v_new[nV + 1] = vn[1];
                u_new[nV + nn] = un[2], v_new[nV + nn] = vn[2];
                u_new[nV + nn + 1] = un[3], v_new[nV + nn + 1] = vn[3];
                if (M.lastcol) {
                    u_new[nV + 2] = 0., v_new[nV + 2] = 0.;
                    u_new[nV + nn + 2] = 0., v_new[nV + nn + 2] = 0.;
                }
                if (q == M.ny - 1) {
                    u_new[nV + 2 * nn] = 0., v_new[nV + 2 * nn] = 0.;
                    u_new[nV + 2 * nn + 1] = 0., v_new[nV + 2 * nn + 1] = 0.;
                    if (M.lastcol)
                        u_new[nV + 2 * nn + 2] = 0., v_new[nV + 2 * nn + 2] = 0.;
                }
            }
        }
        carry_top(cc, cx, cy);
    }
    NSDG_STAMP(10);
}

// Row ranges: [j0, j1) in strips of R rows and, when nsA > 0 strips are given for it, a SECOND disjoint range
// [j0b, j1b) after them in the same launch (the two bands of rows a block sends to its neighbours: as one launch they
// share the resident wave slots instead of paying two pipeline fills in a row).
__global__ __launch_bounds__(64) void mevp_fused3_kernel(NodalConsts K, int nx, int ny, int j0, int j1, int j0b, int j1b, int nsA, int R, int ncw, double hx, double hy,
    double ialpha, double dmin2, StressPtrs3 S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    __shared__ double park[2 * PARK_SLOT]; // 32 KB: the B -> C hand-over of this wave
    const int lane = threadIdx.x;
    const int wave = xcd_contiguous_block(blockIdx.x, gridDim.x); // one wave per workgroup
    int strip = wave / ncw;
    const int cw = wave - strip * ncw;
    if (strip >= nsA) { // wave-uniform: a strip of the second range
        strip -= nsA;
        j0 = j0b, j1 = j1b;
    }
    MarchConst3 M;
    M.y0 = j0 + strip * R;
    if (M.y0 >= j1)
        return; // wave-uniform
    M.y1 = min(M.y0 + R, j1);
    const int ixr = cw * 59 - 3 + lane;
    const bool valid = ixr >= 0 && ixr < nx;
    M.K = K;
    M.nx = nx, M.ny = ny, M.lane = lane;
    M.own = valid && lane >= 3 && lane <= 61;
    M.ix = min(max(ixr, 0), nx - 1);
    M.hasL = M.ix > 0, M.lastcol = M.ix == nx - 1;
    M.ntx = tiles_per_row(nx);
    M.nn = 2 * nx + 1;
    M.nplane = nodal_plane((long)M.nn * (2 * ny + 1));
    M.hx = hx, M.hy = hy, M.ihx = 1. / hx, M.ihy = 1. / hy, M.iarea = M.ihx * M.ihy;
    M.ialpha = ialpha, M.dmin2 = dmin2;
    M.tbeg = max(M.y0 - 3, 0);
    M.tendA = min(M.y1 + 1, ny - 1), M.tendB = min(M.y1, ny - 1); // A runs on rows tbeg .. tendA, B up to tendB

    RowCarry3 X, Y; // alternate between "written by A" and "read by B": no copies when the march advances
#pragma unroll
    for (int k = 0; k < 4; ++k)
        Y.u[k] = Y.v[k] = 0.;
    TopCarry3 ca, cb, cc; // sub-iterations p (row t-1), p+1 (row t-2), p+2 (row t-3)
    const int tlast = M.y1 + 1; // C(tlast) finishes row y1 - 1
#ifdef NSDG_STAMPS
    unsigned stamp_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned stamp_last = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned stamp_t0 = stamp_last, stamp_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime(); // 100 MHz reference
#endif
    for (int t = M.tbeg; t <= tlast; t += 2) {
        march_step3(M, t, X, Y, ca, cb, cc, park, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
        if (t + 1 <= tlast)
            march_step3(M, t + 1, Y, X, ca, cb, cc, park, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
    }
#ifdef NSDG_STAMPS
    if (lane == 0 && (wave & 63) == 0 && wave / 64 < 64) {
        for (int k = 0; k < 11; ++k)
            nsdg_stamp_acc3[(wave / 64) * 16 + k] = stamp_acc[k];
        nsdg_stamp_acc3[(wave / 64) * 16 + 11] = tlast + 1 - M.tbeg; // march steps
        nsdg_stamp_acc3[(wave / 64) * 16 + 12] = (unsigned)__builtin_amdgcn_s_memtime() - stamp_t0;
        nsdg_stamp_acc3[(wave / 64) * 16 + 13] = (unsigned)__builtin_amdgcn_s_memrealtime() - stamp_rt0;
    }
#endif
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

#ifdef NSDG_STAMPS
extern "C" int nsdg_debug_read_stamps3(unsigned* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nsdg_stamp_acc3), sizeof(unsigned) * 64 * 16);
}
#endif

// three sub-iterations on the rows [j0, j1) of the local array and, if j0b < j1b, on a second disjoint range [j0b, j1b)
// in the same launch
int nsdg_launch_mevp_fused3_ranges(nsdg_ctx* ctx, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, 59); // 59 owned columns per wave
    const int rowsB = j0b < j1b ? j1b - j0b : 0;
    int R = ctx->strip_rows;
    if (R <= 0) {
        // every wave marches R+5 rows of A, R+3 of B and R+1 of C: R+5 march steps for R useful rows;
        // one resident wave per SIMD (register file) and four 32 KB LDS allocations per CU
        const long slots = 1L * 4 * ctx->num_cus;
        double best = 1e30;
        R = 16;
        for (int r = 1; r <= 256; ++r) {
            const long waves = ((long)nsdg_div_up(j1 - j0, r) + nsdg_div_up(rowsB, r)) * ncw;
            const long rounds = (waves + slots - 1) / slots;
            const double cost = rounds * (r + 5.0) + (rounds == 1 ? 2.0 : 0.0);
            if (cost < best) {
                best = cost;
                R = r;
            }
        }
    }
    const int nsA = nsdg_div_up(j1 - j0, R), nsB = nsdg_div_up(rowsB, R);
    const long nwaves = (long)ncw * (nsA + nsB);
    const StressPtrs3 S = { s11i, s12i, s22i, s11, s12, s22 };
    const nsdg_mevp_params& P = ctx->mevp;
    const NodalConsts K = { P.rho_ice * P.beta / ctx->pack_dt, P.rho_ice * (1. + P.beta) / ctx->pack_dt, P.rho_ice * P.fc };
    hipLaunchKernelGGL(mevp_fused3_kernel, dim3(nwaves), dim3(64), 0, ctx->stream, K, ctx->nx, ctx->ny, j0, j1, j0b, j1b, nsA, R, ncw, ctx->hx,
        ctx->hy, 1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, S, u_old, v_old, packed, pg, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_launch_mevp_fused3(nsdg_ctx* ctx, int j0, int j1, const double* s11i, const double* s12i, const double* s22i, double* s11, double* s12,
    double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed, const double* pg)
{
    return nsdg_launch_mevp_fused3_ranges(ctx, j0, j1, 0, 0, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
}
