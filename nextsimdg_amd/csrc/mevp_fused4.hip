// mevp_fused4.hip -- variant 4 of the mEVP sub-cycle: FOUR sub-iterations per kernel pass, ONE PIPELINE STAGE PER WAVE.
//
// Variant 3 runs its three stages one after the other in ONE wave: 486 registers (a sixth of its vector instructions
// only move values between the two halves of the register file), 1024 waves to fill the chip and therefore short strips
// that pay the pipeline fill again and again.  Here a workgroup of four waves -- one per SIMD of a CU -- marches through a
// strip of 57 owned columns x R rows, and wave s performs sub-iteration p+s:
//     at march step t wave s works on element row  t - 2 s
// (two rows of skew per stage: wave s needs u^{p+s-1} on the bottom nodes of the row above, which wave s-1 produced one
// step earlier, and everything of its own row, produced two steps earlier).  The hand-over between two waves -- 24 stress
// coefficients and u, v at the 4 owned nodes per lane -- goes through LDS: three rotating slots of 16 KB per hand-over
// (the row being written, the row above and the row being read), 144 KB per workgroup, ONE workgroup barrier per march
// step.  Wave 0 reads stress and velocity from memory, wave 3 writes them; every wave reads the ice strength and the
// packed nodal coefficients of its row itself (waves 1-3: L2 / Infinity Cache hits), one march step AHEAD of their use.
// 256 workgroups fill the chip, so strips are four times taller than variant 3's, and a pass streams the stress once per
// FOUR sub-iterations (776 B per element and pass = 194 B per element and sub-iteration).
//
// Redundancy instead of synchronisation between workgroups, one more level than variant 3: a workgroup owns 57 of its 64
// columns (lanes 0-3 recompute the four columns to its left, lanes 61-63 the three to its right), and a strip of R rows
// runs stage s on rows y0-4+s .. y1+2-s.  The arithmetic is the same sequence of inlined functions as in the other
// variants: one pass of variant 4 and four passes of variant 1 agree to the last bit.
//
// Row ranges: a launch updates the owned element rows [j0, j1) and reads four rows below and three above them.  Where
// those rows do not exist the edge of the local array is the physical boundary.
#include "mevp_pipeline.h"

namespace nsdg_mevp_detail {

struct StressPtrs4 {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

constexpr int F4_OWNED = 57, F4_LEFT = 4; // lanes 4 .. 60 own a column
constexpr int F4_SLOTS = 3; // rotating slots per hand-over
constexpr int F4_HAND = 32; // doubles per lane and slot: 24 stress coefficients + u, v at the 4 owned nodes
constexpr int F4_SLOT = F4_HAND * 64; // doubles per slot
constexpr int F4_LDS = 3 * F4_SLOTS * F4_SLOT; // three hand-overs

// what a wave fetches from memory for one element row, one march step before it is used
struct Fetch4 {
    double P[9]; // ice strength at the Gauss points
    double c[4][6]; // packed momentum coefficients of the 4 owned nodes (V, EX, EY, C)
    double s11[8], s12[8], s22[8]; // stage 0 only: the stress the pass starts from
    double ul[9], vl[9]; // stage 0 only: the velocity the pass starts from at the 9 nodes of the element
};

struct Stage4 {
    int s; // pipeline stage of this wave = sub-iteration p + s
    int first, last; // element rows this stage works on
    int last_prev; // last row of the previous stage (the row above `last` exists unless the strip ends at the physical top)
    int upd0; // node updates from this row on (the first row of a stage only feeds the carried contributions)
};

__device__ __forceinline__ void fetch_row4(const MarchConst3& M, const Stage4& G, int row, Fetch4& f, const StressPtrs4& S,
    const double* __restrict__ u_old, const double* __restrict__ v_old, const double* __restrict__ packed, const double* __restrict__ pg)
{
    const int ix = M.ix, nn = M.nn;
    const long nV = (long)(2 * row) * nn + 2 * ix;
    if (G.s == 0) { // wave-uniform
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const long n = nV + (a / 3) * nn + a % 3;
            f.ul[a] = u_old[n];
            f.vl[a] = v_old[n];
        }
    }
    tile_load9(pg, tile_off(ix, row, M.ntx, 9), ix & 63, f.P);
    if (G.s == 0) {
        const long ts = tile_off(ix, row, M.ntx, 8);
        tile_load8(S.i11, ts, f.s11);
        tile_load8(S.i12, ts, f.s12);
        tile_load8(S.i22, ts, f.s22);
    }
    load_nodal(packed, M.nplane, nV, f.c[0]);
    load_nodal(packed, M.nplane, nV + 1, f.c[1]);
    load_nodal(packed, M.nplane, nV + nn, f.c[2]);
    load_nodal(packed, M.nplane, nV + nn + 1, f.c[3]);
}

// all LDS traffic of the step has landed and every wave of the workgroup has arrived; global loads and stores stay in flight
__device__ __forceinline__ void handover_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One march step of one wave: fetch the inputs of the NEXT row of this stage into `nxt`, work on the current row with the
// inputs `cur` fetched one step ago, meet the other three waves.
__device__ __forceinline__ void march_step4(const MarchConst3& M, const Stage4& G, int t, Fetch4& cur, Fetch4& nxt, TopCarry3& carry,
    double* __restrict__ lds, const StressPtrs4& S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    const int row = t - 2 * G.s;
    if (row + 1 >= G.first && row + 1 <= G.last) // wave-uniform
        fetch_row4(M, G, row + 1, nxt, S, u_old, v_old, packed, pg);
    if (row >= G.first && row <= G.last) { // wave-uniform
        double s11[8], s12[8], s22[8], uu[4], vv[4], ul[9], vl[9];
        if (G.s == 0) {
#pragma unroll
            for (int a = 0; a < 9; ++a)
                ul[a] = cur.ul[a], vl[a] = cur.vl[a];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                s11[i] = cur.s11[i], s12[i] = cur.s12[i], s22[i] = cur.s22[i];
            uu[0] = ul[0], uu[1] = ul[1], uu[2] = ul[3], uu[3] = ul[4];
            vv[0] = vl[0], vv[1] = vl[1], vv[2] = vl[3], vv[3] = vl[4];
        } else {
            // hand-over of the previous stage: its row `row` (two steps old) and the bottom nodes of its row `row + 1` (one step old)
            const double* in = lds + ((G.s - 1) * F4_SLOTS + row % F4_SLOTS) * F4_SLOT + M.lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uu[k] = in[(24 + k) * 64];
                vv[k] = in[(28 + k) * 64];
            }
            double tu0 = 0., tu1 = 0., tv0 = 0., tv1 = 0.; // node row 2*ny is the top boundary
            if (row + 1 <= G.last_prev) {
                const double* top = lds + ((G.s - 1) * F4_SLOTS + (row + 1) % F4_SLOTS) * F4_SLOT + M.lane;
                tu0 = top[24 * 64], tu1 = top[25 * 64], tv0 = top[28 * 64], tv1 = top[29 * 64];
            }
            gather_nodes(M, uu, tu0, tu1, ul);
            gather_nodes(M, vv, tv0, tv1, vl);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s11[i] = in[i * 64];
                s12[i] = in[(8 + i) * 64];
                s22[i] = in[(16 + i) * 64];
            }
        }
        stress_update(ul, vl, cur.P, M.ihx, M.ihy, M.ialpha, M.dmin2, s11, s12, s22);
        double cx[9], cy[9];
        node_contrib_all(s11, s12, s22, M.hx, M.hy, cx, cy);
        double un[4] = { 0., 0., 0., 0. }, vn[4] = { 0., 0., 0., 0. };
        if (row >= G.upd0) // wave-uniform
            owned_node_updates(M, row > 0, cur.c, uu, vv, carry, cx, cy, un, vn);
        carry_top(carry, cx, cy);
        if (G.s < 3) {
            double* out = lds + (G.s * F4_SLOTS + row % F4_SLOTS) * F4_SLOT + M.lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                out[i * 64] = s11[i];
                out[(8 + i) * 64] = s12[i];
                out[(16 + i) * 64] = s22[i];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out[(24 + k) * 64] = un[k];
                out[(28 + k) * 64] = vn[k];
            }
        } else if (M.own && row >= M.y0) { // the last stage runs on rows y0-1 .. y1-1
            const int nn = M.nn;
            const long ts = tile_off(M.ix, row, M.ntx, 8);
            const long nV = (long)(2 * row) * nn + 2 * M.ix;
            tile_store8(S.o11, ts, s11);
            tile_store8(S.o12, ts, s12);
            tile_store8(S.o22, ts, s22);
            u_new[nV] = un[0], v_new[nV] = vn[0];
            u_new[nV + 1] = un[1], v_new[nV + 1] = vn[1];
            u_new[nV + nn] = un[2], v_new[nV + nn] = vn[2];
            u_new[nV + nn + 1] = un[3], v_new[nV + nn + 1] = vn[3];
            if (M.lastcol) {
                u_new[nV + 2] = 0., v_new[nV + 2] = 0.;
                u_new[nV + nn + 2] = 0., v_new[nV + nn + 2] = 0.;
            }
            if (row == M.ny - 1) {
                u_new[nV + 2 * nn] = 0., v_new[nV + 2 * nn] = 0.;
                u_new[nV + 2 * nn + 1] = 0., v_new[nV + 2 * nn + 1] = 0.;
                if (M.lastcol)
                    u_new[nV + 2 * nn + 2] = 0., v_new[nV + 2 * nn + 2] = 0.;
            }
        }
    }
    handover_barrier();
}

// Row ranges as in variant 3: [j0, j1) in strips of R rows and, when nsA > 0 strips are given for it, a SECOND disjoint
// range [j0b, j1b) after them in the same launch (the two bands of rows a block sends to its neighbours).
__global__ __launch_bounds__(256) void mevp_fused4_kernel(NodalConsts K, int nx, int ny, int j0, int j1, int j0b, int j1b, int nsA, int R, int ncw,
    double hx, double hy, double ialpha, double dmin2, StressPtrs4 S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    __shared__ double lds[F4_LDS]; // 144 KB: the three hand-overs of this workgroup
    const int lane = threadIdx.x & 63;
    const int group = xcd_contiguous_block(blockIdx.x, gridDim.x);
    int strip = group / ncw;
    const int cw = group - strip * ncw;
    if (strip >= nsA) { // workgroup-uniform: a strip of the second range
        strip -= nsA;
        j0 = j0b, j1 = j1b;
    }
    MarchConst3 M;
    M.y0 = j0 + strip * R;
    if (M.y0 >= j1)
        return; // workgroup-uniform: no wave of this workgroup reaches a barrier
    M.y1 = min(M.y0 + R, j1);
    const int ixr = cw * F4_OWNED - F4_LEFT + lane;
    const bool valid = ixr >= 0 && ixr < nx;
    M.K = K;
    M.nx = nx, M.ny = ny, M.lane = lane;
    M.own = valid && lane >= F4_LEFT && lane < F4_LEFT + F4_OWNED;
    M.ix = min(max(ixr, 0), nx - 1);
    M.hasL = M.ix > 0, M.lastcol = M.ix == nx - 1;
    M.ntx = tiles_per_row(nx);
    M.nn = 2 * nx + 1;
    M.nplane = nodal_plane((long)M.nn * (2 * ny + 1));
    M.hx = hx, M.hy = hy, M.ihx = 1. / hx, M.ihy = 1. / hy, M.iarea = M.ihx * M.ihy;
    M.ialpha = ialpha, M.dmin2 = dmin2;
    M.tbeg = max(M.y0 - 4, 0);
    M.tendA = M.tendB = 0; // (fields of the single-wave pipeline)

    Stage4 G;
    G.s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    G.first = max(M.y0 - 4 + G.s, 0);
    G.last = min(M.y1 + 2 - G.s, ny - 1);
    G.last_prev = min(M.y1 + 3 - G.s, ny - 1);
    G.upd0 = G.s == 0 ? 0 : M.y0 - 3 + G.s;

    Fetch4 X, Y; // alternate between "being fetched" and "being used": no copies when the march advances
    TopCarry3 carry;
    const int tlast = M.y1 + 5; // stage 3 finishes row y1 - 1 at step y1 - 1 + 6
    if (G.s == 0)
        fetch_row4(M, G, G.first, X, S, u_old, v_old, packed, pg); // G.first == M.tbeg for stage 0
    for (int t = M.tbeg; t <= tlast; t += 2) {
        march_step4(M, G, t, X, Y, carry, lds, S, u_old, v_old, packed, pg, u_new, v_new);
        if (t + 1 <= tlast)
            march_step4(M, G, t + 1, Y, X, carry, lds, S, u_old, v_old, packed, pg, u_new, v_new);
    }
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

// four sub-iterations on the rows [j0, j1) of the local array and, if j0b < j1b, on a second disjoint range [j0b, j1b)
// in the same launch
int nsdg_launch_mevp_fused4_ranges(nsdg_ctx* ctx, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, F4_OWNED);
    const int rowsB = j0b < j1b ? j1b - j0b : 0;
    int R = ctx->strip_rows;
    if (R <= 0) {
        // a strip of R rows takes R + 10 march steps (stage 0 runs on R + 7 rows, stage 3 ends six steps after it);
        // one resident workgroup per CU (LDS)
        const long slots = ctx->num_cus;
        double best = 1e30;
        R = 64;
        for (int r = 1; r <= 4096; ++r) {
            const long groups = ((long)nsdg_div_up(j1 - j0, r) + nsdg_div_up(rowsB, r)) * ncw;
            const long rounds = (groups + slots - 1) / slots;
            const double cost = rounds * (r + 10.0);
            if (cost < best) {
                best = cost;
                R = r;
            }
            if (groups <= ncw * (rowsB ? 2 : 1))
                break; // one strip per range: taller strips change nothing
        }
    }
    const int nsA = nsdg_div_up(j1 - j0, R), nsB = nsdg_div_up(rowsB, R);
    const long ngroups = (long)ncw * (nsA + nsB);
    const StressPtrs4 S = { s11i, s12i, s22i, s11, s12, s22 };
    const nsdg_mevp_params& P = ctx->mevp;
    const NodalConsts K = { P.rho_ice * P.beta / ctx->pack_dt, P.rho_ice * (1. + P.beta) / ctx->pack_dt, P.rho_ice * P.fc };
    hipLaunchKernelGGL(mevp_fused4_kernel, dim3(ngroups), dim3(256), 0, ctx->stream, K, ctx->nx, ctx->ny, j0, j1, j0b, j1b, nsA, R, ncw, ctx->hx,
        ctx->hy, 1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, S, u_old, v_old, packed, pg, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int nsdg_launch_mevp_fused4(nsdg_ctx* ctx, int j0, int j1, const double* s11i, const double* s12i, const double* s22i, double* s11, double* s12,
    double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed, const double* pg)
{
    return nsdg_launch_mevp_fused4_ranges(ctx, j0, j1, 0, 0, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
}
