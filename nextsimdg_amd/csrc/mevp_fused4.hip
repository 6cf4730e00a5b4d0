// mevp_fused4.hip -- variant 4 of the mEVP sub-cycle: FOUR sub-iterations per kernel pass, ONE PIPELINE STAGE PER WAVE.
//
// Variant 3 runs its three stages one after the other in ONE wave: 486 registers (a sixth of its vector instructions
// only move values between the two halves of the register file), 1024 waves to fill the chip and therefore short strips
// that pay the pipeline fill again and again.  Here a workgroup of four waves -- one per SIMD of a CU -- marches through a
// strip of 57 owned columns x R rows, and wave s performs sub-iteration p+s on the element row  t - 3 s  at march step t.
// The hand-over between two waves -- 24 stress coefficients and u, v at the 4 owned nodes per lane -- goes through LDS:
// three rotating slots of 16 KB per hand-over, 144 KB per workgroup, 16-byte accesses.  Wave 0 reads stress and velocity
// from memory, wave 3 writes them; every wave reads the ice strength and the packed nodal coefficients of its row itself
// (waves 1-3: L2 / Infinity Cache hits).  256 workgroups fill the chip, so strips are four times taller than variant 3's,
// and a pass streams the stress once per FOUR sub-iterations (776 B per element and pass = 194 B per element and
// sub-iteration).
//
// Synchronisation: ONE workgroup barrier per march step and wave -- but every wave meets it at a DIFFERENT point of its
// step.  A step is four quarters (Q0 velocity inputs: LDS reads or the values fetched from memory, node gather; Q1 projected
// stress, then the stress of the row -- read from LDS only here, after the register peak -- and its relaxation; Q2 nodal
// contributions and node updates; Q3 outputs: LDS writes or global stores); wave s calls the barrier after quarter s.  The
// waves therefore run a quarter of a step apart: while one issues its loads the others compute, instead of all four hitting
// the vector-memory pipeline and the LDS of the CU at the same time (the aligned first version spent half of every step in
// those collisions: profiles/r04_fused4_development.md).  Why this is race-free with three slots (barrier #i = the barrier
// call inside step i of each wave; the LDS reads of a step are in Q0 and Q1, never after the wave's own barrier call --
// wave s >= 1 reads, its barrier comes after quarter s >= 1 -- all LDS writes are in Q3, and the barrier waits for the wave's
// own LDS traffic first):
//   * visibility: stage s reads in step i the rows  rho = i - 3 s  and  rho + 1  of stage s-1.  Row rho + 1 was written in Q3
//     of step i - 2 of that wave, i.e. after its barrier #(i-2) and before its barrier #(i-1) (its barrier comes before Q3:
//     s-1 <= 2); the consumer's reads of step i come after its barrier #(i-1).
//   * slot reuse: the producer overwrites the slot of row rho with row rho + 3 in Q3 of ITS step i, after its barrier #i; the
//     consumer's reads of row rho (Q0 / Q1 of step i, and as the row above in step i - 1) are complete before its barrier #i.
// Inputs from memory are requested one row ahead, each group right after the values of the current row have been
// consumed (the stress after the relaxation, the nodal coefficients after the node updates, ...): ONE register set is
// always either waiting to be used or in flight, instead of two alternating sets.
//
// Redundancy instead of synchronisation between workgroups, one more level than variant 3: a workgroup owns 57 of its 64
// columns (lanes 0-3 recompute the four columns to its left, lanes 61-63 the three to its right), and a strip of R rows
// runs stage s on rows y0-4+s .. y1+2-s.  The arithmetic is the same sequence of inlined functions as in the other
// variants: one pass of variant 4 and four passes of variant 1 agree to the last bit.
//
// Row ranges: a launch updates the owned element rows [j0, j1) and reads four rows below and three above them.  Where
// those rows do not exist the edge of the local array is the physical boundary.
#include "mevp_pipeline.h"

#ifdef NSDG_STAMPS
__device__ unsigned nsdg_stamp_acc4[64 * 16]; // 16 sampled workgroups x 4 stages x 16 values
#endif

namespace nsdg_mevp_detail {

struct StressPtrs4 {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

constexpr int F4_OWNED = 57, F4_LEFT = 4; // lanes 4 .. 60 own a column
constexpr int F4_SLOTS = 3; // rotating slots per hand-over
constexpr int F4_HAND = 32; // doubles per lane and slot: 24 stress coefficients + u, v at the 4 owned nodes
constexpr int F4_SLOT = F4_HAND * 64; // doubles per slot; value k of lane l at (k / 2) * 128 + 2 l + k % 2 (16-byte pairs)
constexpr int F4_LDS = 3 * F4_SLOTS * F4_SLOT; // three hand-overs
constexpr int F4_LAG_TOTAL = 9; // stage 3 works on row t - 9: every link is three steps deep
constexpr int F4_STEPS_EXTRA = 7 + F4_LAG_TOTAL; // a strip of R rows takes R + 16 march steps

// what a wave holds from memory for the row it works on next (requested right after the current row's values were used)
struct Fetch4 {
    double P[9]; // ice strength at the Gauss points
    double c[4][6]; // packed momentum coefficients of the 4 owned nodes (V, EX, EY, C)
    double s11[8], s12[8], s22[8]; // stage 0 only: the stress the pass starts from
    double ub[3], vb[3], um[3], vm[3], ut[3], vt[3]; // stage 0 only: u, v of the pass's start on the three node rows of the element row
};

struct Stage4 {
    int s; // pipeline stage of this wave = sub-iteration p + s
    int lag; // this stage works on row t - lag at march step t (3 s)
    int bar; // the quarter after which this wave meets the workgroup barrier (s)
    int first, last; // element rows this stage works on
    int last_prev; // last row of the previous stage (the row above `last` exists unless the strip ends at the physical top)
    int upd0; // node updates from this row on (the first row of a stage only feeds the carried contributions)
};

// all LDS traffic of this wave has landed and every wave of the workgroup has arrived; global loads and stores stay in flight
__device__ __forceinline__ void handover_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double2 lds_pair(const double* slot, int k) { return *reinterpret_cast<const double2*>(slot + k * 128); }
__device__ __forceinline__ void lds_pair(double* slot, int k, double a, double b) { *reinterpret_cast<double2*>(slot + k * 128) = make_double2(a, b); }

// the three nodes n, n+1, n+2 of one node row: the first two as ONE 16-byte access (node rows start at odd multiples of
// 8 bytes on odd rows: global memory instructions only need 4-byte alignment), the third as an 8-byte access -- 8
// instead of 12 vector-memory instructions per element row, and they are what bounds the loader wave
typedef double nsdg_pair8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void fetch_nodes4(const double* __restrict__ w, long n, double (&o)[3])
{
    const nsdg_pair8 a = *reinterpret_cast<const nsdg_pair8*>(w + n);
    o[0] = a.x, o[1] = a.y, o[2] = w[n + 2];
}

// One march step of one wave.  FIRST: the loader wave (stage 0, inputs from memory; a code path of its own so that its
// register allocation and the compiler's bookkeeping of outstanding loads are not entangled with the other stages').
// The step is straight-line code: a wave whose stage has not reached its first row yet, or is past its last, works on
// that row again and again with whatever the hand-over slots hold and only SUPPRESSES ITS OUTPUTS -- branches around
// the loads made the compiler wait for loads it had just issued (profiles/r04_fused4_development.md).
// the inputs of element row `nrow` that every stage reads from memory itself: ice strength and nodal coefficients
__device__ __forceinline__ void request_P4(const MarchConst3& M, int nrow, Fetch4& f, const double* __restrict__ pg)
{
    tile_load9(pg, tile_off(M.ix, nrow, M.ntx, 9), M.ix & 63, f.P);
}
__device__ __forceinline__ void request_c4(const MarchConst3& M, int nrow, double (&c)[4][6], const double* __restrict__ packed)
{
    const long nVn = (long)(2 * nrow) * M.nn + 2 * M.ix;
    load_nodal(packed, M.nplane, nVn, c[0]);
    load_nodal(packed, M.nplane, nVn + 1, c[1]);
    load_nodal(packed, M.nplane, nVn + M.nn, c[2]);
    load_nodal(packed, M.nplane, nVn + M.nn + 1, c[3]);
}

// One march step of one wave.  FIRST: the loader wave (stage 0, inputs from memory; a code path of its own so that its
// register allocation and the compiler's bookkeeping of outstanding loads are not entangled with the other stages').
// A busy step is straight-line code.  A stage that has not reached its first row yet, or is past its last, takes the IDLE
// step: it meets the barrier and touches nothing -- in particular not the prefetch set, so that no value of it is merged
// between the two paths (at the power cap idle arithmetic costs the busy waves their clock; with loads in the idle path the
// register allocator moved ~100 values per step between the two halves of the register file at the joins).  The loader has no idle step: it works on its first / last row again and only suppresses
// the outputs -- with its arithmetic behind a branch the compiler copied freshly loaded values between registers at the join
// and waited for them, a full HBM round trip (profiles/r04_fused4_development.md).
template <bool FIRST>
__device__ __forceinline__ void march_step4(const MarchConst3& M, const Stage4& G, int t, Fetch4& f, TopCarry3& carry, double* __restrict__ lds,
    const StressPtrs4& S, const double* __restrict__ u_old, const double* __restrict__ v_old, const double* __restrict__ packed,
    const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new NSDG_STAMP_ARGS)
{
    const int stage = FIRST ? 0 : G.s, bar = FIRST ? 0 : G.bar;
    const int rowraw = t - (FIRST ? 0 : G.lag);
    const bool active = rowraw >= G.first && rowraw <= G.last; // wave-uniform
    const int row = min(max(rowraw, G.first), G.last);
    const int nrow = min(max(rowraw + 1, G.first), G.last); // the row this stage works on in the NEXT step: its inputs are requested during this one
    // TIMING-ONLY builds (tools/ab_build.sh NAME -DNSDG_F4_TIMING=bits; wrong results: bench.py --no-guard; never defined in the
    // product): what the re-reads of ice strength (P) and nodal coefficients (c) by the stages 1-3 cost.  bit 0: they read P of a
    // FIXED row (same instructions, L1 / L2 hits, no fabric traffic); bit 1: the same for c; bit 2: P is not re-read at all;
    // bit 3: c is not re-read at all
#ifdef NSDG_F4_TIMING
    const int prow = (!FIRST && (NSDG_F4_TIMING & 1)) ? G.first : nrow, crow = (!FIRST && (NSDG_F4_TIMING & 2)) ? G.first : nrow;
    constexpr bool skipP = !FIRST && (NSDG_F4_TIMING & 4), skipC = !FIRST && (NSDG_F4_TIMING & 8);
#else
    const int prow = nrow, crow = nrow;
    constexpr bool skipP = false, skipC = false;
#endif
    const int ix = M.ix, nn = M.nn;
    NSDG_STAMP(0);
    if (!FIRST && !active) { // ------------------------------------------------------------------- idle step: meet the barrier, touch nothing
        handover_barrier();
        NSDG_STAMP(8);
        return;
    }
    if (!FIRST && row == G.first) { // wave-uniform: the first row of the stage has no predecessor that requested its inputs
        request_P4(M, row, f, pg);
        request_c4(M, row, f.c, packed);
    }
    const long nVn = (long)(2 * nrow) * nn + 2 * ix; // vertex node of the next row
    double s11[8], s12[8], s22[8], uu[4], vv[4], ul[9], vl[9], un[4], vn[4];
    // ------------------------------------------------------------------------------------------ Q0: inputs of the row
    if (FIRST) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ul[a] = f.ub[a], ul[3 + a] = f.um[a], ul[6 + a] = f.ut[a];
            vl[a] = f.vb[a], vl[3 + a] = f.vm[a], vl[6 + a] = f.vt[a];
        }
        uu[0] = ul[0], uu[1] = ul[1], uu[2] = ul[3], uu[3] = ul[4];
        vv[0] = vl[0], vv[1] = vl[1], vv[2] = vl[3], vv[3] = vl[4];
    } else {
        // hand-over of the previous stage: its row `row` and the bottom nodes of its row `row + 1`
        const double* in = lds + ((stage - 1) * F4_SLOTS + row % F4_SLOTS) * F4_SLOT + 2 * M.lane;
        const double* top = lds + ((stage - 1) * F4_SLOTS + (row + 1) % F4_SLOTS) * F4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double2 a = lds_pair(in, 12 + k), b = lds_pair(in, 14 + k);
            uu[2 * k] = a.x, uu[2 * k + 1] = a.y, vv[2 * k] = b.x, vv[2 * k + 1] = b.y;
        }
        double2 tu = lds_pair(top, 12), tv = lds_pair(top, 14);
        if (row + 1 > G.last_prev) // wave-uniform: node row 2*ny is the top boundary
            tu = tv = make_double2(0., 0.);
        gather_nodes(M, uu, tu.x, tu.y, ul);
        gather_nodes(M, vv, tv.x, tv.y, vl);
    }
    NSDG_STAMP(1);
    if (bar == 0)
        handover_barrier();
    NSDG_STAMP(2);
    // ------------------------------------------------------------------------------------------ Q1: stress update
    double r11[8], r12[8], r22[8];
    stress_projected(ul, vl, f.P, M.ihx, M.ihy, M.ialpha, M.dmin2, r11, r12, r22);
    __builtin_amdgcn_sched_barrier(0);
    NSDG_STAMP(3);
    if (!skipP)
        request_P4(M, prow, f, pg); // P, and in stage 0 u, v, of the next row
    if (FIRST) {
        if (nrow > row) { // wave-uniform: the top node row of this element row is the bottom one of the next
#pragma unroll
            for (int a = 0; a < 3; ++a)
                f.ub[a] = f.ut[a], f.vb[a] = f.vt[a];
        }
        fetch_nodes4(u_old, nVn + nn, f.um);
        fetch_nodes4(v_old, nVn + nn, f.vm);
        fetch_nodes4(u_old, nVn + 2 * nn, f.ut);
        fetch_nodes4(v_old, nVn + 2 * nn, f.vt);
    }
    NSDG_STAMP(9);
    if (FIRST) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s11[i] = f.s11[i], s12[i] = f.s12[i], s22[i] = f.s22[i];
    }
    else {
        const double* in = lds + ((stage - 1) * F4_SLOTS + row % F4_SLOTS) * F4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 a = lds_pair(in, k), b = lds_pair(in, 4 + k), c = lds_pair(in, 8 + k);
            s11[2 * k] = a.x, s11[2 * k + 1] = a.y, s12[2 * k] = b.x, s12[2 * k + 1] = b.y, s22[2 * k] = c.x, s22[2 * k + 1] = c.y;
        }
    }
    stress_relax(M.ialpha, r11, r12, r22, s11, s12, s22);
    __builtin_amdgcn_sched_barrier(0);
    if (FIRST) { // stress of the next row
        const long ts = tile_off(ix, nrow, M.ntx, 8);
        tile_load8(S.i11, ts, f.s11);
        tile_load8(S.i12, ts, f.s12);
        tile_load8(S.i22, ts, f.s22);
    }
    NSDG_STAMP(10);
    if (!FIRST && bar == 1)
        handover_barrier();
    NSDG_STAMP(4);
    // ------------------------------------------------------------------------------------------ Q2: contributions, node updates
    {
        double cx[9], cy[9];
        node_contrib_all(s11, s12, s22, M.hx, M.hy, cx, cy);
        owned_node_updates(M, row > 0, f.c, uu, vv, carry, cx, cy, un, vn);
        if (row < G.upd0) { // wave-uniform: the first row of a stage only feeds the carried contributions
#pragma unroll
            for (int k = 0; k < 4; ++k)
                un[k] = vn[k] = 0.;
        }
        carry_top(carry, cx, cy);
    }
    __builtin_amdgcn_sched_barrier(0);
    NSDG_STAMP(5);
    // nodal coefficients of the next row.  (Requested for the CURRENT row after the register peak of the projected stress
    // instead -- 48 registers fewer across the step in the stages 1-3, whose coefficients come from L2 / the Infinity Cache --
    // measured slower twice, alternating runs on one box: 1.000-1.002 against 0.994-0.998 ms per pass in the first structure,
    // 1.005-1.008 against 0.962-0.964 in the final one: the L2 / Infinity-Cache latency is then exposed.)
    if (!skipC)
        request_c4(M, crow, f.c, packed);
    NSDG_STAMP(11);
    if (!FIRST && bar == 2)
        handover_barrier();
    NSDG_STAMP(6);
    // ------------------------------------------------------------------------------------------ Q3: outputs
    if (FIRST && !active) {
        // the loader wave past its last row: nothing to hand over
    } else if (FIRST || stage < 3) {
        double* out = lds + (stage * F4_SLOTS + row % F4_SLOTS) * F4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lds_pair(out, k, s11[2 * k], s11[2 * k + 1]);
            lds_pair(out, 4 + k, s12[2 * k], s12[2 * k + 1]);
            lds_pair(out, 8 + k, s22[2 * k], s22[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lds_pair(out, 12 + k, un[2 * k], un[2 * k + 1]);
            lds_pair(out, 14 + k, vn[2 * k], vn[2 * k + 1]);
        }
    } else if (M.own && row >= M.y0) { // the last stage runs on rows y0-1 .. y1-1
        const long ts = tile_off(ix, row, M.ntx, 8);
        const long nV = (long)(2 * row) * nn + 2 * ix;
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
    NSDG_STAMP(7);
    if (bar == 3)
        handover_barrier();
    NSDG_STAMP(8);
}

// Row ranges as in variant 3: [j0, j1) in strips of R rows and, when nsA > 0 strips are given for it, a SECOND disjoint
// range [j0b, j1b) after them in the same launch (the two bands of rows a block sends to its neighbours).
__global__ __launch_bounds__(256) void mevp_fused4_kernel(NodalConsts K, int nx, int ny, int j0, int j1, int j0b, int j1b, int nsA, int R, int ncw,
    double hx, double hy, double ialpha, double dmin2, StressPtrs4 S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    __shared__ __attribute__((aligned(16))) double lds[F4_LDS]; // 144 KB: the three hand-overs of this workgroup
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
    G.lag = 3 * G.s;
    G.bar = G.s;
    G.first = max(M.y0 - 4 + G.s, 0);
    G.last = min(M.y1 + 2 - G.s, ny - 1);
    G.last_prev = min(M.y1 + 3 - G.s, ny - 1);
    G.upd0 = G.s == 0 ? 0 : M.y0 - 3 + G.s;

    Fetch4 f;
    TopCarry3 carry;
    const int tlast = M.y1 - 1 + F4_LAG_TOTAL; // stage 3 finishes row y1 - 1
    { // the inputs of the first row of the stage; every other row is requested by the row before it
        const int row = G.first;
        const long nV = (long)(2 * row) * M.nn + 2 * M.ix, ts = tile_off(M.ix, row, M.ntx, 8);
        if (G.s == 0) {
            fetch_nodes4(u_old, nV, f.ub);
            fetch_nodes4(v_old, nV, f.vb);
            fetch_nodes4(u_old, nV + M.nn, f.um);
            fetch_nodes4(v_old, nV + M.nn, f.vm);
            fetch_nodes4(u_old, nV + 2 * M.nn, f.ut);
            fetch_nodes4(v_old, nV + 2 * M.nn, f.vt);
            tile_load8(S.i11, ts, f.s11);
            tile_load8(S.i12, ts, f.s12);
            tile_load8(S.i22, ts, f.s22);
            tile_load9(pg, tile_off(M.ix, row, M.ntx, 9), M.ix & 63, f.P);
            request_c4(M, row, f.c, packed);
        }
    }
#ifdef NSDG_STAMPS
    unsigned stamp_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned stamp_last = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned stamp_t0 = stamp_last, stamp_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime(); // 100 MHz reference
#endif
    if (G.s == 0) {
        for (int t = M.tbeg; t <= tlast; ++t)
            march_step4<true>(M, G, t, f, carry, lds, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
    } else {
        for (int t = M.tbeg; t <= tlast; ++t)
            march_step4<false>(M, G, t, f, carry, lds, S, u_old, v_old, packed, pg, u_new, v_new NSDG_STAMP_PASS);
    }
#ifdef NSDG_STAMPS
    if (lane == 0 && (group & 15) == 0 && group / 16 < 16) {
        unsigned* o = nsdg_stamp_acc4 + ((group / 16) * 4 + G.s) * 16;
        for (int k = 0; k < 12; ++k)
            o[k] = stamp_acc[k];
        o[12] = tlast + 1 - M.tbeg; // march steps
        o[13] = (unsigned)__builtin_amdgcn_s_memtime() - stamp_t0;
        o[14] = (unsigned)__builtin_amdgcn_s_memrealtime() - stamp_rt0;
    }
#endif
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

#ifdef NSDG_STAMPS
extern "C" int nsdg_debug_read_stamps4(unsigned* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nsdg_stamp_acc4), sizeof(unsigned) * 64 * 16);
}
#endif

// four sub-iterations on the rows [j0, j1) of the local array and, if j0b < j1b, on a second disjoint range [j0b, j1b)
// in the same launch
// mevp_fused4p.hip: the same pass with the hand-over point to point
int nsdg_launch_mevp_fused4p_ranges(nsdg_ctx* ctx, int nst, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg);

int nsdg_launch_mevp_fused4_ranges(nsdg_ctx* ctx, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg)
{
    if (ctx->f4_p2p)
        return nsdg_launch_mevp_fused4p_ranges(ctx, 4, j0, j1, j0b, j1b, s11i, s12i, s22i, s11, s12, s22, u_old, v_old, u_new, v_new, packed, pg);
    const int ncw = nsdg_div_up(ctx->nx, F4_OWNED);
    const int rowsB = j0b < j1b ? j1b - j0b : 0;
    int R = ctx->strip_rows;
    if (R <= 0) {
        // a strip of R rows takes R + 16 march steps (stage 0 runs on R + 7 rows, stage 3 ends eight steps after it);
        // one resident workgroup per CU (LDS)
        const long slots = ctx->num_cus;
        double best = 1e30;
        R = 64;
        for (int r = 1; r <= 4096; ++r) {
            const long groups = ((long)nsdg_div_up(j1 - j0, r) + nsdg_div_up(rowsB, r)) * ncw;
            const long rounds = (groups + slots - 1) / slots;
            const double cost = rounds * (r + (double)F4_STEPS_EXTRA);
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
