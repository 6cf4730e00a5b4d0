// mevp_fused4.hip -- the stage-per-wave pipeline of the mEVP sub-cycle: up to FOUR sub-iterations per kernel pass, one pipeline stage
// per wave of a four-wave workgroup, hand-over POINT TO POINT (round 5).  A workgroup marches through a strip of 57 owned columns x
// R rows; wave s performs sub-iteration p + s; the hand-over between two waves -- 24 stress coefficients and u, v at the 4 owned
// nodes per lane -- goes through LDS.  Redundancy instead of synchronisation BETWEEN workgroups: lanes 0-3 / 61-63 recompute the
// columns beside the owned ones, stage s of a pass of n sub-iterations runs on the rows y0 - n + s .. y1 + n - 2 - s.
//
// The round-4 kernel (git history: csrc/mevp_fused4.hip before round 5) synchronised its four stage waves with ONE workgroup barrier per march step.  A barrier is
// a rendez-vous of all four waves, so a link of the pipeline can only be as short as "write, barrier, read": every link was three
// march steps deep, a strip of R rows took R + 16 steps, a row's ice strength and nodal coefficients were in flight for ten steps
// -- re-read from the Infinity Cache by each of the stages 1-3 (3.4 of the 6.7 GB a pass moves, -13.5 % without them) -- and the
// three rotating slots per link filled the LDS (144 of 160 KB).  Here a stage waits for exactly what it needs and nothing else:
//
//   * every link has two counters in LDS.  done[k] = the last row stage k has handed over, read[k] = the last row stage k + 1 has
//     taken.  Stage k + 1 starts row r when done[k] >= r + 1 (it needs the bottom nodes of the row above); stage k writes row r
//     into slot r % 2 when read[k] >= r - 2.  A link is TWO steps deep -- the minimum the scheme allows: the velocity of the top
//     node row of element row r is updated by row r + 1 -- a strip takes R + 13 steps (R + 7 rows of stage 0, six steps of lag),
//     and two slots per link are enough: 96 KB for the three hand-overs.
//   * the freed LDS holds two RINGS of seven rows: the loader (stage 0) reads a row's ice strength (nine Gauss-point values) and
//     packed nodal coefficients from memory once and writes the ice strength and ONE of the three coefficient pairs (u_ocean,
//     v_ocean of the 4 owned nodes) into the rings, the stages 1-3 take them from there: 136 of the 264 bytes per element each of
//     them used to re-read.  The other two pairs (128 B) are still re-read from memory -- a ring for them needs another 2 x 28 KB
//     and the LDS is full: 96 + 31.5 + 28 = 155.5 of 160 KB (profiles/r05_fused4_p2p.md: what fits, what was measured).
//   * no barrier after the prologue: the waves run as far apart as their dependencies allow, idle steps do not exist.
//
// Memory ordering of the hand-over, the bounded wait and how a wait that gave up becomes an error status: mevp_p2p.h.  Every wait is
// on an event that is strictly earlier in the dependency graph of the march (row r of stage k + 1 waits for row r + 1 of stage k; row r
// of stage k waits for row r - 2 of stage k + 1, which waited for row r - 1 of stage k), so no wave can wait for ever.
//
// The arithmetic is the same sequence of inlined functions as in every other variant: bit-identical to four passes of variant 1.
#include "mevp_p2p.h"

namespace nsdg_mevp_detail {

#ifdef NSDG_P2P_SPINSTAT
// diagnostic build only (tools/ab_build.sh spin -DNSDG_P2P_SPINSTAT; tools/p2p_spinstat.py): polls per stage and kind of wait -- 0 the
// previous stage's hand-over, 1 a free hand-over slot, 2 a free ring slot -- and the rows the stage worked on in [3]
__device__ unsigned long long nsdg_p2p_spin_dev[4][4];
__device__ unsigned long long nsdg_p2p_phase_dev[4][8]; // shader cycles per stage and phase of a row (s_memtime, fenced)
#define NSDG_SPIN_ARG , unsigned (&spins)[3], unsigned (&phase)[8], unsigned& stamp_last
#define NSDG_SPIN_PASS , spins, phase, stamp_last
#define NSDG_SPIN_COUNT(k, n) spins[k] += (n)
#define NSDG_PHASE(k)                                                   \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();   \
        phase[k] += now_ - stamp_last;                                  \
        stamp_last = now_;                                              \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define NSDG_PHASE(k)
#define NSDG_SPIN_ARG
#define NSDG_SPIN_PASS
#define NSDG_SPIN_COUNT(k, n) (void)(n) /* the wait itself is the argument: it must stay */
#endif

constexpr int P4_OWNED = 57, P4_LEFT = 4; // lanes 4 .. 60 own a column (four sub-iterations reach four columns / rows)
constexpr int P4_HAND = 32; // doubles per lane and hand-over slot: 24 stress coefficients + u, v at the 4 owned nodes
constexpr int P4_SLOT = P4_HAND * 64; // value k of lane l at (k / 2) * 128 + 2 l + k % 2 (16-byte pairs)
constexpr int P4_HSLOTS = 2; // slots per link
constexpr int P4_PRING = 7; // rows in each of the two rings (a row is in flight for ~5.7 march steps; 7 x (4.6 + 4) KB is what fits)
constexpr int P4_PSLOT = P2P_PSLOT; // ice strength (mevp_p2p.h)
constexpr int P4_CSLOT = 8 * 64; // third pair of the nodal coefficients (u_ocean, v_ocean): node n of lane l at n * 128 + 2 l
constexpr int P4_LDS = 3 * P4_HSLOTS * P4_SLOT + P4_PRING * (P4_PSLOT + P4_CSLOT); // doubles: 96 + 31.5 + 28 KB of the 160 KB of a CU

struct FetchP {
    double P[9]; // ice strength at the Gauss points of the row the stage works on next
    double c[4][6]; // packed momentum coefficients of its 4 owned nodes
    double s11[8], s12[8], s22[8]; // stage 0 only: the stress the pass starts from
    double ub[3], vb[3], um[3], vm[3], ut[3], vt[3]; // stage 0 only: u, v of the pass's start on the three node rows
};

struct StageP {
    int s; // pipeline stage of this wave = sub-iteration p + s
    int nst; // stages of this pass = its sub-iterations (4; 3 or 2 for what is left of a sub-cycle whose length is no multiple of 4)
    int first, last; // element rows this stage works on
    int last_prev; // last row of the previous stage
    int last_final; // last row of the last stage (the loader's ring wait)
    int upd0; // node updates from this row on
};

// counters: done[k] at k, read[k] at 3 + k, the sticky give-up flag at 6
struct FlagsP {
    int v[8];
};

__device__ __forceinline__ int ring_slot(int row) { return row % P4_PRING; } // row >= 0
// the third pair (u_ocean, v_ocean) of the 4 owned nodes of a row from / to its ring
__device__ __forceinline__ void ring_read_c(const double* __restrict__ cring, int row, int lane, double (&c)[4][6])
{
    const double* s = cring + ring_slot(row) * P4_CSLOT + 2 * lane;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const double2 t = *reinterpret_cast<const double2*>(s + n * 128);
        c[n][4] = t.x, c[n][5] = t.y;
    }
}
__device__ __forceinline__ void ring_write_c(double* __restrict__ cring, int row, int lane, const double (&c)[4][6])
{
    double* s = cring + ring_slot(row) * P4_CSLOT + 2 * lane;
#pragma unroll
    for (int n = 0; n < 4; ++n)
        *reinterpret_cast<double2*>(s + n * 128) = make_double2(c[n][4], c[n][5]);
}
// the first two pairs of the nodal coefficients from memory (the third comes from the ring)
__device__ __forceinline__ void load_nodal2(const double* __restrict__ packed, long plane, long n, double (&c)[6])
{
    const double* p = packed + nodal_off(n, plane);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double2 t = *reinterpret_cast<const double2*>(p + k * plane);
        c[2 * k] = t.x, c[2 * k + 1] = t.y;
    }
}
__device__ __forceinline__ void request_c2_p(const MarchConst3& M, int nrow, double (&c)[4][6], const double* __restrict__ packed)
{
    const long nVn = (long)(2 * nrow) * M.nn + 2 * M.ix;
    load_nodal2(packed, M.nplane, nVn, c[0]);
    load_nodal2(packed, M.nplane, nVn + 1, c[1]);
    load_nodal2(packed, M.nplane, nVn + M.nn, c[2]);
    load_nodal2(packed, M.nplane, nVn + M.nn + 1, c[3]);
}
// One row of one stage.  FIRST: the loader (stage 0): inputs from memory, ice strength into the ring.
template <bool FIRST, bool AD>
__device__ __forceinline__ void p2p_row(const MarchConst3& M, const StageP& G, int row, FetchP& f, TopCarry3& carry, double* __restrict__ lds,
    volatile lds_int* flags, const P2PReport& rep, const StressPtrsP& S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new NSDG_SPIN_ARG)
{
    const int stage = FIRST ? 0 : G.s;
    const int nrow = min(row + 1, G.last); // the row this stage works on next: its inputs are requested during this one
    const int ix = M.ix, nn = M.nn;
    double* const ring = lds + 3 * P4_HSLOTS * P4_SLOT;
    double* const cring = ring + P4_PRING * P4_PSLOT;
    // the coefficients of row r: a stage >= 1 takes two pairs from memory and the third from the ring
    auto request_c = [&](int r) {
        if (FIRST)
            request_c_p(M, r, f.c, packed);
        else {
            request_c2_p(M, r, f.c, packed);
            ring_read_c(cring, r, M.lane, f.c);
        }
    };
    const long nVn = (long)(2 * nrow) * nn + 2 * ix; // vertex node of the next row
    double s11[8], s12[8], s22[8], uu[4], vv[4], ul[9], vl[9], un[4], vn[4];
    // ------------------------------------------------------------------------------------------ inputs of the row
    if (FIRST) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ul[a] = f.ub[a], ul[3 + a] = f.um[a], ul[6 + a] = f.ut[a];
            vl[a] = f.vb[a], vl[3 + a] = f.vm[a], vl[6 + a] = f.vt[a];
        }
        uu[0] = ul[0], uu[1] = ul[1], uu[2] = ul[3], uu[3] = ul[4];
        vv[0] = vl[0], vv[1] = vl[1], vv[2] = vl[3], vv[3] = vl[4];
    } else {
        // the previous stage has handed over this row and the row above it (whose bottom nodes are this row's top nodes)
        NSDG_SPIN_COUNT(0, flag_wait(flags, stage - 1, min(row + 1, G.last_prev), rep));
        if (row == G.first) { // wave-uniform: the first row of the stage has no predecessor that requested its inputs
            ring_read_P<P4_PRING>(ring, row, M.lane, f.P);
            request_c(row);
        }
        const double* in = lds + ((stage - 1) * P4_HSLOTS + (row & 1)) * P4_SLOT + 2 * M.lane;
        const double* top = lds + ((stage - 1) * P4_HSLOTS + ((row + 1) & 1)) * P4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double2 a = lds_pair_p(in, 12 + k), b = lds_pair_p(in, 14 + k);
            uu[2 * k] = a.x, uu[2 * k + 1] = a.y, vv[2 * k] = b.x, vv[2 * k + 1] = b.y;
        }
        double2 tu = lds_pair_p(top, 12), tv = lds_pair_p(top, 14);
        if (row + 1 > G.last_prev) // wave-uniform: node row 2*ny is the top boundary
            tu = tv = make_double2(0., 0.);
        gather_nodes(M, uu, tu.x, tu.y, ul);
        gather_nodes(M, vv, tv.x, tv.y, vl);
    }
    NSDG_PHASE(0); // inputs of the row (stages >= 1: the wait for the previous stage, LDS reads)
    // ------------------------------------------------------------------------------------------ stress update
    double r11[8], r12[8], r22[8];
    double qe = 0., ialpha = M.ialpha; // adaptive form: this element's offer q_e = alpha_e h'_c and 1 / alpha_e of this sub-iteration (the centre node's h' is coefficient [0] of node 3)
    if constexpr (AD)
        stress_projected_adaptive(ul, vl, f.P, M.ihx, M.ihy, M.dmin2, f.c[3][0], M.AC, r11, r12, r22, qe, ialpha);
    else
        stress_projected(ul, vl, f.P, M.ihx, M.ihy, M.ialpha, M.dmin2, r11, r12, r22);
    __builtin_amdgcn_sched_barrier(0);
    NSDG_PHASE(1); // projected stress
    if (FIRST) {
        // the ice strength of this row goes to the ring for the stages 1-3 (slot of row - 8: stage 3 has passed it), then the
        // register set takes the next row's; u, v of the next row
        NSDG_SPIN_COUNT(2, flag_wait(flags, 3 + G.nst - 2, min(row - P4_PRING, G.last_final), rep)); // read[] of the LAST link: the last stage has passed that row
        ring_write_P<P4_PRING>(ring, row, M.lane, f.P);
        ring_write_c(cring, row, M.lane, f.c); // this row's coefficients were requested a step ago; the stages 1-3 take the pair from here
        tile_load9_p<(NSDG_P2P_NT & 4) != 0>(pg, tile_off(ix, nrow, M.ntx, 9), ix & 63, f.P);
        if (nrow > row) { // wave-uniform: the top node row of this element row is the bottom one of the next
#pragma unroll
            for (int a = 0; a < 3; ++a)
                f.ub[a] = f.ut[a], f.vb[a] = f.vt[a];
        }
        fetch_nodes_p(u_old, nVn + nn, f.um);
        fetch_nodes_p(v_old, nVn + nn, f.vm);
        fetch_nodes_p(u_old, nVn + 2 * nn, f.ut);
        fetch_nodes_p(v_old, nVn + 2 * nn, f.vt);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s11[i] = f.s11[i], s12[i] = f.s12[i], s22[i] = f.s22[i];
    } else {
        ring_read_P<P4_PRING>(ring, nrow, M.lane, f.P); // written by the loader before it handed row nrow over: done[stage - 1] >= row + 1 implies done[0] >= nrow
        const double* in = lds + ((stage - 1) * P4_HSLOTS + (row & 1)) * P4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 a = lds_pair_p(in, k), b = lds_pair_p(in, 4 + k), c = lds_pair_p(in, 8 + k);
            s11[2 * k] = a.x, s11[2 * k + 1] = a.y, s12[2 * k] = b.x, s12[2 * k + 1] = b.y, s22[2 * k] = c.x, s22[2 * k + 1] = c.y;
        }
        // this row's slot (stress, u, v) and the next row's ice strength have been taken: the producer may write row + 2 into the slot
        flag_publish(flags, 3 + stage - 1, row);
    }
    NSDG_PHASE(2); // loader: ring wait, ring writes, requests of P, u, v; stages: ring read, stress read, read[] published
    if constexpr (AD)
        stress_relax_adaptive(ialpha, r11, r12, r22, s11, s12, s22);
    else
        stress_relax(M.ialpha, r11, r12, r22, s11, s12, s22);
    __builtin_amdgcn_sched_barrier(0);
    if (FIRST) { // stress of the next row
        const long ts = tile_off(ix, nrow, M.ntx, 8);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i11, ts, f.s11);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i12, ts, f.s12);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i22, ts, f.s22);
    }
    NSDG_PHASE(3); // relaxation, (loader) stress request
    // ------------------------------------------------------------------------------------------ contributions, node updates
    {
        double cx[9], cy[9];
        node_contrib_all(s11, s12, s22, M.hx, M.hy, cx, cy);
        owned_node_updates<AD>(M, row > 0, f.c, uu, vv, carry, cx, cy, un, vn, qe);
        if (row < G.upd0) { // wave-uniform: the first row of a stage only feeds the carried contributions
#pragma unroll
            for (int k = 0; k < 4; ++k)
                un[k] = vn[k] = 0.;
        }
        carry_top<AD>(carry, cx, cy, qe);
    }
    __builtin_amdgcn_sched_barrier(0);
    NSDG_PHASE(4); // contributions, node updates (the wait for the nodal coefficients is here)
    request_c(nrow); // nodal coefficients of the next row
    NSDG_PHASE(5); // request of the coefficients
    // ------------------------------------------------------------------------------------------ outputs
    if (FIRST || stage < G.nst - 1) {
        NSDG_SPIN_COUNT(1, flag_wait(flags, 3 + stage, row - P4_HSLOTS, rep)); // the consumer has taken the row this slot held
        double* out = lds + (stage * P4_HSLOTS + (row & 1)) * P4_SLOT + 2 * M.lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lds_pair_p(out, k, s11[2 * k], s11[2 * k + 1]);
            lds_pair_p(out, 4 + k, s12[2 * k], s12[2 * k + 1]);
            lds_pair_p(out, 8 + k, s22[2 * k], s22[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lds_pair_p(out, 12 + k, un[2 * k], un[2 * k + 1]);
            lds_pair_p(out, 14 + k, vn[2 * k], vn[2 * k + 1]);
        }
        flag_publish(flags, stage, row);
    } else if (M.own && row >= M.y0) { // the last stage runs on rows y0-1 .. y1-1
        const long ts = tile_off(ix, row, M.ntx, 8);
        const long nV = (long)(2 * row) * nn + 2 * ix;
        tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o11, ts, s11);
        tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o12, ts, s12);
        tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o22, ts, s22);
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
    NSDG_PHASE(6); // outputs: slot wait, LDS writes, done[] published / global stores
}

template <bool AD>
__global__ __launch_bounds__(256) void mevp_fused4_kernel(NodalConsts K, AdaptConsts AC, int nst, int nx, int ny, int j0, int j1, int j0b, int j1b, int nsA, int R, int ncw,
    double hx, double hy, double ialpha, double dmin2, P2PReport rep, StressPtrsP S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    __shared__ __attribute__((aligned(16))) double lds[P4_LDS]; // 96 KB of hand-over slots + 31.5 + 28 KB of rings (ice strength, one coefficient pair)
    __shared__ FlagsP flagmem;
    volatile lds_int* flags = (volatile lds_int*)flagmem.v;
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
        return; // workgroup-uniform: no wave of this workgroup reaches the barrier or a counter
    M.y1 = min(M.y0 + R, j1);
    const int ixr = cw * P4_OWNED - P4_LEFT + lane;
    const bool valid = ixr >= 0 && ixr < nx;
    M.K = K;
    M.nx = nx, M.ny = ny, M.lane = lane;
    M.own = valid && lane >= P4_LEFT && lane < P4_LEFT + P4_OWNED;
    M.ix = min(max(ixr, 0), nx - 1);
    M.hasL = M.ix > 0, M.lastcol = M.ix == nx - 1;
    M.ntx = tiles_per_row(nx);
    M.nn = 2 * nx + 1;
    M.nplane = nodal_plane((long)M.nn * (2 * ny + 1));
    M.hx = hx, M.hy = hy, M.ihx = 1. / hx, M.ihy = 1. / hy, M.iarea = M.ihx * M.ihy;
    M.ialpha = ialpha, M.dmin2 = dmin2;
    M.AC = AC;
    M.tbeg = M.tendA = M.tendB = 0; // (fields of the other pipelines)

    // a pass of nst sub-iterations (2 <= nst <= 4): stage s works on the rows y0 - nst + s .. y1 + nst - 2 - s, the last one (nst - 1)
    // on y0 - 1 .. y1 - 1; the waves s >= nst have nothing to do
    StageP G;
    G.s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    G.nst = nst;
    auto first_of = [&](int s) { return max(M.y0 - nst + s, 0); };
    auto last_of = [&](int s) { return min(M.y1 + nst - 2 - s, ny - 1); };
    G.first = first_of(G.s);
    G.last = last_of(G.s);
    G.last_prev = last_of(G.s - 1);
    G.last_final = last_of(nst - 1);
    G.upd0 = G.s == 0 ? 0 : M.y0 - (nst - 1) + G.s;

    // counters: done[k] = first row of stage k - 1 (nothing handed over yet), read[k] = first row of stage k + 1 - 1 (every row
    // below the consumer's first one counts as taken: the consumer never looks at it)
    if (threadIdx.x < 3) {
        flags[threadIdx.x] = first_of(threadIdx.x) - 1;
        flags[3 + threadIdx.x] = first_of(threadIdx.x + 1) - 1;
    }
    if (threadIdx.x == P2P_GIVEUP)
        flags[P2P_GIVEUP] = 0;
    __syncthreads(); // the only barrier of the kernel
    if (G.s >= nst)
        return; // wave-uniform

    FetchP f;
    TopCarry3 carry; // zero by its member initialisers: the first row of a stage adds nothing from a row below
#ifdef NSDG_P2P_SPINSTAT
    unsigned spins[3] = { 0, 0, 0 }, phase[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned stamp_last = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    if (G.s == 0) {
        const int row = G.first;
        const long nV = (long)(2 * row) * M.nn + 2 * M.ix, ts = tile_off(M.ix, row, M.ntx, 8);
        fetch_nodes_p(u_old, nV, f.ub);
        fetch_nodes_p(v_old, nV, f.vb);
        fetch_nodes_p(u_old, nV + M.nn, f.um);
        fetch_nodes_p(v_old, nV + M.nn, f.vm);
        fetch_nodes_p(u_old, nV + 2 * M.nn, f.ut);
        fetch_nodes_p(v_old, nV + 2 * M.nn, f.vt);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i11, ts, f.s11);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i12, ts, f.s12);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i22, ts, f.s22);
        tile_load9_p<(NSDG_P2P_NT & 4) != 0>(pg, tile_off(M.ix, row, M.ntx, 9), M.ix & 63, f.P);
        request_c_p(M, row, f.c, packed);
        for (int row = G.first; row <= G.last; ++row)
            p2p_row<true, AD>(M, G, row, f, carry, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new NSDG_SPIN_PASS);
    } else {
        for (int row = G.first; row <= G.last; ++row)
            p2p_row<false, AD>(M, G, row, f, carry, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new NSDG_SPIN_PASS);
    }
#ifdef NSDG_P2P_SPINSTAT
    if (lane == 0) {
        for (int k = 0; k < 3; ++k)
            atomicAdd(&nsdg_p2p_spin_dev[G.s][k], (unsigned long long)spins[k]);
        atomicAdd(&nsdg_p2p_spin_dev[G.s][3], (unsigned long long)(G.last - G.first + 1));
        for (int k = 0; k < 8; ++k)
            atomicAdd(&nsdg_p2p_phase_dev[G.s][k], (unsigned long long)phase[k]);
    }
#endif
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

#ifdef NSDG_P2P_SPINSTAT
extern "C" int nsdg_debug_p2p_spinstat(unsigned long long* out48) // [16] polls, then [32] phase cycles
{
    static const unsigned long long zero[32] = { 0 };
    hipError_t e = hipMemcpyFromSymbol(out48, HIP_SYMBOL(nsdg_p2p_spin_dev), 16 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemcpyFromSymbol(out48 + 16, HIP_SYMBOL(nsdg_p2p_phase_dev), 32 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(nsdg_p2p_spin_dev), zero, 16 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(nsdg_p2p_phase_dev), zero, 32 * sizeof(unsigned long long));
    return (int)e;
}
#endif

int nsdg_launch_mevp_fused4_ranges(nsdg_ctx* ctx, int nst, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i,
    double* s11, double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed,
    const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, P4_OWNED);
    const int rowsB = j0b < j1b ? j1b - j0b : 0;
    int R = ctx->strip_rows;
    if (R <= 0) {
        // a strip of R rows takes about R + 4 nst - 3 march steps (R + 2 nst - 1 rows of stage 0, 2 (nst - 1) steps of lag: R + 13 for
        // four sub-iterations); one resident workgroup per CU (LDS)
        const double extra = 4. * nst - 3.;
        const long slots = ctx->num_cus;
        double best = 1e30;
        R = 64;
        for (int r = 1; r <= 4096; ++r) {
            const long groups = ((long)nsdg_div_up(j1 - j0, r) + nsdg_div_up(rowsB, r)) * ncw;
            const long rounds = (groups + slots - 1) / slots;
            const double cost = rounds * (r + extra);
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
    const StressPtrsP S = { s11i, s12i, s22i, s11, s12, s22 };
    const NodalConsts K = nsdg_nodal_consts(ctx);
    const AdaptConsts AC = nsdg_adapt_consts(ctx);
    const P2PReport rep = { ctx->p2p_count_dev, ctx->p2p_flag_dev };
    if (nsdg_adaptive(ctx)) // local, solution-adaptive alpha and beta (mevp_common.h)
        hipLaunchKernelGGL(mevp_fused4_kernel<true>, dim3(ngroups), dim3(256), 0, ctx->stream, K, AC, nst, ctx->nx, ctx->ny, j0, j1, j0b, j1b, nsA, R, ncw, ctx->hx,
            ctx->hy, 1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, rep, S, u_old, v_old, packed, pg, u_new, v_new);
    else
        hipLaunchKernelGGL(mevp_fused4_kernel<false>, dim3(ngroups), dim3(256), 0, ctx->stream, K, AC, nst, ctx->nx, ctx->ny, j0, j1, j0b, j1b, nsA, R, ncw, ctx->hx,
            ctx->hy, 1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, rep, S, u_old, v_old, packed, pg, u_new, v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}
