// mevp_fused8.hip -- EIGHT sub-iterations of the mEVP sub-cycle per kernel pass (round 6): the stage-per-wave pipeline of mevp_fused4.hip
// with TWO sub-iterations per stage wave.  At march step t the wave s of a four-wave workgroup performs
//     A(t): sub-iteration p + 2 s      on element row t      (stress and u, v at the row's 4 owned nodes stay in registers)
//     B(t): sub-iteration p + 2 s + 1  on element row t - 1  (from what A(t - 1) left in registers; the row's top nodes are A(t)'s)
// and hands B's row to the wave s + 1 through LDS, point to point (mevp_p2p.h: counters, bounded waits).  The hand-over inside a wave
// costs nothing: two register sets alternate between "written by A(t)" and "read by B(t + 1)".
//
// Why (profiles/r05_fused4_p2p.md section 6, profiles/r06_fused8.md): the four-sub-iteration pass runs at the power cap and spends more of
// its energy on memory traffic than on arithmetic.  A pass of eight moves the same compulsory 776 bytes per element -- stress in and
// out, ice strength, packed coefficients, u, v in and out -- for twice the sub-iterations, and halves the LDS hand-overs and ring
// reads per sub-iteration.  The price is redundancy: eight sub-iterations reach eight columns and rows, so a wave owns 49 of its 64
// columns (lanes 8 .. 56; the four-pass owns 57) and sub-iteration q of a pass runs on the rows y0 - 8 + q .. y1 + 6 - q.
//
// Per lane: the register sets X, Y hold a row's 24 stress coefficients, u, v at its 4 owned nodes and its 24 packed nodal coefficients
// (loaded once per row and wave, used by A and, a step later, by B); the ice strength of a row is read twice from a ring in LDS
// (12 rows, written by the loader = wave 0 from memory).  LDS: 3 links x 2 slots x 16 KB + 12 x 4.5 KB = 150 KB.
//
// Counters (LDS): done[k] = last row wave k's B handed over; read[k] = last row wave k + 1's A took; ring = last row whose ice strength
// the LAST wave's B has read.  Wave s + 1 starts A on row t when done[s] >= t + 1 (it needs the bottom nodes of the row above); wave s
// writes B's row r into slot r % 2 when read[s] >= r - 2; the loader overwrites the ring slot of row t - 12 when ring >= t - 12.  Every
// wait is on an event strictly earlier in the dependency graph of the march (mevp_fused4.hip), and bounded anyway.
//
// The arithmetic is the same sequence of inlined functions as in every other variant: bit-identical to eight passes of variant 1.
#include "mevp_p2p.h"

namespace nsdg_mevp_detail {

constexpr int P8_SUB = 8; // sub-iterations per pass
constexpr int P8_WAVES = 4; // stage waves, two sub-iterations each
constexpr int P8_LEFT = 8, P8_OWNED = 65 - 2 * P8_SUB; // lanes 8 .. 56 own a column
constexpr int P8_HAND = 32; // doubles per lane and hand-over slot: 24 stress coefficients + u, v at the 4 owned nodes
constexpr int P8_SLOT = P8_HAND * 64; // value k of lane l at (k / 2) * 128 + 2 l + k % 2 (16-byte pairs)
constexpr int P8_HSLOTS = 2; // slots per link
#ifndef NSDG_P8_PRING
#ifdef NSDG_P8_TIMING
#define NSDG_P8_PRING 10
#else
#define NSDG_P8_PRING 12
#endif
#endif
constexpr int P8_PRING = NSDG_P8_PRING; // rows of the ice-strength ring: a row is in flight for ~8 march steps (loader's A .. last wave's B)
#ifdef NSDG_P8_TIMING
constexpr int P8_STAGE = 24 * 64 + (NSDG_P8_TIMING >= 2 ? 8 * 64 : 0); // the loader's next row of stress (and of u, v), filled by LDS-DMA
#else
constexpr int P8_STAGE = 0;
#endif
constexpr int P8_LDS = 3 * P8_HSLOTS * P8_SLOT + P8_PRING * P2P_PSLOT + P8_STAGE; // doubles
constexpr int P8_RINGFLAG = 7; // counter index: last row whose ice strength the last wave's B has read
static_assert(P8_LDS * 8 + 64 <= 160 * 1024, "LDS of a compute unit");

// what sub-iteration p + 1 needs from sub-iteration p of the same wave, per element row
struct RowSet8 {
    double s11[8], s12[8], s22[8]; // stress after A (relaxed in place by B)
    double u[4], v[4]; // u, v after A at the 4 owned nodes
    double c[4][6]; // packed momentum coefficients of those nodes
};

// the loader's requests for the row it works on next
struct Fetch8 {
    double P[9];
    double s11[8], s12[8], s22[8];
    double ub[3], vb[3], um[3], vm[3], ut[3], vt[3];
};

struct Wave8 {
    int s; // stage wave: sub-iterations 2 s (A) and 2 s + 1 (B) of the pass
    int firstA, lastA, firstB, lastB; // element rows A / B work on
    int updA0, updB0; // node updates from this row on (the first row of a sub-iteration only feeds the carried contributions)
    int lastB_prev; // last row of the previous wave's B
    int lastB_final; // last row of the last wave's B (the loader's ring wait)
};

#ifdef NSDG_P8_TIMING
// 16 bytes per lane from a per-lane address straight into LDS at lds_dst + 16 * lane; hidden from the compiler's s_waitcnt bookkeeping
__device__ __forceinline__ void glds16(const double* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_stress(const double* __restrict__ a, long t, double* lds_stage, int comp)
{
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)lds_stage + comp * 4 * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        glds16(a + t + 128 * k, __builtin_amdgcn_readfirstlane(base + k * 1024));
}
#endif
// One march step of one wave.  FIRST: the loader (wave 0): A's inputs from memory, ice strength into the ring.
template <bool FIRST>
__device__ __forceinline__ void p8_step(const MarchConst3& M, const Wave8& G, int t, RowSet8& cur, RowSet8& prev, Fetch8& f, TopCarry3& ca, TopCarry3& cb,
    double* __restrict__ lds, volatile lds_int* flags, const P2PReport& rep, const StressPtrsP& S, const double* __restrict__ u_old,
    const double* __restrict__ v_old, const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    const int s = FIRST ? 0 : G.s;
    const int ix = M.ix, nn = M.nn;
    double* const ring = lds + 3 * P8_HSLOTS * P8_SLOT;
    const int nrow = min(t + 1, G.lastA); // the row A works on next
#if defined(NSDG_P8_TIMING) && NSDG_P8_TIMING >= 2
    constexpr bool LOADS = false; // TIMING ONLY: the loader computes on what it finds in LDS (garbage), like a stage wave; its memory traffic is LDS-DMA
    const int sp = FIRST ? 2 : s - 1;
#else
    constexpr bool LOADS = FIRST;
    const int sp = s - 1;
#endif
    // ================================================================================== A(t): sub-iteration 2 s on row t -> cur
    if (t <= G.lastA) { // wave-uniform (the last step of a strip that ends at the top boundary only drains B)
        double ul[9], vl[9], uu[4], vv[4], P[9];
        if (LOADS) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ul[a] = f.ub[a], ul[3 + a] = f.um[a], ul[6 + a] = f.ut[a];
                vl[a] = f.vb[a], vl[3 + a] = f.vm[a], vl[6 + a] = f.vt[a];
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
                P[q] = f.P[q];
            uu[0] = ul[0], uu[1] = ul[1], uu[2] = ul[3], uu[3] = ul[4];
            vv[0] = vl[0], vv[1] = vl[1], vv[2] = vl[3], vv[3] = vl[4];
        } else {
            // the previous wave has handed over this row and the row above it (whose bottom nodes are this row's top nodes)
            if (!FIRST)
                (void)flag_wait(flags, s - 1, min(t + 1, G.lastB_prev), rep);
            ring_read_P<P8_PRING>(ring, t, M.lane, P); // written by the loader before it handed row t over: done[s - 1] >= t + 1 implies its A(t)
            const double* in = lds + (sp * P8_HSLOTS + (t & 1)) * P8_SLOT + 2 * M.lane;
            const double* top = lds + (sp * P8_HSLOTS + ((t + 1) & 1)) * P8_SLOT + 2 * M.lane;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double2 a = lds_pair_p(in, 12 + k), b = lds_pair_p(in, 14 + k);
                uu[2 * k] = a.x, uu[2 * k + 1] = a.y, vv[2 * k] = b.x, vv[2 * k + 1] = b.y;
            }
            double2 tu = lds_pair_p(top, 12), tv = lds_pair_p(top, 14);
            if (t + 1 > G.lastB_prev) // wave-uniform: node row 2 * ny is the top boundary
                tu = tv = make_double2(0., 0.);
            gather_nodes(M, uu, tu.x, tu.y, ul);
            gather_nodes(M, vv, tv.x, tv.y, vl);
        }
        double r11[8], r12[8], r22[8];
        stress_projected(ul, vl, P, M.ihx, M.ihy, M.ialpha, M.dmin2, r11, r12, r22);
        __builtin_amdgcn_sched_barrier(0);
        if (LOADS) {
            // the ice strength of this row goes to the ring (slot of row t - 12: the last wave's B has passed it)
            (void)flag_wait(flags, P8_RINGFLAG, min(t - P8_PRING, G.lastB_final), rep);
            ring_write_P<P8_PRING>(ring, t, M.lane, P);
#ifdef NSDG_P8_TIMING
            {
                const double* st = ring + P8_PRING * P2P_PSLOT + 2 * M.lane;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double2 a = lds_pair_p(st, k), b = lds_pair_p(st, 4 + k), c = lds_pair_p(st, 8 + k);
                    cur.s11[2 * k] = a.x, cur.s11[2 * k + 1] = a.y, cur.s12[2 * k] = b.x, cur.s12[2 * k + 1] = b.y, cur.s22[2 * k] = c.x, cur.s22[2 * k + 1] = c.y;
                }
            }
#else
#pragma unroll
            for (int i = 0; i < 8; ++i)
                cur.s11[i] = f.s11[i], cur.s12[i] = f.s12[i], cur.s22[i] = f.s22[i];
#endif
            // requests for the next row: ice strength, u, v (its bottom node row is this row's top one)
            tile_load9_p<(NSDG_P2P_NT & 4) != 0>(pg, tile_off(ix, nrow, M.ntx, 9), ix & 63, f.P);
            if (nrow > t) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    f.ub[a] = f.ut[a], f.vb[a] = f.vt[a];
            }
            const long nVn = (long)(2 * nrow) * nn + 2 * ix;
            fetch_nodes_p(u_old, nVn + nn, f.um);
            fetch_nodes_p(v_old, nVn + nn, f.vm);
            fetch_nodes_p(u_old, nVn + 2 * nn, f.ut);
            fetch_nodes_p(v_old, nVn + 2 * nn, f.vt);
        } else {
            const double* in = lds + (sp * P8_HSLOTS + (t & 1)) * P8_SLOT + 2 * M.lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2 a = lds_pair_p(in, k), b = lds_pair_p(in, 4 + k), c = lds_pair_p(in, 8 + k);
                cur.s11[2 * k] = a.x, cur.s11[2 * k + 1] = a.y, cur.s12[2 * k] = b.x, cur.s12[2 * k + 1] = b.y, cur.s22[2 * k] = c.x, cur.s22[2 * k + 1] = c.y;
            }
            if (!FIRST)
                flag_publish(flags, 3 + s - 1, t); // this row's slot has been taken: the producer may write row t + 2 into it
        }
        stress_relax(M.ialpha, r11, r12, r22, cur.s11, cur.s12, cur.s22);
        __builtin_amdgcn_sched_barrier(0);
        if (FIRST) { // stress of the next row
            const long ts = tile_off(ix, nrow, M.ntx, 8);
#if defined(NSDG_P8_TIMING) && NSDG_P8_TIMING == 3
            (void)ts; // mode 3: no memory traffic at all in the loader (the speed of the stage arithmetic and hand-over alone)
#elif defined(NSDG_P8_TIMING)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the staging buffer has been read
            double* st = ring + P8_PRING * P2P_PSLOT;
            glds_stress(S.i11, ts, st, 0);
            glds_stress(S.i12, ts, st, 1);
            glds_stress(S.i22, ts, st, 2);
#if NSDG_P8_TIMING == 2
            {
                // ice strength of the next row straight into its ring slot (the ninth value is skipped), u, v of its two new node rows
                (void)flag_wait(flags, P8_RINGFLAG, min(nrow - P8_PRING, G.lastB_final), rep);
                glds_stress(pg, tile_off(ix, nrow, M.ntx, 9), ring + (nrow % P8_PRING) * P2P_PSLOT, 0);
                const unsigned ub = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(st + 24 * 64);
                const long nVn = (long)(2 * nrow) * nn + 2 * ix;
                glds16(u_old + nVn + nn, __builtin_amdgcn_readfirstlane(ub));
                glds16(v_old + nVn + nn, __builtin_amdgcn_readfirstlane(ub + 1024));
                glds16(u_old + nVn + 2 * nn, __builtin_amdgcn_readfirstlane(ub + 2048));
                glds16(v_old + nVn + 2 * nn, __builtin_amdgcn_readfirstlane(ub + 3072));
            }
#endif
#else
            tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i11, ts, f.s11);
            tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i12, ts, f.s12);
            tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i22, ts, f.s22);
#endif
        }
        {
            double cx[9], cy[9];
            node_contrib_all(cur.s11, cur.s12, cur.s22, M.hx, M.hy, cx, cy);
            owned_node_updates(M, t > 0, cur.c, uu, vv, ca, cx, cy, cur.u, cur.v);
            if (t < G.updA0) { // wave-uniform
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    cur.u[k] = cur.v[k] = 0.;
            }
            carry_top(ca, cx, cy);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cur.u[k] = cur.v[k] = 0.; // node row 2 * ny is the top boundary
    }
    __builtin_amdgcn_sched_barrier(0);
    // ================================================================================== B(t): sub-iteration 2 s + 1 on row r = t - 1, from prev
    const int r = t - 1;
    if (r >= G.firstB) { // wave-uniform
        double ul[9], vl[9], P[9], un[4], vn[4];
        ring_read_P<P8_PRING>(ring, r, M.lane, P);
        if (!FIRST && s == P8_WAVES - 1)
            flag_publish(flags, P8_RINGFLAG, r); // the last reader of the ring has taken row r
        gather_nodes(M, prev.u, cur.u[0], cur.u[1], ul);
        gather_nodes(M, prev.v, cur.v[0], cur.v[1], vl);
        {
            double r11[8], r12[8], r22[8];
            stress_projected(ul, vl, P, M.ihx, M.ihy, M.ialpha, M.dmin2, r11, r12, r22);
            __builtin_amdgcn_sched_barrier(0);
            stress_relax(M.ialpha, r11, r12, r22, prev.s11, prev.s12, prev.s22);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double cx[9], cy[9];
            node_contrib_all(prev.s11, prev.s12, prev.s22, M.hx, M.hy, cx, cy);
#ifdef NSDG_P8_TIMING
            // TIMING ONLY (wrong values): the loader's B uses A's coefficients, so that the loader carries one set of them, not two
            owned_node_updates(M, r > 0, FIRST ? cur.c : prev.c, prev.u, prev.v, cb, cx, cy, un, vn);
#else
            owned_node_updates(M, r > 0, prev.c, prev.u, prev.v, cb, cx, cy, un, vn);
#endif
            if (r < G.updB0) { // wave-uniform
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    un[k] = vn[k] = 0.;
            }
            carry_top(cb, cx, cy);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ------------------------------------------------------------------------------------------ outputs
        if (FIRST || s < P8_WAVES - 1) {
            (void)flag_wait(flags, 3 + s, r - P8_HSLOTS, rep); // the consumer has taken the row this slot held
            double* out = lds + (s * P8_HSLOTS + (r & 1)) * P8_SLOT + 2 * M.lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds_pair_p(out, k, prev.s11[2 * k], prev.s11[2 * k + 1]);
                lds_pair_p(out, 4 + k, prev.s12[2 * k], prev.s12[2 * k + 1]);
                lds_pair_p(out, 8 + k, prev.s22[2 * k], prev.s22[2 * k + 1]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                lds_pair_p(out, 12 + k, un[2 * k], un[2 * k + 1]);
                lds_pair_p(out, 14 + k, vn[2 * k], vn[2 * k + 1]);
            }
            flag_publish(flags, s, r);
        } else if (M.own && r >= M.y0) { // the last sub-iteration runs on rows y0 - 1 .. y1 - 1
            const long ts = tile_off(ix, r, M.ntx, 8);
            const long nV = (long)(2 * r) * nn + 2 * ix;
            tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o11, ts, prev.s11);
            tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o12, ts, prev.s12);
            tile_store8_p<(NSDG_P2P_NT & 2) != 0>(S.o22, ts, prev.s22);
            u_new[nV] = un[0], v_new[nV] = vn[0];
            u_new[nV + 1] = un[1], v_new[nV + 1] = vn[1];
            u_new[nV + nn] = un[2], v_new[nV + nn] = vn[2];
            u_new[nV + nn + 1] = un[3], v_new[nV + nn + 1] = vn[3];
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
    __builtin_amdgcn_sched_barrier(0);
    // nodal coefficients of the row A works on next, into the set B has just finished with (prev becomes cur)
    if (t < G.lastA)
        request_c_p(M, t + 1, prev.c, packed);
}

__global__ __launch_bounds__(256) void mevp_fused8_kernel(NodalConsts K, int nx, int ny, int j0, int j1, int j0b, int j1b, int nsA, int R, int ncw, double hx,
    double hy, double ialpha, double dmin2, P2PReport rep, StressPtrsP S, const double* __restrict__ u_old, const double* __restrict__ v_old,
    const double* __restrict__ packed, const double* __restrict__ pg, double* __restrict__ u_new, double* __restrict__ v_new)
{
    __shared__ __attribute__((aligned(16))) double lds[P8_LDS]; // 96 KB of hand-over slots + 54 KB of ice-strength ring
    __shared__ int flagmem[8];
    volatile lds_int* flags = (volatile lds_int*)flagmem;
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
    const int ixr = cw * P8_OWNED - P8_LEFT + lane;
    const bool valid = ixr >= 0 && ixr < nx;
    M.K = K;
    M.nx = nx, M.ny = ny, M.lane = lane;
    M.own = valid && lane >= P8_LEFT && lane < P8_LEFT + P8_OWNED;
    M.ix = min(max(ixr, 0), nx - 1);
    M.hasL = M.ix > 0, M.lastcol = M.ix == nx - 1;
    M.ntx = tiles_per_row(nx);
    M.nn = 2 * nx + 1;
    M.nplane = nodal_plane((long)M.nn * (2 * ny + 1));
    M.hx = hx, M.hy = hy, M.ihx = 1. / hx, M.ihy = 1. / hy, M.iarea = M.ihx * M.ihy;
    M.ialpha = ialpha, M.dmin2 = dmin2;
    M.tbeg = M.tendA = M.tendB = 0; // (fields of the other pipelines)

    // sub-iteration q of the pass works on the rows y0 - 8 + q .. y1 + 6 - q, the last one (7) on y0 - 1 .. y1 - 1
    auto first_of = [&](int q) { return max(M.y0 - P8_SUB + q, 0); };
    auto last_of = [&](int q) { return min(M.y1 + P8_SUB - 2 - q, ny - 1); };
    auto upd0_of = [&](int q) { return q == 0 ? 0 : M.y0 - (P8_SUB - 1) + q; };
    Wave8 G;
    G.s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    G.firstA = first_of(2 * G.s), G.lastA = last_of(2 * G.s);
    G.firstB = first_of(2 * G.s + 1), G.lastB = last_of(2 * G.s + 1);
    G.updA0 = upd0_of(2 * G.s), G.updB0 = upd0_of(2 * G.s + 1);
    G.lastB_prev = last_of(2 * G.s - 1);
    G.lastB_final = last_of(P8_SUB - 1);

    // counters: done[k] = first row of wave k's B - 1 (nothing handed over yet); read[k] = first row of wave k + 1's A - 1 and ring = first
    // row of the last B - 1 (every row below the consumer's first one counts as taken: the consumer never looks at it)
    if (threadIdx.x < 3) {
        flags[threadIdx.x] = first_of(2 * threadIdx.x + 1) - 1;
        flags[3 + threadIdx.x] = first_of(2 * threadIdx.x + 2) - 1;
    }
    if (threadIdx.x == P2P_GIVEUP)
        flags[P2P_GIVEUP] = 0;
    if (threadIdx.x == P8_RINGFLAG)
        flags[P8_RINGFLAG] = first_of(P8_SUB - 1) - 1;
    __syncthreads(); // the only barrier of the kernel

    RowSet8 X, Y; // alternate between "written by A" and "read by B": nothing is copied when the march advances
    Fetch8 f;
    TopCarry3 ca, cb; // zero by their member initialisers
#pragma unroll
    for (int k = 0; k < 4; ++k)
        Y.u[k] = Y.v[k] = 0.;
    const int tend = G.lastB + 1; // the last step: B on the last row
    if (G.s == 0) {
        const int row = G.firstA;
        const long nV = (long)(2 * row) * M.nn + 2 * M.ix, ts = tile_off(M.ix, row, M.ntx, 8);
        fetch_nodes_p(u_old, nV, f.ub);
        fetch_nodes_p(v_old, nV, f.vb);
        fetch_nodes_p(u_old, nV + M.nn, f.um);
        fetch_nodes_p(v_old, nV + M.nn, f.vm);
        fetch_nodes_p(u_old, nV + 2 * M.nn, f.ut);
        fetch_nodes_p(v_old, nV + 2 * M.nn, f.vt);
#ifdef NSDG_P8_TIMING
        {
            double* st = lds + 3 * P8_HSLOTS * P8_SLOT + P8_PRING * P2P_PSLOT;
            glds_stress(S.i11, ts, st, 0);
            glds_stress(S.i12, ts, st, 1);
            glds_stress(S.i22, ts, st, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i11, ts, f.s11);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i12, ts, f.s12);
        tile_load8_p<(NSDG_P2P_NT & 1) != 0>(S.i22, ts, f.s22);
#endif
        tile_load9_p<(NSDG_P2P_NT & 4) != 0>(pg, tile_off(M.ix, row, M.ntx, 9), M.ix & 63, f.P);
        request_c_p(M, row, X.c, packed);
        for (int t = G.firstA; t <= tend; t += 2) {
            p8_step<true>(M, G, t, X, Y, f, ca, cb, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new);
            if (t + 1 <= tend)
                p8_step<true>(M, G, t + 1, Y, X, f, ca, cb, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new);
        }
    } else {
        request_c_p(M, G.firstA, X.c, packed);
        for (int t = G.firstA; t <= tend; t += 2) {
            p8_step<false>(M, G, t, X, Y, f, ca, cb, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new);
            if (t + 1 <= tend)
                p8_step<false>(M, G, t + 1, Y, X, f, ca, cb, lds, flags, rep, S, u_old, v_old, packed, pg, u_new, v_new);
        }
    }
}

} // namespace nsdg_mevp_detail

using namespace nsdg_mevp_detail;

// a pass of eight sub-iterations on the rows [j0, j1) and, if j0b < j1b, on a second disjoint range
int nsdg_launch_mevp_fused8_ranges(nsdg_ctx* ctx, int j0, int j1, int j0b, int j1b, const double* s11i, const double* s12i, const double* s22i, double* s11,
    double* s12, double* s22, const double* u_old, const double* v_old, double* u_new, double* v_new, const double* packed, const double* pg)
{
    const int ncw = nsdg_div_up(ctx->nx, P8_OWNED);
    const int rowsB = j0b < j1b ? j1b - j0b : 0;
    int R = ctx->strip_rows;
    if (R <= 0) {
        // a strip of R rows takes about R + 23 march steps: the loader's A runs on R + 15 rows, the last wave follows six steps behind it and B
        // drains one step after A; one resident workgroup per CU (LDS)
        const double extra = 23.;
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
    const nsdg_mevp_params& P = ctx->mevp;
    const NodalConsts K = { P.rho_ice * P.beta / ctx->pack_dt, P.rho_ice * (1. + P.beta) / ctx->pack_dt, P.rho_ice * P.fc };
    hipLaunchKernelGGL(mevp_fused8_kernel, dim3(ngroups), dim3(256), 0, ctx->stream, K, ctx->nx, ctx->ny, j0, j1, j0b, j1b, nsA, R, ncw, ctx->hx, ctx->hy,
        1. / ctx->mevp.alpha, ctx->mevp.delta_min * ctx->mevp.delta_min, P2PReport { ctx->p2p_count_dev, ctx->p2p_flag_dev }, S, u_old, v_old, packed, pg, u_new,
        v_new);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}
