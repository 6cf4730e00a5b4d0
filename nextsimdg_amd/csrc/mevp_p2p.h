// mevp_p2p.h -- the point-to-point primitives of the stage-per-wave pipeline of the mEVP sub-cycle (mevp_fused4.hip; round 6 built a second
// pipeline on them, two sub-iterations per stage wave, and withdrew it: profiles/r06_fused8.md): the hand-over between the waves of a workgroup --
// counters in LDS, a bounded wait, the report of a wait that gave up --, the streaming 16-byte accesses, the rings of a row's ice
// strength / nodal coefficients in LDS.
//
// Memory ordering.  All hand-over traffic is LDS traffic of ONE compute unit; the LDS executes the instructions of a wave in order.
// A producer waits for its own LDS writes (s_waitcnt lgkmcnt(0)) before it raises its counter; a consumer reads the counter,
// waits for that read, and only then issues its reads of the slot -- the pattern of an LDS-scope release / acquire, written out
// with compiler barriers around it.  Counters only ever increase.
//
// A wait that gives up (NSDG_P2P_SPIN_LIMIT polls: never in a correct program) raises the workgroup's sticky flag, which releases
// every other wait of the workgroup, counts the event in the context's device counter and sets the context's flag in HOST memory:
// the next nsdg_ctx_synchronize / nsdg_mevp_subcycle / nsdg_rb_mevp_run on the context returns NSDG_ERR_HIP (nsdg_ctx.hip:
// nsdg_p2p_check).  A wrong result that is reported as an error, never a hung GPU.
#pragma once
#include "mevp_pipeline.h"

namespace nsdg_mevp_detail {

#ifndef NSDG_P2P_SPIN_LIMIT
#define NSDG_P2P_SPIN_LIMIT (1 << 20) // polls of ~0.2 us: a fifth of a second; a legitimate wait is a few march steps (a few microseconds)
#endif

struct StressPtrsP {
    const double *i11, *i12, *i22;
    double *o11, *o12, *o22;
};

// where a wait that gave up is reported: both belong to the context (nsdg_internal.h)
struct P2PReport {
    unsigned* count; // device memory: events since the last nsdg_mevp_pipeline_health
    unsigned* flag; // host memory mapped into the device: non-zero = at least one event
};

constexpr int P2P_GIVEUP = 6; // index of the sticky give-up flag among a workgroup's counters

// the counters are accessed through LDS-typed pointers: a volatile access through a generic pointer would be a FLAT instruction, which
// counts on vmcnt as well and so waits for every global load in flight
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int flag_peek(const volatile lds_int* p)
{
    asm volatile("" ::: "memory");
    const int x = *p;
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(x);
}
// the counter flags[which] has reached `need` (or the workgroup has given up)
__device__ __forceinline__ int flag_wait(volatile lds_int* flags, int which, int need, const P2PReport& rep) // returns the polls it took beyond the first
{
    if (flag_peek(flags + which) >= need)
        return 0;
    for (int spin = 0; spin < NSDG_P2P_SPIN_LIMIT; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (flag_peek(flags + which) >= need || flag_peek(flags + P2P_GIVEUP) != 0)
            return spin + 1;
    }
    flags[P2P_GIVEUP] = 1; // give up: release everybody, report the event
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(rep.count, 1u);
        __hip_atomic_store(rep.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return NSDG_P2P_SPIN_LIMIT;
}
// everything this wave has written to (or read from) LDS so far is complete before the counter moves
__device__ __forceinline__ void flag_publish(volatile lds_int* flags, int which, int value)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    flags[which] = value;
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ double2 lds_pair_p(const double* slot, int k) { return *reinterpret_cast<const double2*>(slot + k * 128); }
__device__ __forceinline__ void lds_pair_p(double* slot, int k, double a, double b) { *reinterpret_cast<double2*>(slot + k * 128) = make_double2(a, b); }

// Streaming accesses (NSDG_P2P_NT bits: 1 the loader's stress loads, 2 the last stage's stress stores, 4 the loader's ice-strength
// loads): data a pass touches exactly once need not displace the coefficient rows the later stages re-read through the L2.  Measured
// on one box, three alternations (profiles/r05_fused4_p2p.md section 3): 0 0.8727-0.8741 ms per pass at 2048^2, 2 0.8609-0.8679,
// 3 0.8619-0.8702, 7 0.8641-0.8708 -- the same bits in memory, about 1 % less time: 3 is the default.
#ifndef NSDG_P2P_NT
#define NSDG_P2P_NT 3
#endif
typedef double nsdg_pair16p __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void tile_load8_p(const double* __restrict__ a, long t, double (&c)[8])
{
    if (!NT)
        return tile_load8(a, t, c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const nsdg_pair16p v = __builtin_nontemporal_load(reinterpret_cast<const nsdg_pair16p*>(a + t + 128 * k));
        c[2 * k] = v.x, c[2 * k + 1] = v.y;
    }
}
template <bool NT>
__device__ __forceinline__ void tile_load9_p(const double* __restrict__ a, long t, int l, double (&c)[9])
{
    if (!NT)
        return tile_load9(a, t, l, c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const nsdg_pair16p v = __builtin_nontemporal_load(reinterpret_cast<const nsdg_pair16p*>(a + t + 128 * k));
        c[2 * k] = v.x, c[2 * k + 1] = v.y;
    }
    c[8] = __builtin_nontemporal_load(a + t + 512 - l);
}
template <bool NT>
__device__ __forceinline__ void tile_store8_p(double* __restrict__ a, long t, const double (&c)[8])
{
    if (!NT)
        return tile_store8(a, t, c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        nsdg_pair16p v;
        v.x = c[2 * k], v.y = c[2 * k + 1];
        __builtin_nontemporal_store(v, reinterpret_cast<nsdg_pair16p*>(a + t + 128 * k));
    }
}
typedef double nsdg_pair8p __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void fetch_nodes_p(const double* __restrict__ w, long n, double (&o)[3])
{
    const nsdg_pair8p a = *reinterpret_cast<const nsdg_pair8p*>(w + n);
    o[0] = a.x, o[1] = a.y, o[2] = w[n + 2];
}

// all three pairs of the packed coefficients of the 4 owned nodes of element row nrow
__device__ __forceinline__ void request_c_p(const MarchConst3& M, int nrow, double (&c)[4][6], const double* __restrict__ packed)
{
    const long nVn = (long)(2 * nrow) * M.nn + 2 * M.ix;
    load_nodal(packed, M.nplane, nVn, c[0]);
    load_nodal(packed, M.nplane, nVn + 1, c[1]);
    load_nodal(packed, M.nplane, nVn + M.nn, c[2]);
    load_nodal(packed, M.nplane, nVn + M.nn + 1, c[3]);
}

// Rings of NR rows in LDS.  Ice strength: a row takes 9 * 64 doubles, pairs k < 4 of lane l at k * 128 + 2 l, the ninth value at 512 + l.
constexpr int P2P_PSLOT = 9 * 64;
template <int NR>
__device__ __forceinline__ void ring_read_P(const double* __restrict__ ring, int row, int lane, double (&P)[9]) // row >= 0
{
    const double* s = ring + (row % NR) * P2P_PSLOT;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double2 t = *reinterpret_cast<const double2*>(s + k * 128 + 2 * lane);
        P[2 * k] = t.x, P[2 * k + 1] = t.y;
    }
    P[8] = s[512 + lane];
}
template <int NR>
__device__ __forceinline__ void ring_write_P(double* __restrict__ ring, int row, int lane, const double (&P)[9])
{
    double* s = ring + (row % NR) * P2P_PSLOT;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<double2*>(s + k * 128 + 2 * lane) = make_double2(P[2 * k], P[2 * k + 1]);
    s[512 + lane] = P[8];
}

} // namespace nsdg_mevp_detail
