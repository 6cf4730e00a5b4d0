// vmem_overlap.hip -- what does a vector-memory instruction cost the wave that issues it, and can the wave's own fp64
// arithmetic run in its shadow?  (Round 4: the four-wave mEVP pipeline is bounded by its loader wave; its 41 loads per
// march step cost ~75 cycles each when issued in bursts.)  One wave per SIMD (or one per CU), cache-resident data:
//   burst      : 12 x global_load_dwordx4, then NF dependent-free fp64 FMAs on the values of the previous batch
//   interleaved: one load, NF/12 FMAs, one load, ... (pinned with sched_group_barrier)
//   fma only   : the FMAs alone;  loads only: the loads alone
//   saddr      : loads addressed as scalar base + 32-bit lane offset instead of a 64-bit address per lane
//   lane 0 only: the same loads with 63 lanes masked off
// build: hipcc -O2 --offload-arch=gfx950 vmem_overlap.hip -o vmem_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NL = 12; // loads per batch
constexpr int BATCH = 256;
constexpr int FPL = 20; // FMAs per load

typedef double d2 __attribute__((ext_vector_type(2)));
enum Mode { BURST = 0, INTERLEAVED = 1, FMA_ONLY = 2, LOAD_ONLY = 3, SADDR = 4, LANE0 = 5 };

__device__ __forceinline__ d2 ld(const double* q)
{
    d2 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(q));
    return r;
}
__device__ __forceinline__ d2 ld_saddr(const double* base, unsigned off)
{
    d2 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base));
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, double* out, long long* cyc)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const double* p = src + (wave & 1023) * (long)NL * 128;
    const unsigned long pa = (unsigned long)p;
    const double* pu = (const double*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pa));
    double acc[8] = { 1., 2., 3., 4., 5., 6., 7., 8. };
    d2 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
        v[i] = d2{1e-3 * i, 1e-4 * lane};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < BATCH; ++b) {
        d2 w[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i)
            w[i] = d2{v[i].y, v[i].x};
        if (MODE == BURST || MODE == LOAD_ONLY || MODE == LANE0) {
            if (MODE != LANE0 || lane == 0) {
#pragma unroll
                for (int i = 0; i < NL; ++i)
                    w[i] = ld(p + i * 128 + lane * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (MODE == INTERLEAVED)
                w[i] = ld(p + i * 128 + lane * 2);
            if (MODE == SADDR)
                w[i] = ld_saddr(pu + i * 128, (unsigned)lane * 16u);
            if (MODE != LOAD_ONLY && MODE != LANE0) {
#pragma unroll
                for (int f = 0; f < FPL; ++f)
                    acc[f & 7] = __builtin_fma(acc[f & 7], v[i].x, v[(i + f) % NL].y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NL; ++i)
            asm volatile("" : "+v"(w[i])); // the loaded values are "used" here
#pragma unroll
        for (int i = 0; i < NL; ++i)
            v[i] = w[i];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s += acc[i];
#pragma unroll
    for (int i = 0; i < NL; ++i)
        s += v[i].x + v[i].y;
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0)
        cyc[wave] = t1 - t0;
}

int main()
{
    const int blocks = 256;
    double *src, *out;
    long long* cyc;
    hipMalloc((void**)&src, 1024L * NL * 128 * 8);
    hipMalloc((void**)&out, blocks * 256 * 8);
    hipMalloc((void**)&cyc, blocks * 4 * 8);
    hipMemset(src, 0, 1024L * NL * 128 * 8);
    std::vector<long long> h(blocks * 4);
    for (int threads : { 256, 64 }) {
        printf("-- %d wave(s) per CU, %d loads per batch, %d FMAs per load\n", threads / 64, NL, FPL);
        auto report = [&](const char* name) {
            hipDeviceSynchronize();
            const int nw = blocks * threads / 64;
            hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < nw; ++i)
                s += h[i];
            s /= nw;
            printf("%-34s %8.1f cycles per batch = %6.1f per load slot (%d loads + %d FMAs)\n", name, s / BATCH, s / BATCH / NL, NL, NL * FPL);
        };
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<FMA_ONLY>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("fma only");
            hipLaunchKernelGGL(k<LOAD_ONLY>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("loads only (64-bit lane address)");
            hipLaunchKernelGGL(k<LANE0>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("loads only, lane 0 alone active");
            hipLaunchKernelGGL(k<BURST>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("burst of loads, then FMAs");
            hipLaunchKernelGGL(k<INTERLEAVED>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("interleaved 1 load : 20 FMAs");
            hipLaunchKernelGGL(k<SADDR>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
            report("interleaved, scalar base + offset");
        }
    }
    return 0;
}
