// vmem_issue.hip -- issue cost of global loads / stores for one wave per SIMD (4 waves per CU, every CU busy):
// does a 16-byte-per-lane access cost the issuing wave as much as an 8-byte one?  The marching mEVP kernel spends
// ~30 cycles of issue per vector-memory instruction, so halving their number by pairing coefficients would pay
// only if the cost is per instruction, not per byte.
// build: hipcc -O2 --offload-arch=gfx950 vmem_issue.hip -o vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NL = 24; // loads in flight per batch (like the 24 stress loads of a row)
constexpr int BATCH = 64;

template <int WIDTH> // doubles per lane and instruction: 1 (dwordx2) or 2 (dwordx4)
__global__ __launch_bounds__(256) void k_load(const double* __restrict__ src, double* out, long long* cyc, long stride_batch)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double* p = src + wave * (long)BATCH * NL * 64;
    double acc = 0.;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < BATCH; ++b) {
        const double* q = p + (long)b * stride_batch;
        if (WIDTH == 1) {
            double v[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i)
                v[i] = q[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < NL; ++i)
                acc += v[i];
        } else {
            double2 v[NL / 2];
#pragma unroll
            for (int i = 0; i < NL / 2; ++i)
                v[i] = reinterpret_cast<const double2*>(q + i * 128)[lane];
#pragma unroll
            for (int i = 0; i < NL / 2; ++i)
                acc += v[i].x + v[i].y;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0)
        cyc[wave] = t1 - t0;
}

// 16-byte loads with a 48-byte lane stride (the packed nodal coefficients: 6 doubles per node, node-major)
__global__ __launch_bounds__(256) void k_load_strided(const double* __restrict__ src, double* out, long long* cyc, long stride_batch)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double* p = src + wave * (long)BATCH * NL * 64;
    double acc = 0.;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < BATCH; ++b) {
        const double* q = p + (long)b * stride_batch;
        double2 v[NL / 2];
#pragma unroll
        for (int i = 0; i < NL / 2; ++i) // 12 loads = 4 nodes x 3 pieces; lanes 48 B apart
            v[i] = *reinterpret_cast<const double2*>(q + (i / 3) * 384 + lane * 6 + (i % 3) * 2);
#pragma unroll
        for (int i = 0; i < NL / 2; ++i)
            acc += v[i].x + v[i].y;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0)
        cyc[wave] = t1 - t0;
}

template <int WIDTH>
__global__ __launch_bounds__(256) void k_store(double* dst, long long* cyc, long stride_batch)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double* p = dst + wave * (long)BATCH * NL * 64;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < BATCH; ++b) {
        double* q = p + (long)b * stride_batch;
        const double x = b + lane;
        if (WIDTH == 1) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                q[i * 64 + lane] = x + i;
        } else {
#pragma unroll
            for (int i = 0; i < NL / 2; ++i)
                reinterpret_cast<double2*>(q + i * 128)[lane] = make_double2(x + i, x - i);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        cyc[wave] = t1 - t0;
}

int main()
{
    const int blocks = 256; // one workgroup of 4 waves per CU: one wave per SIMD
    const long stride = NL * 64; // doubles per batch and wave
    const long n = (long)blocks * 4 * BATCH * stride;
    double *src, *out;
    long long* cyc;
    hipMalloc((void**)&src, n * 8);
    hipMalloc((void**)&out, blocks * 256 * 8);
    hipMalloc((void**)&cyc, blocks * 4 * 8);
    hipMemset(src, 0, n * 8);
    std::vector<long long> h(blocks * 4);
    auto report = [&](const char* name, int instr) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto c : h)
            s += c;
        s /= h.size();
        printf("%-44s %8.1f cycles per instruction per wave, %6.1f B/cycle/CU\n", name, s / ((double)BATCH * instr),
            4.0 * BATCH * NL * 512.0 / s);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (long st : { stride, 0L }) { // HBM stream / the same 12 KB per wave again and again (cache hits: issue cost)
            printf(st ? "-- streaming from HBM\n" : "-- cache-resident (each wave re-reads its 12 KB)\n");
            hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, st);
            report("global_load_dwordx2 (8 B/lane), 24 per batch", NL);
            hipLaunchKernelGGL(k_load<2>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, st);
            report("global_load_dwordx4 (16 B/lane), 12 per batch", NL / 2);
            hipLaunchKernelGGL(k_load_strided, dim3(blocks), dim3(256), 0, 0, src, out, cyc, st);
            report("global_load_dwordx4, lanes 48 B apart, 12 per batch", NL / 2);
            hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, src, cyc, st);
            report("global_store_dwordx2, 24 per batch", NL);
            hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, src, cyc, st);
            report("global_store_dwordx4, 12 per batch", NL / 2);
        }
    }
    printf("(%ld MB working set: HBM-resident streams, every CU busy)\n", n * 8 >> 20);
    return 0;
}
