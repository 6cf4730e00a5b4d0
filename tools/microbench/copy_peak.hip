// copy_peak.hip -- which device copy is the fastest on this box?  (the "measured device-copy peak" of bench.py must not be a
// badly tuned kernel: the roofline fraction quoted against it would flatter the product)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/copy_peak.hip -o tools/microbench/copy_peak && tools/microbench/copy_peak
// Variants: 16-byte accesses with U of them in flight per lane (one workgroup per 256*U*16 bytes, no loop), the same with
// non-temporal loads and stores, a grid-stride loop with a few workgroups per CU, and hipMemcpyAsync device-to-device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_block(double2* __restrict__ dst, const double2* __restrict__ src, long n2)
{
    const long i0 = (long)blockIdx.x * (256 * U) + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k)
        if (i0 + 256 * k < n2) {
            if (NT) {
                v[k].x = __builtin_nontemporal_load(&src[i0 + 256 * k].x);
                v[k].y = __builtin_nontemporal_load(&src[i0 + 256 * k].y);
            } else
                v[k] = src[i0 + 256 * k];
        }
#pragma unroll
    for (int k = 0; k < U; ++k)
        if (i0 + 256 * k < n2) {
            if (NT) {
                __builtin_nontemporal_store(v[k].x, &dst[i0 + 256 * k].x);
                __builtin_nontemporal_store(v[k].y, &dst[i0 + 256 * k].y);
            } else
                dst[i0 + 256 * k] = v[k];
        }
}

template <int U>
__global__ __launch_bounds__(256) void copy_stride(double2* __restrict__ dst, const double2* __restrict__ src, long n2)
{
    const long stride = (long)gridDim.x * 256 * U;
    for (long i0 = (long)blockIdx.x * (256 * U) + threadIdx.x; i0 < n2; i0 += stride) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (i0 + 256 * k < n2)
                v[k] = src[i0 + 256 * k];
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (i0 + 256 * k < n2)
                dst[i0 + 256 * k] = v[k];
    }
}

template <class F>
double time_ms(F&& launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; ++i)
        launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a), hipEventDestroy(b);
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (long mib : { 512L, 1024L, 4096L }) {
        const long n2 = mib * (1L << 20) / 16;
        double2 *src, *dst;
        CHECK(hipMalloc(&src, n2 * 16));
        CHECK(hipMalloc(&dst, n2 * 16));
        CHECK(hipMemset(src, 1, n2 * 16));
        const double gb = 2.0 * n2 * 16 / 1e9;
        auto report = [&](const char* name, double ms) { std::printf("%5ld MiB  %-44s %8.3f ms  %8.1f GB/s (read + write)\n", mib, name, ms, gb / ms * 1e3); };
        report("16 B x 1 in flight, block per 4 KB", time_ms([&] { hipLaunchKernelGGL((copy_block<1, false>), dim3((n2 + 255) / 256), dim3(256), 0, 0, dst, src, n2); }, 10));
        report("16 B x 2 in flight", time_ms([&] { hipLaunchKernelGGL((copy_block<2, false>), dim3((n2 + 511) / 512), dim3(256), 0, 0, dst, src, n2); }, 10));
        report("16 B x 4 in flight (nsdg_copy_f64)", time_ms([&] { hipLaunchKernelGGL((copy_block<4, false>), dim3((n2 + 1023) / 1024), dim3(256), 0, 0, dst, src, n2); }, 10));
        report("16 B x 8 in flight", time_ms([&] { hipLaunchKernelGGL((copy_block<8, false>), dim3((n2 + 2047) / 2048), dim3(256), 0, 0, dst, src, n2); }, 10));
        report("16 B x 4, non-temporal loads and stores", time_ms([&] { hipLaunchKernelGGL((copy_block<4, true>), dim3((n2 + 1023) / 1024), dim3(256), 0, 0, dst, src, n2); }, 10));
        report("16 B x 8, non-temporal", time_ms([&] { hipLaunchKernelGGL((copy_block<8, true>), dim3((n2 + 2047) / 2048), dim3(256), 0, 0, dst, src, n2); }, 10));
        for (int per_cu : { 4, 8, 16 }) {
            char name[64];
            std::snprintf(name, sizeof name, "grid-stride, 16 B x 4, %d workgroups per CU", per_cu);
            report(name, time_ms([&] { hipLaunchKernelGGL((copy_stride<4>), dim3(per_cu * cus), dim3(256), 0, 0, dst, src, n2); }, 10));
        }
        report("hipMemcpyAsync device to device", time_ms([&] { (void)hipMemcpyAsync(dst, src, n2 * 16, hipMemcpyDeviceToDevice, 0); }, 10));
        CHECK(hipFree(src));
        CHECK(hipFree(dst));
    }
    return 0;
}
