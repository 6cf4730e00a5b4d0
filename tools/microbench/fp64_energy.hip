// fp64_energy.hip -- energy per fp64 flop of the vector pipe (v_fma_f64) against the matrix pipe (v_mfma_f64_16x16x4_f64) at the
// rate each sustains on the whole chip.  The mEVP sub-cycle runs at the 1400 W socket cap, so "would MFMA help?" is a question
// about energy per flop, not about peak rate (both peaks are 78.6 TFLOP/s on MI355X).  Each kernel runs for `seconds` on every
// CU (4, 8 or 16 waves per CU) while tools/fp64_energy.sh samples rocm-smi; the program prints the flop rate it achieved.
// build: hipcc -O2 --offload-arch=gfx950 fp64_energy.hip -o fp64_energy      usage: fp64_energy {fma|mfma|idle} seconds waves_per_cu
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int INNER = 4096; // instructions per accumulator chain and launch

__global__ __launch_bounds__(1024) void k_fma(double* out, double x, double y)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = threadIdx.x + i;
    for (int it = 0; it < INNER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            a[i] = __builtin_fma(a[i], x, y); // 8 independent chains: the pipe is never starved by latency
    }
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_mfma(double* out, double x, double y)
{
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        acc[i] = d4 { 0., 0., 0., 0. };
    const double a = x + 1e-9 * threadIdx.x, b = y;
    for (int it = 0; it < INNER / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0); // 4 independent accumulators
    }
    d4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "fma";
    const double seconds = argc > 2 ? atof(argv[2]) : 8.;
    const int waves = argc > 3 ? atoi(argv[3]) : 8;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, threads = 64 * waves;
    double* out;
    hipMalloc((void**)&out, (size_t)cus * threads * 8);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    if (!strcmp(mode, "idle")) {
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        }
    } else {
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int k = 0; k < 20; ++k, ++launches) {
                if (!strcmp(mode, "fma"))
                    hipLaunchKernelGGL(k_fma, dim3(cus), dim3(threads), 0, 0, out, 0.999999, 1e-7);
                else
                    hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(threads), 0, 0, out, 0.999999, 1e-7);
            }
            hipDeviceSynchronize();
        }
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // flops per launch: fma: 2 per lane and instruction; mfma 16x16x4: 2 * 16 * 16 * 4 per wave and instruction
    const double per_launch = !strcmp(mode, "fma") ? 2.0 * 8 * INNER * (double)cus * threads : 2048.0 * INNER * (double)cus * waves;
    printf("%s: %d CUs x %d waves, %.2f s, %ld launches, %.2f TFLOP/s fp64\n", mode, cus, waves, el, launches, launches * per_launch / el / 1e12);
    return 0;
}
