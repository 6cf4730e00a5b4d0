// seed_accuracy.hip -- relative error of the hardware seeds v_rsq_f64 / v_rcp_f64 on gfx950 and of the refinement
// schemes built on them (two Newton steps vs one third-order step), against long-double references on the host.
// build: hipcc -O2 --offload-arch=gfx950 seed_accuracy.hip -o seed_accuracy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double* x, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double a = x[i];
    // seeds
    const double y0 = __builtin_amdgcn_rsq(a), r0 = __builtin_amdgcn_rcp(a);
    // rsqrt: two Newton steps (the round-1 form)
    double y = y0;
    double e = __builtin_fma(-a * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-a * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    // rsqrt: one third-order step  y (1 + e/2 + 3 e^2 / 8)
    double z = y0;
    const double f = __builtin_fma(-a * z, z, 1.0);
    z = __builtin_fma(z * f, __builtin_fma(0.375, f, 0.5), z);
    // rcp: two Newton steps / one second-order step r (1 + e + e^2)
    double r = r0;
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    double q = r0;
    const double g = __builtin_fma(-a, q, 1.0);
    q = __builtin_fma(q * g, 1.0 + g, q);
    out[i] = y0, out[n + i] = r0, out[2 * n + i] = y, out[3 * n + i] = z, out[4 * n + i] = r, out[5 * n + i] = q;
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), o(6 * n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> ex(-60., 60.), m(1., 2.);
    for (auto& v : x)
        v = std::ldexp(m(rng), (int)ex(rng));
    double *dx, *dout;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* names[6] = { "v_rsq_f64 seed", "v_rcp_f64 seed", "rsqrt, two Newton steps", "rsqrt, one third-order step", "rcp, two Newton steps",
        "rcp, one second-order step" };
    for (int k = 0; k < 6; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = (k == 0 || k == 2 || k == 3) ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
            const long double err = fabsl(((long double)o[(size_t)k * n + i] - ref) / ref);
            if (err > worst)
                worst = err;
        }
        std::printf("%-30s max relative error %.3Le = 2^%.2Lf  (%.2Lf ulp of fp64)\n", names[k], worst, log2l(worst), worst / 1.1102230246251565e-16L);
    }
    return 0;
}
