// fp64_issue.hip -- issue cost of the vector instructions the mEVP marching kernel is made of, for ONE wave
// on a SIMD (the kernel runs at one wave per SIMD, so nothing hides dependent-issue latency).
// Prints shader cycles (s_memtime) per instruction for independent streams and dependent chains.
// build: hipcc -O2 --offload-arch=gfx950 fp64_issue.hip -o fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

constexpr int LOOPS = 200;

#define BENCH_KERNEL(name, body, ninst)                                               \
    __global__ void name(double* out, long long* cyc, double b, double c)             \
    {                                                                                 \
        double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;       \
        double a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                   \
        double vb = b + threadIdx.x, vc = c + threadIdx.x;                            \
        __syncthreads();                                                              \
        long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int i = 0; i < LOOPS; ++i) {                                             \
            asm volatile(body                                                         \
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                : "v"(vb), "v"(vc), "s"(b)                                            \
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");                                           \
        }                                                                             \
        long long t1 = __builtin_amdgcn_s_memtime();                                  \
        out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                     \
        if ((threadIdx.x & 63) == 0)                                                  \
            cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;   \
        if (threadIdx.x == 0)                                                         \
            cyc[64] = (long long)(ninst)*LOOPS;                                       \
    }

// 8 independent accumulators
BENCH_KERNEL(k_fma_indep8, REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                 "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"), 128)
BENCH_KERNEL(k_fma_indep4, REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"), 64)
BENCH_KERNEL(k_fma_indep2, REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"), 32)
BENCH_KERNEL(k_fma_dep, REP64("v_fma_f64 %0, %0, %8, %9\n"), 64)
BENCH_KERNEL(k_fmac_indep8, REP16("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                                  "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9\n"), 128)
BENCH_KERNEL(k_fmac_dep, REP64("v_fmac_f64 %0, %8, %9\n"), 64)
BENCH_KERNEL(k_mul_indep8, REP16("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                                 "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"), 128)
BENCH_KERNEL(k_mul_dep, REP64("v_mul_f64 %0, %0, %8\n"), 64)
BENCH_KERNEL(k_add_indep8, REP16("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                                 "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"), 128)
BENCH_KERNEL(k_add_dep, REP64("v_add_f64 %0, %0, %8\n"), 64)
BENCH_KERNEL(k_fma_sgpr_indep8, REP16("v_fma_f64 %0, %0, %10, %9\n v_fma_f64 %1, %1, %10, %9\n v_fma_f64 %2, %2, %10, %9\n v_fma_f64 %3, %3, %10, %9\n"
                                      "v_fma_f64 %4, %4, %10, %9\n v_fma_f64 %5, %5, %10, %9\n v_fma_f64 %6, %6, %10, %9\n v_fma_f64 %7, %7, %10, %9\n"), 128)
BENCH_KERNEL(k_mul_lit_indep8, REP16("v_mul_f64 %0, %0, 0.5\n v_mul_f64 %1, %1, 0.5\n v_mul_f64 %2, %2, 0.5\n v_mul_f64 %3, %3, 0.5\n"
                                     "v_mul_f64 %4, %4, 0.5\n v_mul_f64 %5, %5, 0.5\n v_mul_f64 %6, %6, 0.5\n v_mul_f64 %7, %7, 0.5\n"), 128)
BENCH_KERNEL(k_rsq_indep8, REP16("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                                 "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"), 128)
BENCH_KERNEL(k_rcp_indep8, REP16("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                                 "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"), 128)
BENCH_KERNEL(k_rsq_dep, REP64("v_rsq_f64 %0, %0\n"), 64)
// one rsq followed by 7 independent fmas: does the transcendental overlap with the FMA pipe?
BENCH_KERNEL(k_rsq_plus_7fma, REP16("v_rsq_f64 %0, %0\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                    "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"), 128)
BENCH_KERNEL(k_mov_b64_indep8, REP16("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                                     "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"), 128)
BENCH_KERNEL(k_acc_write_read, REP16("v_accvgpr_write_b32 a0, v100\n v_accvgpr_write_b32 a1, v101\n v_accvgpr_write_b32 a2, v102\n v_accvgpr_write_b32 a3, v103\n"
                                     "v_accvgpr_read_b32 v104, a4\n v_accvgpr_read_b32 v105, a5\n v_accvgpr_read_b32 v106, a6\n v_accvgpr_read_b32 v107, a7\n"), 128)
// a double parked in two AGPRs, read back and fed to an fma directly
BENCH_KERNEL(k_acc_read_then_fma, REP16("v_accvgpr_read_b32 v100, a0\n v_accvgpr_read_b32 v101, a1\n v_fma_f64 %1, v[100:101], %8, %9\n"
                                        "v_accvgpr_read_b32 v102, a2\n v_accvgpr_read_b32 v103, a3\n v_fma_f64 %3, v[102:103], %8, %9\n"
                                        "v_accvgpr_read_b32 v104, a4\n v_accvgpr_read_b32 v105, a5\n v_fma_f64 %5, v[104:105], %8, %9\n"
                                        "v_accvgpr_read_b32 v106, a6\n v_accvgpr_read_b32 v107, a7\n v_fma_f64 %7, v[106:107], %8, %9\n"), 192)
BENCH_KERNEL(k_dpp_mov, REP16("v_mov_b32_dpp v100, v100 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp v101, v101 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                              "v_mov_b32_dpp v102, v102 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp v103, v103 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"), 64)
// long straight-line body (4096 x 8-byte instructions = 32 KB): instruction fetch of one wave
#define REP256(x) REP4(REP64(x))
BENCH_KERNEL(k_fma_long_body, REP256(REP4("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n ")), 8192)
// VGPR bank pressure: three 64-bit sources from the same register-file banks vs spread
BENCH_KERNEL(k_fma_same_bank, REP16("v_fma_f64 v[100:101], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[116:117], v[120:121], v[124:125], v[128:129]\n"
                                    "v_fma_f64 v[132:133], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[136:137], v[120:121], v[124:125], v[128:129]\n"), 64)
BENCH_KERNEL(k_fma_spread_bank, REP16("v_fma_f64 v[100:101], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[116:117], v[120:121], v[126:127], v[128:129]\n"
                                      "v_fma_f64 v[132:133], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[136:137], v[120:121], v[126:127], v[128:129]\n"), 64)
// fp32 reference points
BENCH_KERNEL(k_f32_fma_indep8, REP16("v_fma_f32 v100, v100, v108, v109\n v_fma_f32 v101, v101, v108, v109\n v_fma_f32 v102, v102, v108, v109\n v_fma_f32 v103, v103, v108, v109\n v_fma_f32 v104, v104, v108, v109\n v_fma_f32 v105, v105, v108, v109\n v_fma_f32 v106, v106, v108, v109\n v_fma_f32 v107, v107, v108, v109\n "), 128)
BENCH_KERNEL(k_f32_fma_dep, REP64("v_fma_f32 v100, v100, v108, v109\n"), 64)

typedef void (*kern_t)(double*, long long*, double, double);

int main()
{
    double* out;
    long long* cyc;
    hipMalloc((void**)&out, 1024 * 8);
    hipMalloc((void**)&cyc, 65 * 8);
    struct T { const char* name; kern_t k; };
    std::vector<T> tests = { { "v_fma_f64 8 independent", k_fma_indep8 }, { "v_fma_f64 4 independent", k_fma_indep4 },
        { "v_fma_f64 2 independent", k_fma_indep2 }, { "v_fma_f64 dependent chain", k_fma_dep },
        { "v_fmac_f64 8 independent", k_fmac_indep8 }, { "v_fmac_f64 dependent chain", k_fmac_dep },
        { "v_mul_f64 8 independent", k_mul_indep8 }, { "v_mul_f64 dependent chain", k_mul_dep },
        { "v_add_f64 8 independent", k_add_indep8 }, { "v_add_f64 dependent chain", k_add_dep },
        { "v_fma_f64 SGPR operand, 8 indep", k_fma_sgpr_indep8 }, { "v_mul_f64 inline const, 8 indep", k_mul_lit_indep8 },
        { "v_rsq_f64 8 independent", k_rsq_indep8 }, { "v_rcp_f64 8 independent", k_rcp_indep8 }, { "v_rsq_f64 dependent chain", k_rsq_dep },
        { "1 v_rsq_f64 + 7 v_fma_f64 (per instr)", k_rsq_plus_7fma }, { "v_mov_b64 8 independent", k_mov_b64_indep8 },
        { "v_accvgpr write x4 + read x4", k_acc_write_read }, { "2 v_accvgpr_read + v_fma_f64 (per instr)", k_acc_read_then_fma },
        { "v_mov_b32_dpp wave_shr (+s_nop 1)", k_dpp_mov }, { "v_fma_f64 straight-line 32 KB body", k_fma_long_body }, { "v_fma_f64 sources in the same banks", k_fma_same_bank }, { "v_fma_f64 sources in spread banks", k_fma_spread_bank }, { "v_fma_f32 8 independent", k_f32_fma_indep8 },
        { "v_fma_f32 dependent chain", k_f32_fma_dep } };
    for (int waves = 1; waves <= 4; waves *= 2) {
        printf("--- %d wave(s) per SIMD (one workgroup of %d threads); cycles per instruction per wave, span of the workgroup / instructions of one wave\n", waves, waves * 256);
        for (auto& t : tests) {
            long long h[65];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(t.k, dim3(1), dim3(waves * 256), 0, 0, out, cyc, 1.0000001, 1e-9);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            long long lo = h[0], hi = h[1];
            for (int w = 0; w < waves * 4; ++w)
                lo = h[2 * w] < lo ? h[2 * w] : lo, hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi;
            printf("%-42s %7.2f   %7.2f\n", t.name, (double)(h[1] - h[0]) / (double)h[64], (double)(hi - lo) / (double)h[64]);
        }
    }
    return 0;
}
