// Microbenchmark 7 (round 3): the issue floor of ONE wave per SIMD, and what the SQ counters read at that floor.
// Settles DESIGN.md 5's two readings (4.2 cycles per FP64 instruction by the cycle counter vs 3.2-3.8 ns by wall clock):
// every mode is its own kernel (own PMC row), launches last several ms, 1024 blocks x 64 lanes = one wave on every SIMD.
//   ./issue_floor.bin                     wall clock + s_memtime per instruction
//   rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -- ./issue_floor.bin
//   rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -- ./issue_floor.bin         (real clock = GRBM_GUI_ACTIVE / kernel time)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP (1 << 16)
#define BODY 64                          // instructions per loop iteration (>= 4k instructions per 64 iterations)

#define R8(X) X X X X X X X X
// 8 independent FP64 chains, 64 instructions per iteration
__global__ __launch_bounds__(64) void k_f64_indep(double *out, uint64_t *cyc, double b) {
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                        "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// one dependent FP64 chain
__global__ __launch_bounds__(64) void k_f64_dep(double *out, uint64_t *cyc, double b) {
    double x0 = threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                        "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n") : "+v"(x0) : "v"(b));
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 8 independent 32-bit VALU chains
__global__ __launch_bounds__(64) void k_u32_indep(double *out, uint64_t *cyc, double b) {
    int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7; const int c = (int)b;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// one dependent 32-bit VALU chain
__global__ __launch_bounds__(64) void k_u32_dep(double *out, uint64_t *cyc, double b) {
    int x0 = threadIdx.x; const int c = (int)b;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                        "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n") : "+v"(x0) : "v"(c));
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// VALU (FP64, independent) and SALU alternating: does a SALU instruction take a VALU issue slot of a lone wave?
__global__ __launch_bounds__(64) void k_f64_salu_mix(double *out, uint64_t *cyc, double b) {
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3; int s = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_add_f64 %0, %0, %5\n s_add_u32 %4, %4, 3\n v_add_f64 %1, %1, %5\n s_add_u32 %4, %4, 3\n"
                        "v_add_f64 %2, %2, %5\n s_add_u32 %4, %4, 3\n v_add_f64 %3, %3, %5\n s_add_u32 %4, %4, 3\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(b) : "scc");
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// the slice kernel's instruction mix in miniature: FP64 op -> compare -> 2 selects (dependent), 32-bit add alongside
__global__ __launch_bounds__(64) void k_slice_mix(double *out, uint64_t *cyc, double b) {
    double x0 = threadIdx.x, x1 = 1.5; int n = 0, lo = threadIdx.x, hi = 7;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it)
        asm volatile(R8("v_mul_f64 %1, %0, %0\n v_add_f64 %1, %1, -%5\n v_cmp_lt_f64 vcc, %1, %5\n v_add_u32 %2, 1, %2\n"
                        "v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %4, %4, %2, vcc\n v_min_f64 %1, %1, |%0|\n v_add_f64 %0, %0, %5\n")
                     : "+v"(x0), "+v"(x1), "+v"(n), "+v"(lo), "+v"(hi) : "v"(b) : "vcc");
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + n + lo + hi;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K> void run(const char *name, K kern, int per_iter = BODY) {
    double *out; uint64_t *cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int nb : {1024, 2048}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0000001);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)REP * per_iter;
        printf("%-16s %4d waves: %6.3f ms  %5.2f ns/instr/wave  %5.2f s_memtime ticks/instr  => %.0f MHz if a tick is a shader cycle\n",
               name, nb, best, best * 1e6 / n, (double)c / n, (double)c / (best * 1e3));
    }
}
int main() {
    run("k_f64_indep", k_f64_indep); run("k_f64_dep", k_f64_dep); run("k_u32_indep", k_u32_indep); run("k_u32_dep", k_u32_dep);
    run("k_f64_salu_mix", k_f64_salu_mix); run("k_slice_mix", k_slice_mix);
    return 0;
}
