// Microbenchmark 5: dependent vs independent VALU issue for ONE wave (inline asm, no compiler scheduling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 4096
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, uint64_t *cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        if (MODE == 0) {          // 8 dependent v_add_f64
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                         "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(x0) : "v"(b));
        } else if (MODE == 1) {   // 8 v_add_f64, 4 independent chains interleaved
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
        } else if (MODE == 2) {   // 8 dependent v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n"
                         "v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n" : "+v"(x0) : "v"(b));
        } else if (MODE == 3) {   // 8 dependent v_add_u32
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(i0) : "v"(i1));
        } else if (MODE == 4) {   // 8 independent v_add_u32 (2 chains)
            asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n"
                         "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n" : "+v"(i0), "+v"(i1) : "v"(7));
        } else if (MODE == 5) {   // cmp -> cndmask pairs, dependent: 4 x (v_cmp_lt_f64 vcc; v_cndmask lo; v_cndmask hi)
            asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n"
                         "v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n"
                         : "+v"(i0) : "v"(x0), "v"(x1), "v"(i1) : "vcc");
        } else if (MODE == 6) {   // 8 dependent v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n"
                         "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" : "+v"(x0) : "v"(b));
        } else if (MODE == 7) {   // 8 v_min_f64 dependent
            asm volatile("v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n"
                         "v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n v_min_f64 %0, %0, %1\n" : "+v"(x0) : "v"(b));
        } else if (MODE == 8) {   // 8 s_add_u32 dependent (SALU)
            int s = it;
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n"
                         "s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n" : "+s"(s) :: "scc");
            i0 += s;
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + i0 + i1;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name) {
    double *out; uint64_t *cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 1.0000001); (void)hipDeviceSynchronize(); }
    uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // the same with one wave on EVERY SIMD (1024 blocks), timed by wall clock: ns per instruction, independent of the counter's unit
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int nb : {1024, 2048}) {
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a, 0);
            for (int r2 = 0; r2 < 20; ++r2) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0, 1.0000001);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-52s %d blocks: %.2f ns per instruction per wave (20 launches)\n", name, nb, best * 1e6 / 20 / REP / 8);
    }
    printf("%-52s %.2f counter ticks per instruction, one wave on the chip\n", name, (double)c / REP / 8);
}
int main() {
    run<0>("v_add_f64 dependent"); run<1>("v_add_f64 4 independent chains"); run<2>("v_mul_f64 dependent");
    run<6>("v_fma_f64 dependent"); run<7>("v_min_f64 dependent");
    run<3>("v_add_u32 dependent"); run<4>("v_add_u32 2 independent chains"); run<5>("v_cmp_lt_f64 -> v_cndmask_b32 dependent (per instr)");
    run<8>("s_add_u32 dependent");
    // wall-clock calibration of the cycle counter
    return 0;
}
