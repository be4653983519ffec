// Microbenchmark 6: what do the SQ wait/active counters read for a lone wave of pure back-to-back ALU work?
// Calibrates the reading of SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY for k_explore_slice8 (profiles/r01_slice8_summary.txt).
// Long launches (~10 ms) so the clock has ramped; one wave per SIMD (1024 blocks) and two (2048).
//   rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- ./issue_pmc.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP (1 << 18)
template <int MODE>
__global__ __launch_bounds__(64) void k_issue(double *out, double a, double b) {
    double x0 = a + threadIdx.x;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3;
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        if (MODE == 0) {
            asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n"
                         "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" : "+v"(x0) : "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(i0) : "v"(i1));
        } else {
            int s = it;
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n"
                         "s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n" : "+s"(s) :: "scc");
            i0 += s;
        }
    }
    out[threadIdx.x] = x0 + i0 + i1;
}
template <int MODE> void run(const char *name) {
    double *out; (void)hipMalloc(&out, 64 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int nb : {1024, 2048}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_issue<MODE>, dim3(nb), dim3(64), 0, 0, out, 1.0, 1.0000001);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-24s %d blocks: %.2f ns per instruction per wave (launch %.2f ms)\n", name, nb, best * 1e6 / REP / 8, best);
    }
}
int main() {
    run<0>("v_fma_f64 dependent"); run<1>("v_add_u32 dependent"); run<2>("s_add_u32 dependent");
    return 0;
}
