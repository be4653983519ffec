// Microbenchmark: how fast can ONE wavefront issue instructions on gfx950?
// Prints cycles per instruction for dependent / independent FP64 adds, SALU chains, readlane, branches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, uint64_t *cyc, int iters, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4;
    int s0 = (int)a, s1 = 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // dependent v_add_f64
#pragma unroll
            for (int i = 0; i < REP; ++i) x0 = x0 + b;
        } else if (MODE == 1) { // 4 independent chains
#pragma unroll
            for (int i = 0; i < REP / 4; ++i) { x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; }
        } else if (MODE == 2) { // dependent SALU
#pragma unroll
            for (int i = 0; i < REP; ++i) { s0 = __builtin_amdgcn_readfirstlane(s0) * 3 + s1; }
        } else if (MODE == 3) { // dependent v_mul_f64
#pragma unroll
            for (int i = 0; i < REP; ++i) x0 = x0 * b;
        } else if (MODE == 4) { // readlane -> valu -> readlane chain
#pragma unroll
            for (int i = 0; i < REP / 2; ++i) {
                int lo = __builtin_amdgcn_readlane(__double2loint(x0), i & 63);
                x0 = x0 + __hiloint2double(0x3ff00000, lo);
            }
        } else if (MODE == 5) { // dependent f32 add
            float f = (float)x0;
#pragma unroll
            for (int i = 0; i < REP; ++i) f = f + (float)b;
            x0 = f;
        } else if (MODE == 6) { // uniform branch every 2 fp64 adds (taken alternately)
#pragma unroll 1
            for (int i = 0; i < REP / 2; ++i) {
                x0 = x0 + b;
                if (x0 > 1e300) x0 = x0 * 0.5;
                x0 = x0 + b;
            }
        } else if (MODE == 7) { // 2 independent fp64 chains
#pragma unroll
            for (int i = 0; i < REP / 2; ++i) { x0 = x0 + b; x1 = x1 + b; }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + s0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks) {
    double *out; uint64_t *cyc;
    hipMalloc(&out, sizeof(double) * 64 * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 10, 1.0, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0, 1e-9);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * REP;
    if (MODE == 4 || MODE == 6) n = (double)iters * REP;   // counted as REP "slots"
    printf("%-34s blocks=%5d  %.2f ns/instr  (s_memtime ticks/instr %.2f)  kernel %.3f ms\n", name, blocks, ms * 1e6 / n, (double)h / n, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 1024, 2048, 4096}) {
        run<0>("dependent v_add_f64", blocks);
        run<7>("2 independent v_add_f64 chains", blocks);
        run<1>("4 independent v_add_f64 chains", blocks);
        run<3>("dependent v_mul_f64", blocks);
        run<5>("dependent v_add_f32", blocks);
        run<2>("dependent SALU (mul+add)", blocks);
        run<4>("readlane+v_add_f64 dependent pair", blocks);
        run<6>("2 fp64 adds + cmp + untaken branch", blocks);
        printf("\n");
    }
    return 0;
}
