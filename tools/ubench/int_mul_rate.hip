// Microbenchmark: VALU issue cost of the integer multiplies SplitMix64 is made of (gfx950), per SIMD at full occupancy.
// Every mode runs REP independent-enough instructions per iteration on 8 waves per SIMD (2048 workgroups of 256 threads); the line says how many
// SIMD cycles one wave64 instruction occupies (4 = full rate: 16 lanes per cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t a, uint32_t b) {
    uint32_t x0 = a + threadIdx.x, x1 = a ^ 0x9e3779b9u, x2 = a * 3u + 1u, x3 = a + 77u;
    uint64_t y0 = ((uint64_t)a << 32) | threadIdx.x, y1 = y0 * 3 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            if (MODE == 0) { x0 += b; x1 += b; x2 += b; x3 += b; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (MODE == 1) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (MODE == 2) { x0 = __umulhi(x0, b); x1 = __umulhi(x1, b); x2 = __umulhi(x2, b); x3 = __umulhi(x3, b); asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (MODE == 3) { x0 = __umul24(x0, b); x1 = __umul24(x1, b); x2 = __umul24(x2, b); x3 = __umul24(x3, b); asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (MODE == 4) {   // v_mad_u64_u32: 32 x 32 + 64 -> 64
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(y0), "+v"(y1) : "v"(x0), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(y0), "+v"(y1) : "v"(x1), "v"(b) : "vcc");
            }
            else if (MODE == 5) {   // SplitMix64's finaliser as the compiler builds it (4 evaluations = REP / 4 iterations of one each here)
                uint64_t z = y0 + (uint64_t)i * 0x9e3779b97f4a7c15ull;
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; y1 ^= z ^ (z >> 31);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (uint32_t)y0 + (uint32_t)(y0 >> 32) + (uint32_t)y1 + (uint32_t)(y1 >> 32);
}
template <int MODE> void run(const char *name, double per_iter) {
    uint32_t *out; const int blocks = 2048; hipMalloc(&out, 4 * 256 * blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 12345u, 77u); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 77u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 2048 workgroups x 4 waves = 8192 waves on 1024 SIMDs = 8 waves per SIMD; SIMD-seconds per wave-instruction:
    const double n = (double)iters * per_iter * 8.0;       // wave-instructions (or evaluations) per SIMD
    printf("%-44s %8.3f ms   %.2f ns per wave64 %s per SIMD  (= %.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / n, MODE == 5 ? "evaluation" : "instruction", ms * 1e6 / n * 2.4);
    hipFree(out);
}
int main() {
    run<0>("v_add_u32", REP); run<1>("v_mul_lo_u32", REP); run<2>("v_mul_hi_u32", REP); run<3>("v_mul_u32_u24", REP);
    run<4>("v_mad_u64_u32", REP); run<5>("mix64 (seed + i gamma), compiler's code", REP / 4);
    return 0;
}
