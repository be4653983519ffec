// Microbenchmark 13 (round 4): what two waves of one workgroup pay to hand a word back and forth through LDS once per "round" when each
// sits alone on its SIMD (256 workgroups of 128 threads = the C2 shape): wave 0 works W cycles' worth of dependent FP64, publishes a word,
// s_barrier; wave 1 reads it, works a short chain, publishes, s_barrier; both continue.  Against the same work without the exchange.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
template <int SYNC>
__global__ __launch_bounds__(128) void k(double *out, uint64_t *cyc, double b, int work0, int work1) {
    __shared__ volatile int box[4];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double x = b + threadIdx.x; int word = threadIdx.x;
    if (threadIdx.x < 4) box[threadIdx.x] = 0;
    __syncthreads();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        const int nwork = wv == 0 ? work0 : work1;
#pragma unroll 1
        for (int k = 0; k < nwork; ++k) asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(x) : "v"(b));
        if (SYNC) {
            if (wv == 0 && (threadIdx.x & 63) == 0) box[0] = word + it;
            __syncthreads();
            if (wv == 1) { word += box[0]; asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(x) : "v"(b)); if ((threadIdx.x & 63) == 0) box[1] = word; }
            __syncthreads();
            word += box[1];
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 128 + threadIdx.x] = x + word;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out; uint64_t *cyc, c0, c1;
    (void)hipMalloc(&out, 256 * 128 * 8); (void)hipMalloc(&cyc, 8);
    for (int w0 : {128, 512}) for (int w1 : {96, 384}) {
        if ((w0 == 128) != (w1 == 96)) continue;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<0>, dim3(256), dim3(128), 0, 0, out, cyc, 1.5, w0, w1); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<1>, dim3(256), dim3(128), 0, 0, out, cyc, 1.5, w0, w1); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
        printf("work %4d / %4d x 4 adds per round: %7.1f cycles per round alone, %7.1f with publish + barrier + read + publish + barrier: +%.1f cycles\n",
               w0, w1, (double)c0 / ITER, (double)c1 / ITER, (double)(c1 - c0) / ITER);
    }
    return 0;
}
