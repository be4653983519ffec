// Microbenchmark 4: is a lone wave latency-bound or issue-bound on the shrink-loop body?
// NS independent per-lane streams of the slice7 shrink step run interleaved in one loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int NS>
__global__ __launch_bounds__(64) void k(double *out, uint64_t *cyc, double a, double b, const double *uu) {
    __shared__ double s_u[512];
    for (int i = threadIdx.x; i < 512; i += 64) s_u[i] = uu[i];
    __syncthreads();
    double Lb[NS], Rb[NS], xold[NS], dmin[NS], Q[NS], Vn[NS];
    const double *up[NS];
    for (int s = 0; s < NS; ++s) { Lb[s] = a - s; Rb[s] = a + 10.0 + s; xold[s] = a + 3.0 + 0.1 * threadIdx.x; dmin[s] = 1e300; Q[s] = b * s; up[s] = &s_u[(threadIdx.x + 7 * s) & 63]; Vn[s] = *up[s]; }
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double W = Rb[s] - Lb[s];
            double v = Lb[s] + Vn[s] * W;
            up[s] = &s_u[((up[s] - s_u) + 1) & 255];
            Vn[s] = *up[s];
            double d = v * v - Q[s];
            dmin[s] = fmin(dmin[s], fabs(d));
            bool below = v < xold[s];
            Lb[s] = below ? v : Lb[s];
            Rb[s] = below ? Rb[s] : v;
            if (Rb[s] - Lb[s] < 1e-3) { Lb[s] -= 5.0; Rb[s] += 5.0; }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    double acc = 0;
    for (int s = 0; s < NS; ++s) acc += Lb[s] + Rb[s] + dmin[s];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NS> void run() {
    double *out; uint64_t *cyc; double *uu;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&uu, 512 * 8);
    double h[512]; for (int i = 0; i < 512; ++i) h[i] = (i * 0.6180339887) - (int)(i * 0.6180339887);
    hipMemcpy(uu, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<NS>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 0.5, uu); hipDeviceSynchronize(); }
    uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("NS=%d: %.1f cycles/iteration, %.1f per stream-step\n", NS, (double)c / ITER, (double)c / ITER / NS);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); return 0; }
