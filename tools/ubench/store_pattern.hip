// Microbenchmark 11 (round 3): HBM write rate of 8192 waves x 4096 doubles (256 MiB) by store shape.
//  A: 8 B per lane, a wave-instruction writes 512 contiguous bytes             (one leaf per lane: the old layout)
//  B: 16 B per lane, a wave-instruction writes 1 KiB contiguous
//  C: 2 x 16 B per lane, 32 B between lanes: each instruction writes every other 16 B  (four consecutive outputs per lane)
//  D: as C, but the two halves exchanged between lane pairs first, so that each instruction writes 32-byte runs
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k_store(double *x, int d, double v) {
    double *row = x + (size_t)blockIdx.x * d;
    const int lane = threadIdx.x;
    if (MODE == 0) { for (int i = lane; i < d; i += 64) row[i] = v + i; }
    else if (MODE == 1) { for (int i = 2 * lane; i < d; i += 128) *reinterpret_cast<double2 *>(row + i) = make_double2(v + i, v); }
    else if (MODE == 2) { for (int i = 4 * lane; i < d; i += 256) { double2 *p = reinterpret_cast<double2 *>(row + i); p[0] = make_double2(v + i, v); p[1] = make_double2(v, v + i); } }
    else { for (int i = 4 * lane; i < d; i += 256) {
            // lane pair (2m, 2m+1): the even lane writes both first halves (32 contiguous bytes ... of different quads), i.e. run of 32 B
            const int pair = lane >> 1, odd = lane & 1;
            double2 *p = reinterpret_cast<double2 *>(row + (i - 4 * lane) + 8 * pair + 2 * odd);
            p[0] = make_double2(v + i, v); p[2] = make_double2(v, v + i);
        } }
}
template <int MODE> void run(const char *name) {
    const int N = 8192, d = 4096; double *x; (void)hipMalloc(&x, (size_t)N * d * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a, 0); hipLaunchKernelGGL(k_store<MODE>, dim3(N), dim3(64), 0, 0, x, d, 1.0); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-60s %.3f ms  %.0f GB/s\n", name, best, (double)N * d * 8 / best / 1e6);
    (void)hipFree(x);
}
int main() {
    run<0>("A  8 B per lane, 512 B contiguous per instruction");
    run<1>("B 16 B per lane, 1 KiB contiguous per instruction");
    run<2>("C 2 x 16 B per lane at a 32 B lane stride");
    run<3>("D 2 x 16 B per lane, 32-byte runs per instruction");
    return 0;
}
