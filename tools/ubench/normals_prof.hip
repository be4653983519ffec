// Where a LONE wave of the bulk normal generator (pigeons.jl_amd/csrc/pte_normals.hpp) spends its cycles, per 512-output chunk:
// positions (9 per lane) | event pass | scalar walk over the events | gather + divide + store + tree.  One wave per SIMD (1024 blocks).
#define NRM_PROF 1
#include "../../pigeons.jl_amd/csrc/pte_normals.hpp"
#include <cstdio>
using namespace pte;
__global__ __launch_bounds__(64) void k(double *x, unsigned long long *prof, int d, double sd) {
    __shared__ NormalsLds L;
    const int lane = lane_id();
    if (lane < 8) L.prof[lane] = 0;
    normals_lds_init(L, lane);
    SeqRng r{0x1234567ull * (blockIdx.x + 1), 0x9e3779b97f4a7c15ull | 1ull};
    const double bs = normals_row(L, r, x + (size_t)blockIdx.x * d, d, sd, lane);
    if (lane == 0) x[(size_t)blockIdx.x * d] += bs;
    __syncthreads();
    if (blockIdx.x == 0 && lane < 8) prof[lane] = L.prof[lane];
}
int main() {
    const int d = 4096;
    for (int nb : {1024, 4096, 8192}) {
        double *x; unsigned long long *p, h[8];
        (void)hipMalloc(&x, (size_t)nb * d * 8); (void)hipMalloc(&p, 64);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(a, 0); hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, x, p, d, 1.7); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); }
        (void)hipMemcpy(h, p, 64, hipMemcpyDeviceToHost);
        const double c = (double)h[4];
        printf("%5d waves: launch %.3f ms (%.0f GB/s); block 0 per chunk: positions %.0f  event pass %.0f  walk %.0f  gather+store+tree %.0f cycles; %.1f events per chunk\n",
               nb, ms, (double)nb * d * 8 / ms / 1e6, h[0] / c, h[1] / c, h[2] / c, h[3] / c, h[5] / c);
        (void)hipFree(x); (void)hipFree(p);
    }
    return 0;
}
