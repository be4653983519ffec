// Development harness of the bulk normal generator (pigeons.jl_amd/csrc/pte_normals.hpp): the shape the HBM-bound kernels are profiled at
// (N replicas x d normals, NRM_WPB waves per workgroup, the kernel's occupancy attribute), timed with HIP events, plus an FNV-1a checksum of
// every output byte, every final stream position and every block-sum root -- two builds of the header (-D variants) must print the same
// checksum to be the same generator.  Usage: normals_dev.bin [N [d [sd]]]
#include "../../pigeons.jl_amd/csrc/pte_normals.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace pte;
__global__ __launch_bounds__(64 * NRM_WPB) NRM_ATTR void k(double *x, unsigned long long *seeds, double *roots, int N, int d, double sd0) {
    __shared__ NormalsLds L;
    const int lane = lane_id();
    normals_lds_init(L, lane);
    const int i = blockIdx.x * NRM_WPB + (NRM_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0);
    if (i >= N) return;
    SeqRng r{0x1234567ull * (unsigned long long)(i + 1), mix_gamma(0x9e3779b97f4a7c15ull * (unsigned long long)(i + 3))};
    const double sd = sd0 + 1e-3 * (i & 255);
    const double S = upper_tree_root_dyn(normals_row(L, r, x + (size_t)i * d, d, sd, lane), 6);
    if (lane == 0) { seeds[i] = r.seed; roots[i] = S; }
}
static unsigned long long fnv(const void *p, size_t n, unsigned long long h) {
    const unsigned long long *q = (const unsigned long long *)p;
    for (size_t i = 0; i < n / 8; ++i) { h ^= q[i]; h *= 0x100000001b3ull; }
    return h;
}
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8192, d = argc > 2 ? atoi(argv[2]) : 4096;
    const double sd = argc > 3 ? atof(argv[3]) : 1.7;
    const size_t dyn_lds = argc > 4 ? (size_t)atoi(argv[4]) : 0;     // extra dynamic LDS per workgroup: caps the workgroups resident per CU (160 KB / (31 KB + this))
    double *x, *roots; unsigned long long *seeds;
    (void)hipMalloc(&x, (size_t)N * d * 8); (void)hipMalloc(&roots, (size_t)N * 8); (void)hipMalloc(&seeds, (size_t)N * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f, sum = 0; const int reps = 10;
    for (int rep = 0; rep < reps + 2; ++rep) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k, dim3((N + NRM_WPB - 1) / NRM_WPB), dim3(64 * NRM_WPB), dyn_lds, 0, x, seeds, roots, N, d, sd);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 2) { if (ms < best) best = ms; sum += ms; }
    }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<double> hx((size_t)N * d), hr(N); std::vector<unsigned long long> hs(N);
    (void)hipMemcpy(hx.data(), x, (size_t)N * d * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), roots, (size_t)N * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs.data(), seeds, (size_t)N * 8, hipMemcpyDeviceToHost);
    unsigned long long h = fnv(hx.data(), (size_t)N * d * 8, 0xcbf29ce484222325ull);
    h = fnv(hr.data(), (size_t)N * 8, h); h = fnv(hs.data(), (size_t)N * 8, h);
    const double bytes = (double)N * d * 8;
    if (dyn_lds) printf("dynLDS=%zu ", dyn_lds);
    printf("N=%d d=%d sd=%g  best %.4f ms = %.0f GB/s   mean %.4f ms = %.0f GB/s   checksum %016llx\n", N, d, sd, best, bytes / best / 1e6, sum / reps, bytes / (sum / reps) / 1e6, h);
    return 0;
}
