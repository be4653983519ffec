// Development harness of the bulk normal generator (pigeons.jl_amd/csrc/pte_normals.hpp): the shape the HBM-bound kernels are profiled at
// (N replicas x d normals, NRM_WPB waves per workgroup, the kernel's occupancy attribute), timed with HIP events, plus an FNV-1a checksum of
// every output byte, every final stream position and every block-sum root -- two builds of the header (-D variants) must print the same
// checksum to be the same generator.  Usage: normals_dev.bin [N [d [sd]]]
#include "../../pigeons.jl_amd/csrc/pte_normals.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace pte;
#ifdef NRM_STAMP            // per-wave time stamps on the 100 MHz clock: entry, after the table build, exit (stamps[3 i ..])
__device__ unsigned long long *g_stamps;
#define STAMP(k_) do { if (lane == 0 && i < N) g_stamps[3 * i + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k_) do {} while (0)
#endif
__global__ __launch_bounds__(64 * NRM_WPB) NRM_ATTR void k(double *x, unsigned long long *seeds, double *roots, int N, int d, double sd0) {
    __shared__ NormalsLds L;
    const int lane = lane_id();
#ifdef NRM_STAMP
    { const int i = blockIdx.x * NRM_WPB + (int)(threadIdx.x >> 6); STAMP(0); }
#endif
    normals_lds_init(L, lane);
    const int i = blockIdx.x * NRM_WPB + (NRM_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0);
    if (i >= N) return;
    STAMP(1);
    SeqRng r{0x1234567ull * (unsigned long long)(i + 1), mix_gamma(0x9e3779b97f4a7c15ull * (unsigned long long)(i + 3))};
    const double sd = sd0 + 1e-3 * (i & 255);
    const double S = upper_tree_root_dyn(normals_row(L, r, x + (size_t)i * d, d, sd, lane), 6);
    if (lane == 0) { seeds[i] = r.seed; roots[i] = S; }
    STAMP(2);
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
#ifdef NRM_STAMP
    unsigned long long *stamps; (void)hipMalloc(&stamps, (size_t)N * 24); (void)hipMemset(stamps, 0, (size_t)N * 24);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof stamps);
#endif
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
#ifdef NRM_STAMP
    {   // the LAST launch's stamps: when do waves start, finish the table build, end (us after the first wave's entry)
        std::vector<unsigned long long> st((size_t)N * 3);
        (void)hipMemcpy(st.data(), stamps, (size_t)N * 24, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int i = 0; i < N; ++i) if (st[3 * i] && st[3 * i] < t0) t0 = st[3 * i];
        std::vector<double> s0, s1, s2; for (int i = 0; i < N; ++i) { s0.push_back((st[3 * i] - t0) / 100.0); s1.push_back((st[3 * i + 1] - t0) / 100.0); s2.push_back((st[3 * i + 2] - t0) / 100.0); }
        auto q = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        printf("stamps (us after the first wave's entry): entry p0/p50/p62/p63/p99/max %.1f %.1f %.1f %.1f %.1f %.1f | tables built p50/max %.1f %.1f | exit p0/p10/p50/p62/p90/p99/max %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n",
               q(s0, 0), q(s0, .5), q(s0, .62), q(s0, .63), q(s0, .99), q(s0, 1), q(s1, .5), q(s1, 1), q(s2, 0), q(s2, .1), q(s2, .5), q(s2, .62), q(s2, .9), q(s2, .99), q(s2, 1));
        std::vector<double> dur; for (int i = 0; i < N; ++i) dur.push_back(s2[i] - s0[i]);
        printf("wave life (us): first 62 %% of the rows (first to start) p50 %.1f; all p10/p50/p90/max %.1f %.1f %.1f %.1f\n", q(std::vector<double>(dur.begin(), dur.begin() + (size_t)(0.62 * N)), .5), q(dur, .1), q(dur, .5), q(dur, .9), q(dur, 1));
    }
#endif
    unsigned long long h = fnv(hx.data(), (size_t)N * d * 8, 0xcbf29ce484222325ull);
    h = fnv(hr.data(), (size_t)N * 8, h); h = fnv(hs.data(), (size_t)N * 8, h);
    const double bytes = (double)N * d * 8;
    if (dyn_lds) printf("dynLDS=%zu ", dyn_lds);
    printf("N=%d d=%d sd=%g  best %.4f ms = %.0f GB/s   mean %.4f ms = %.0f GB/s   checksum %016llx\n", N, d, sd, best, bytes / best / 1e6, sum / reps, bytes / (sum / reps) / 1e6, h);
    return 0;
}
