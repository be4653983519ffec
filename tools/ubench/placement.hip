// Microbenchmark 9 (round 3): where do the 1024 single-wave workgroups of a launch land, and do they all run at the same speed?
// issue_floor.hip showed every VALU instruction of the MEASURED wave (block 0) costs 4.44 s_memtime ticks, yet the 1024-block
// launches took 10.4 ms where 4.44 ticks x 4.19 M instructions / 2.39 GHz = 7.8 ms -- the launch is as slow as its slowest wave.
// Each block records its hardware id (XCC, SE, CU, SIMD), its start / end on the constant 100 MHz clock and its own tick count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#define REP (1 << 15)
#define R8(X) X X X X X X X X

struct Rec { uint32_t hw_id, xcc_id; uint64_t t0, t1, ticks; };

template <int MODE>
__global__ __launch_bounds__(64) void k_place(Rec *rec, double *out, double b, int lds_bytes) {
    extern __shared__ double dyn[];
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_readcyclecounter();
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3; int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3; const int c = (int)b;
    if (lds_bytes) dyn[threadIdx.x] = b;
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        if (MODE == 0)
            asm volatile(R8("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                            "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(c));
        else
            asm volatile(R8("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                            "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + i0 + i1 + i2 + i3 + (lds_bytes ? dyn[63 - threadIdx.x] : 0.0);
    if (threadIdx.x == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        r.t0 = r0; r.t1 = r1; r.ticks = c1 - c0;
        rec[blockIdx.x] = r;
    }
}

template <int MODE> void run(const char *name, int nb, int lds_bytes) {
    Rec *d; double *out;
    (void)hipMalloc(&d, sizeof(Rec) * nb); (void)hipMalloc(&out, 64 * 8);
    std::vector<Rec> h(nb);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_place<MODE>, dim3(nb), dim3(64), lds_bytes, 0, d, out, 1.0000001, lds_bytes);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h.data(), d, sizeof(Rec) * nb, hipMemcpyDeviceToHost);
    // placement: waves per (xcc, se, cu, simd)
    std::map<uint32_t, int> per_simd, per_cu;
    uint64_t tmin = ~0ull, tmax = 0;
    for (auto &r : h) {
        const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 15;
        const uint32_t cu_key = (xcc << 16) | (se << 12) | (sh << 8) | cu;
        per_simd[(cu_key << 2) | simd]++; per_cu[cu_key]++;
        tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
    }
    int hist_simd[9] = {0}, hist_cu[33] = {0};
    for (auto &kv : per_simd) hist_simd[std::min(kv.second, 8)]++;
    for (auto &kv : per_cu) hist_cu[std::min(kv.second, 32)]++;
    printf("%s: %d blocks, lds %d B, launch %.3f ms (first start -> last end on the 100 MHz clock: %.3f ms)\n", name, nb, lds_bytes, ms, (tmax - tmin) / 1e5);
    printf("  CUs used %zu, SIMDs used %zu; SIMDs holding 1/2/3/4+ waves: %d/%d/%d/%d; CUs holding 1..8 waves:", per_cu.size(), per_simd.size(),
           hist_simd[1], hist_simd[2], hist_simd[3], hist_simd[4] + hist_simd[5] + hist_simd[6] + hist_simd[7] + hist_simd[8]);
    for (int k = 1; k <= 8; ++k) printf(" %d", hist_cu[k]);
    printf("\n");
    // per-wave duration by how many waves share its SIMD
    double sum[9] = {0}, mx[9] = {0}, tk[9] = {0}; int cnt[9] = {0};
    double start_spread = 0;
    for (auto &r : h) {
        const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 15;
        const uint32_t key = ((((xcc << 16) | (se << 12) | (sh << 8) | cu)) << 2) | simd;
        const int s = std::min(per_simd[key], 8);
        const double dur = (r.t1 - r.t0) / 1e5;
        sum[s] += dur; mx[s] = std::max(mx[s], dur); tk[s] += (double)r.ticks; cnt[s]++;
        start_spread = std::max(start_spread, (double)(r.t0 - tmin) / 1e5);
    }
    for (int s = 1; s <= 8; ++s)
        if (cnt[s]) printf("  waves on a SIMD shared by %d: %5d waves, mean %.3f ms, max %.3f ms, %.2f ticks per instruction\n", s, cnt[s], sum[s] / cnt[s], mx[s], tk[s] / cnt[s] / ((double)REP * 64));
    printf("  last wave started %.3f ms after the first\n", start_spread);
    (void)hipFree(d); (void)hipFree(out);
}
int main() {
    run<0>("v_add_u32", 1024, 0);
    run<0>("v_add_u32", 1024, 14336);
    run<0>("v_add_u32", 1024, 40960);     // 3 blocks per CU by LDS at most ... (160 KB / 40 KB = 4)
    run<0>("v_add_u32", 2048, 0);
    run<0>("v_add_u32", 256, 0);
    run<1>("v_add_f64", 1024, 0);
    run<1>("v_add_f64", 1024, 14336);
    run<1>("v_add_f64", 2048, 0);
    run<1>("v_add_f64", 256, 0);
    return 0;
}
