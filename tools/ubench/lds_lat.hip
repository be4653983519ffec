// Microbenchmark 8 (round 3): what one wave alone on its SIMD pays per LDS round trip and per readlane hop.
// A "round trip" here is what a speculative round of k_explore_slice8 does: N ds_read_b64 at per-lane addresses that depend
// on the previous round trip's data (so nothing overlaps), then s_waitcnt lgkmcnt(0), then one dependent VALU use.
// Reported in s_memtime ticks (= shader cycles, MI355X_MICROARCH.md) per round trip, one wave per SIMD (1024 blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 20000

template <int N>
__global__ __launch_bounds__(64) void k_lds(double *out, uint64_t *cyc, int stride) {
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (double)((i * 7) & 63);
    __syncthreads();
    int idx = threadIdx.x;
    double acc = 0.0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        double v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = s[(idx + k * stride) & 1023];
        double t = v[0];
#pragma unroll
        for (int k = 1; k < N; ++k) t += v[k];
        acc += t;
        idx = (int)t + threadIdx.x;                   // the next addresses depend on this round trip's data
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// the same loop without the LDS reads (address arithmetic + adds + conversion only): subtract to get the round trip itself
template <int N>
__global__ __launch_bounds__(64) void k_nolds(double *out, uint64_t *cyc, int stride) {
    int idx = threadIdx.x;
    double acc = 0.0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        double v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { int a = (idx + k * stride) & 1023; asm volatile("" : "+v"(a)); v[k] = (double)(a & 63); }
        double t = v[0];
#pragma unroll
        for (int k = 1; k < N; ++k) t += v[k];
        acc += t;
        idx = (int)t + threadIdx.x;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// readlane hop chain: v_readlane_b32 s, v, s_prev (lane index from the previous readlane), H hops per iteration
template <int H>
__global__ __launch_bounds__(64) void k_hop(double *out, uint64_t *cyc, int stride) {
    int packed = (threadIdx.x * 5 + stride) & 63;
    int cur = 0, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int pk = __builtin_amdgcn_readlane(packed, cur);
            acc += pk;
            cur = (pk + it) & 63;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// ds_write_b64 by a ballot-selected subset, then a dependent ds_read of what was written (the round's store -> next head)
__global__ __launch_bounds__(64) void k_wr(double *out, uint64_t *cyc, int stride) {
    __shared__ double s[256];
    s[threadIdx.x] = threadIdx.x; s[threadIdx.x + 64] = 1; s[threadIdx.x + 128] = 2; s[threadIdx.x + 192] = 3;
    __syncthreads();
    double acc = 0.0; int l = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        const double x = s[(l + (threadIdx.x & 7)) & 255];
        acc += x;
        if ((threadIdx.x & 15) == (it & 15)) s[(l + (threadIdx.x & 7)) & 255] = acc;
        __builtin_amdgcn_wave_barrier();
        l += 3 + stride;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K> double run(const char *name, K kern, int nb = 1024) {
    double *out; uint64_t *cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    uint64_t best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(nb), dim3(64), 0, 0, out, cyc, 1);
        (void)hipDeviceSynchronize();
        uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); if (c < best) best = c;
    }
    const double per = (double)best / REP;
    printf("%-44s %4d waves: %7.1f ticks per iteration\n", name, nb, per);
    return per;
}
int main() {
    for (int nb : {1024, 2048}) {
        const double l1 = run("1 x ds_read_b64 round trip + use", k_lds<1>, nb), n1 = run("  (same without the LDS read)", k_nolds<1>, nb);
        const double l2 = run("2 x ds_read_b64 round trip + use", k_lds<2>, nb), n2 = run("  (same without the LDS reads)", k_nolds<2>, nb);
        const double l5 = run("5 x ds_read_b64 round trip + use", k_lds<5>, nb), n5 = run("  (same without the LDS reads)", k_nolds<5>, nb);
        const double l9 = run("9 x ds_read_b64 round trip + use", k_lds<9>, nb), n9 = run("  (same without the LDS reads)", k_nolds<9>, nb);
        const double l14 = run("14 x ds_read_b64 round trip + use", k_lds<14>, nb), n14 = run("  (same without the LDS reads)", k_nolds<14>, nb);
        printf("=> LDS round trip alone, %d waves: 1 read %.0f, 2 reads %.0f, 5 reads %.0f, 9 reads %.0f, 14 reads %.0f ticks\n", nb, l1 - n1, l2 - n2, l5 - n5, l9 - n9, l14 - n14);
        run("1 readlane hop (index from the previous)", k_hop<1>, nb);
        run("4 readlane hops", k_hop<4>, nb);
        run("ds_read -> masked ds_write -> next ds_read", k_wr, nb);
    }
    return 0;
}
