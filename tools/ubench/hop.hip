// Microbenchmark 12 (round 3): the price of one step of a scalar pointer chase through the lanes, by how the next lane index is
// formed.  One wave per SIMD; s_memtime ticks per hop.
//   A  v_readlane -> s_bfe_u32 (extract the index) -> v_readlane            (the slice kernel's chase)
//   B  v_readlane -> v_readlane, the word read IS the next lane select (bits 5:0)  (tried in round 3: slower in the kernel)
//   C  as A with two more independent SALU operations per hop (mask / sum bookkeeping)
//   D  v_readlane -> v_mov (broadcast to a VGPR) -> ds_bpermute_b32 -> ... a vector-only chase through the LDS crossbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 20000
template <int MODE>
__global__ __launch_bounds__(64) void k(int *out, uint64_t *cyc, int salt) {
    // lane i holds the index of "the next lane" in bits 5:0 (MODE B) or bits 13:8 (A, C); a permutation so that the chain wanders
    const int nxt = (threadIdx.x * 37 + 11 + salt) & 63;
    const int word = (MODE == 1 || MODE == 3) ? (nxt | 0x1000100) : ((nxt << 8) | 0x10005);
    int cur = 0, acc = 0; uint64_t m = 1;
    int vcur = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (MODE == 0) { const int pk = __builtin_amdgcn_readlane(word, cur); cur = (pk >> 8) & 63; }
            else if (MODE == 1) { cur = __builtin_amdgcn_readlane(word, cur); }
            else if (MODE == 2) { const int pk = __builtin_amdgcn_readlane(word, cur); m |= 1ull << cur; acc += pk & 0x100FF; cur = (pk >> 8) & 63; }
            else { vcur = __builtin_amdgcn_ds_bpermute((vcur & 63) << 2, word); }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = cur + acc + (int)m + vcur;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name) {
    int *out; uint64_t *cyc; (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8);
    uint64_t best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, cyc, 3); (void)hipDeviceSynchronize(); uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); if (c < best) best = c; }
    printf("%-78s %6.1f ticks per hop\n", name, (double)best / REP / 4);
}
int main() {
    run<0>("A  v_readlane -> s_lshr/s_and -> v_readlane");
    run<1>("B  v_readlane -> v_readlane (the word is the lane select)");
    run<2>("C  A + mask / sum bookkeeping in the loop");
    run<3>("D  ds_bpermute_b32 chain (vector only)");
    return 0;
}
