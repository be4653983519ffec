// Does v_readlane -> v_readlane (lane select = the scalar the previous one wrote) need software wait states on gfx950?  The ISA manuals list
// "VALU writes SGPR -> V_READLANE lane select: 4 wait states"; hipcc inserts s_nop 3.  Chained hops through a permutation with and
// without the nops, checked against the host, with the cycles per hop of a lone wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define R8(X) X X X X X X X X
__global__ __launch_bounds__(64) void k_nonop(int *out, uint64_t *cyc) {
    int v = ((threadIdx.x * 37 + 11) & 63) | 0x12345600; int s = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) asm volatile(R8("v_readlane_b32 %0, %1, %0\n") : "+s"(s) : "v"(v));
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; if (blockIdx.x == 0) cyc[0] = t1 - t0; }
}
__global__ __launch_bounds__(64) void k_nop3(int *out, uint64_t *cyc) {
    int v = ((threadIdx.x * 37 + 11) & 63) | 0x12345600; int s = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) asm volatile(R8("v_readlane_b32 %0, %1, %0\n s_nop 3\n") : "+s"(s) : "v"(v));
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; if (blockIdx.x == 0) cyc[0] = t1 - t0; }
}
__global__ __launch_bounds__(64) void k_builtin(int *out, uint64_t *cyc) {
    int v = ((threadIdx.x * 37 + 11) & 63) | 0x12345600; int s = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s = __builtin_amdgcn_readlane(v, s);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; if (blockIdx.x == 0) cyc[0] = t1 - t0; }
}
template <typename K> void run(const char *name, K kern) {
    int *out, h[1024]; uint64_t *cyc, c;
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, cyc); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, out, 1024 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int s = 0; for (long i = 0; i < 8L * ITER; ++i) s = ((((s & 63) * 37 + 11) & 63) | 0x12345600);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (h[i] != s);
    printf("%-10s %6.2f cycles per hop; %d of 1024 waves end on the wrong value (expected %08x, wave 0 has %08x)\n", name, (double)c / (8.0 * ITER), bad, s, h[0]);
}
int main() { run("no nop", k_nonop); run("s_nop 3", k_nop3); run("builtin", k_builtin); return 0; }
