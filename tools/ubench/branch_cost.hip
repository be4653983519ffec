// Microbenchmark 2: cost of control flow and SGPR<->VGPR hand-offs for ONE wavefront on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096

#define BODY_BEGIN uint64_t t0 = __builtin_readcyclecounter(); for (int it = 0; it < ITER; ++it) {
#define BODY_END } uint64_t t1 = __builtin_readcyclecounter(); out[threadIdx.x] = x0; if (threadIdx.x == 0) cyc[0] = t1 - t0;

// empty loop: s_add, s_cmp, s_cbranch (taken)
__global__ void k_loop(double *out, uint64_t *cyc, double b) { double x0 = b; BODY_BEGIN asm volatile("s_nop 0"); BODY_END }
// loop + 8 dependent fp64 adds
__global__ void k_loop_add8(double *out, uint64_t *cyc, double b) { double x0 = b; BODY_BEGIN
    asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                 "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(b));
    BODY_END }
// loop + v_cmp -> s_cbranch_vccnz never taken
__global__ void k_vcc_untaken(double *out, uint64_t *cyc, double b) { double x0 = b; BODY_BEGIN
    asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_cbranch_vccnz 1f\n v_add_f64 %0, %0, %1\n 1:\n" : "+v"(x0) : "v"(b) : "vcc");
    BODY_END }
// loop + v_cmp -> s_cbranch_vccz always taken (skips one add)
__global__ void k_vcc_taken(double *out, uint64_t *cyc, double b) { double x0 = b; BODY_BEGIN
    asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_f64 %0, %0, %1\n 1:\n v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(b) : "vcc");
    BODY_END }
// 4 x (scalar cmp + untaken branch)
__global__ void k_scc_untaken4(double *out, uint64_t *cyc, double b, int z) { double x0 = b; BODY_BEGIN
    asm volatile("s_cmp_eq_u32 %1, 77\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %1, 78\n s_cbranch_scc1 1f\n"
                 "s_cmp_eq_u32 %1, 79\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %1, 80\n s_cbranch_scc1 1f\n 1:\n v_add_f64 %0, %0, %0" : "+v"(x0) : "s"(z) : "scc");
    BODY_END }
// 4 x (scalar cmp + taken branch to next line)
__global__ void k_scc_taken4(double *out, uint64_t *cyc, double b, int z) { double x0 = b; BODY_BEGIN
    asm volatile("s_cmp_lg_u32 %1, 77\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n s_cmp_lg_u32 %1, 78\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n"
                 "s_cmp_lg_u32 %1, 79\n s_cbranch_scc1 3f\n s_nop 0\n 3:\n s_cmp_lg_u32 %1, 80\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n v_add_f64 %0, %0, %0" : "+v"(x0) : "s"(z) : "scc");
    BODY_END }
// readlane (2) -> v_add using the SGPR pair, x4 dependent
__global__ void k_readlane_use4(double *out, uint64_t *cyc, double b) { double x0 = b; BODY_BEGIN
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int lo = __builtin_amdgcn_readlane(__double2loint(x0), 3 + 2 * i);
        int hi = __builtin_amdgcn_readlane(__double2hiint(x0), 3 + 2 * i);
        x0 = x0 + __hiloint2double(hi, lo);
    }
    BODY_END }
// v_cmp -> 4 cndmask (2 x 64-bit select), x4 dependent
__global__ void k_cmp_cndmask4(double *out, uint64_t *cyc, double b) { double x0 = b; double y = b * 2; BODY_BEGIN
#pragma unroll
    for (int i = 0; i < 4; ++i) { bool c = x0 < y; double nx = c ? y : x0; double ny = c ? x0 : y; x0 = nx + b; y = ny; }
    BODY_END out[64 + threadIdx.x] = y; }
// ds_bpermute round trip x4 dependent (shfl_xor of a double = 2 bpermutes)
__global__ void k_shfl4(double *out, uint64_t *cyc, double b) { double x0 = b + threadIdx.x; BODY_BEGIN
#pragma unroll
    for (int i = 0; i < 4; ++i) x0 = x0 + __shfl_xor(x0, 1 << i, 64);
    BODY_END }
// DPP row_shr based xor-1 exchange x4 dependent (quad_perm [1,0,3,2])
__global__ void k_dpp4(double *out, uint64_t *cyc, double b) { double x0 = b + threadIdx.x; BODY_BEGIN
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int lo = __builtin_amdgcn_mov_dpp(__double2loint(x0), 0xB1, 0xF, 0xF, true);
        int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x0), 0xB1, 0xF, 0xF, true);
        x0 = x0 + __hiloint2double(hi, lo);
    }
    BODY_END }
// ballot -> scalar ffs -> readlane with dynamic lane -> use  (x2)
__global__ void k_ballot_chain2(double *out, uint64_t *cyc, double b) { double x0 = b + threadIdx.x; BODY_BEGIN
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        unsigned long long m = __ballot(x0 > b);
        int l = __builtin_ctzll(m | (1ull << 63));
        int lo = __builtin_amdgcn_readlane(__double2loint(x0), l);
        int hi = __builtin_amdgcn_readlane(__double2hiint(x0), l);
        x0 = x0 + __hiloint2double(hi, lo);
    }
    BODY_END }

#define RUN(K, label, ...) do { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, cyc, __VA_ARGS__); hipDeviceSynchronize(); \
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-52s %8.1f cycles/iter\n", label, (double)h / ITER); } while (0)

int main() {
    double *out; uint64_t *cyc; hipMalloc(&out, 8 * 256); hipMalloc(&cyc, 8);
    RUN(k_loop, "empty loop (s_add,s_cmp,taken s_cbranch)", 1e-9);
    RUN(k_loop_add8, "loop + 8 dependent v_add_f64", 1e-9);
    RUN(k_vcc_untaken, "loop + v_cmp + s_cbranch_vccnz(untaken) + v_add", 1e-9);
    RUN(k_vcc_taken, "loop + v_cmp + s_cbranch_vccz(taken) + v_add", 1e-9);
    RUN(k_scc_untaken4, "loop + 4x(s_cmp + untaken s_cbranch) + v_add", 1e-9, 5);
    RUN(k_scc_taken4, "loop + 4x(s_cmp + taken s_cbranch over nop) + v_add", 1e-9, 5);
    RUN(k_readlane_use4, "loop + 4x(2 readlane -> v_add_f64 sgpr) dependent", 1e-9);
    RUN(k_cmp_cndmask4, "loop + 4x(v_cmp + 4 cndmask + v_add) dependent", 1e-9);
    RUN(k_shfl4, "loop + 4x(shfl_xor f64 + add) dependent", 1e-9);
    RUN(k_dpp4, "loop + 4x(2 dpp mov + add) dependent", 1e-9);
    RUN(k_ballot_chain2, "loop + 2x(ballot,ffs,2 readlane dyn,add) dependent", 1e-9);
    return 0;
}
