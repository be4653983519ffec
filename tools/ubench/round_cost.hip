// Microbenchmark 12 (round 4): what the idioms of the slice kernel's round cost a LONE wave (one per SIMD), in cycles per block of
// instructions, against the 4.44-cycle issue floor: the EXEC-narrowing shrinkage step (v_cmpx), the same step with a plain compare,
// compares that write a scalar pair and are combined on the scalar side, the ballot -> inverse-ballot idiom, a compare feeding a
// scalar branch.  Each kernel repeats its block 8 x per loop iteration; s_memtime around 4096 iterations, block 0 reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define R8(X) X X X X X X X X
#define STEP_CMPX \
    "v_add_f64 %[W], %[R], -%[L]\n v_mul_f64 %[t], %[u], %[W]\n v_add_f64 %[x], %[L], %[t]\n v_cmp_lt_f64 vcc, %[x], %[xo]\n" \
    "v_mul_f64 %[t], %[x], %[x]\n v_add_u32 %[n], 1, %[n]\n v_add_f64 %[t], %[t], -%[Q]\n" \
    "v_cndmask_b32 %[La], %[La], %[xa], vcc\n v_cndmask_b32 %[Lb], %[Lb], %[xb], vcc\n v_cndmask_b32 %[Ra], %[xa], %[Ra], vcc\n v_cndmask_b32 %[Rb], %[xb], %[Rb], vcc\n" \
    "v_min_f64 %[dm], %[dm], |%[t]|\n"
#define KBEGIN(NAME) __global__ __launch_bounds__(64) void NAME(double *out, uint64_t *cyc, double b, double q) { \
    double L = -b, R = b, W = 0, t = 0, x = 0, dm = 1e300, xo = 0.25 * b, Q = q, u = 0.37; int n = 0; \
    int La, Lb, Ra, Rb, xa = 0, xb = 0; (void)La; (void)Lb; (void)Ra; (void)Rb; \
    uint64_t t0 = __builtin_amdgcn_s_memtime(); \
    _Pragma("unroll 1") for (int it = 0; it < ITER; ++it) {
#define KEND(NI) } uint64_t t1 = __builtin_amdgcn_s_memtime(); out[threadIdx.x + 64 * (blockIdx.x & 1)] = L + R + W + t + x + dm + n; \
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (uint64_t)(NI) * 8 * ITER; } }
// the halves of L / R / x as separate 32-bit registers are not expressible: the cndmask's operate on unrelated ints (same issue cost)
#define OPS : [L] "+v"(L), [R] "+v"(R), [W] "+v"(W), [t] "+v"(t), [x] "+v"(x), [dm] "+v"(dm), [n] "+v"(n), [La] "+v"(La), [Lb] "+v"(Lb), [Ra] "+v"(Ra), [Rb] "+v"(Rb) \
            : [u] "v"(u), [xo] "v"(xo), [Q] "v"(Q), [xa] "v"(xa), [xb] "v"(xb) : "vcc", "s40", "s41", "s42", "s43", "scc"
KBEGIN(k_step_cmpx)   La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmpx_ngt_f64 vcc, 0, %[Q]\n") "s_mov_b64 exec, -1\n" OPS); KEND(13)
KBEGIN(k_step_cmp)    La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmp_ngt_f64 vcc, 0, %[Q]\n") OPS); KEND(13)
KBEGIN(k_step_nocmp)  La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_add_u32 %[n], 1, %[n]\n") OPS); KEND(13)
KBEGIN(k_step_cmpx_s) La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmp_ngt_f64 vcc, 0, %[Q]\n s_and_b64 exec, exec, vcc\n") "s_mov_b64 exec, -1\n" OPS); KEND(14)
// six compares into scalar pairs, combined by s_and, then one select (the validity test of the round)
KBEGIN(k_valid_salu)  La = Lb = Ra = Rb = 0; asm volatile(R8(
    "v_cmp_gt_f64 s[40:41], %[W], %[Q]\n v_cmp_gt_f64 s[42:43], %[dm], %[t]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n"
    "v_cmp_lt_u32 s[42:43], %[n], %[La]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_cmp_lt_f64 s[42:43], %[x], %[xo]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n"
    "v_cndmask_b32 %[Ra], 0, %[Rb], s[40:41]\n") OPS); KEND(8)
// the same conditions as lane-mask VALU ops only: compares into vcc chained with v_cmp ... (no scalar side): cmp; cndmask 0/1; and ...
KBEGIN(k_valid_valu)  La = Lb = Ra = Rb = 0; asm volatile(R8(
    "v_cmp_gt_f64 vcc, %[W], %[Q]\n v_cndmask_b32 %[Ra], 0, %[Rb], vcc\n v_cmp_gt_f64 vcc, %[dm], %[t]\n v_cndmask_b32 %[Ra], 0, %[Ra], vcc\n"
    "v_cmp_lt_u32 vcc, %[n], %[La]\n v_cndmask_b32 %[Ra], 0, %[Ra], vcc\n v_cmp_lt_f64 vcc, %[x], %[xo]\n v_cndmask_b32 %[Ra], 0, %[Ra], vcc\n") OPS); KEND(8)
// v_cmpx-narrowed EXEC instead of combining masks: the word survives only in lanes that pass every test
KBEGIN(k_valid_cmpx)  La = Lb = Ra = Rb = 0; asm volatile(R8(
    "v_cmpx_gt_f64 vcc, %[W], %[Q]\n v_cmpx_gt_f64 vcc, %[dm], %[t]\n v_cmpx_lt_u32 vcc, %[n], %[La]\n v_cmpx_lt_f64 vcc, %[x], %[xo]\n v_mov_b32 %[Ra], %[Rb]\n s_mov_b64 exec, -1\n") OPS); KEND(6)
// readlane hop chain as the chase has it: readlane -> s_lshr -> (s_bitset, s_add) -> readlane
KBEGIN(k_hop)         La = threadIdx.x << 24; Lb = Ra = Rb = 0; asm volatile(R8(
    "v_readlane_b32 s40, %[La], s40\n s_lshr_b32 s40, s40, 24\n s_bitset1_b64 s[42:43], s40\n s_add_u32 s41, s41, s40\n") OPS); KEND(4)
KBEGIN(k_hop_bare)    La = threadIdx.x << 24; Lb = Ra = Rb = 0; asm volatile(R8(
    "v_readlane_b32 s40, %[La], s40\n s_lshr_b32 s40, s40, 24\n") OPS); KEND(2)
KBEGIN(k_hop_direct)  La = threadIdx.x; Lb = Ra = Rb = 0; asm volatile(R8(
    "v_readlane_b32 s40, %[La], s40\n") OPS); KEND(1)
// ballot idiom the compiler emits: v_cndmask 0/1 + v_cmp_ne (2 VALU) against using the pair directly
KBEGIN(k_fp_chain13)  La = Lb = Ra = Rb = 0; asm volatile(R8(
    "v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n"
    "v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n"
    "v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n v_add_f64 %[W], %[W], %[Q]\n") OPS); KEND(13)
KBEGIN(k_step_cmpx64) La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmpx_ngt_f64_e64 s[40:41], 0, %[Q]\n") "s_mov_b64 exec, -1\n" OPS); KEND(13)
KBEGIN(k_step_cmp64)  La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmp_ngt_f64_e64 s[40:41], 0, %[Q]\n") OPS); KEND(13)
KBEGIN(k_step_cmpu32) La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_CMPX "v_cmp_lt_u32 vcc, %[n], %[La]\n") OPS); KEND(13)
#define STEP_S42 \
    "v_add_f64 %[W], %[R], -%[L]\n v_mul_f64 %[t], %[u], %[W]\n v_add_f64 %[x], %[L], %[t]\n v_cmp_lt_f64_e64 s[42:43], %[x], %[xo]\n" \
    "v_mul_f64 %[t], %[x], %[x]\n v_add_u32 %[n], 1, %[n]\n v_add_f64 %[t], %[t], -%[Q]\n" \
    "v_cndmask_b32_e64 %[La], %[La], %[xa], s[42:43]\n v_cndmask_b32_e64 %[Lb], %[Lb], %[xb], s[42:43]\n v_cndmask_b32_e64 %[Ra], %[xa], %[Ra], s[42:43]\n v_cndmask_b32_e64 %[Rb], %[xb], %[Rb], s[42:43]\n" \
    "v_min_f64 %[dm], %[dm], |%[t]|\n"
KBEGIN(k_step_s42_cmpx) La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_S42 "v_cmpx_ngt_f64 vcc, 0, %[Q]\n") "s_mov_b64 exec, -1\n" OPS); KEND(13)
KBEGIN(k_step_s42_cmpx64) La = Lb = Ra = Rb = 0; asm volatile(R8(STEP_S42 "v_cmpx_ngt_f64_e64 s[40:41], 0, %[Q]\n") "s_mov_b64 exec, -1\n" OPS); KEND(13)
// the chained chase: five direct hops, then per level s_bitset1 + s_add
KBEGIN(k_chase_chained) La = threadIdx.x; Lb = Ra = Rb = 0; asm volatile(R8(
    "v_readlane_b32 s40, %[La], s40\n v_readlane_b32 s41, %[La], s40\n v_readlane_b32 s42, %[La], s41\n v_readlane_b32 s43, %[La], s42\n"
    "s_bitset1_b64 s[44:45], s40\n s_add_u32 s46, s46, s40\n s_bitset1_b64 s[44:45], s41\n s_add_u32 s46, s46, s41\n s_bitset1_b64 s[44:45], s42\n s_add_u32 s46, s46, s42\n s_add_u32 s46, s46, s43\n") OPS, "s44", "s45", "s46"); KEND(11)
KBEGIN(k_chase_now) La = threadIdx.x << 24; Lb = Ra = Rb = 0; asm volatile(R8(
    "v_readlane_b32 s40, %[La], s40\n s_lshr_b32 s41, s40, 24\n s_bitset1_b64 s[44:45], s41\n s_add_u32 s46, s46, s40\n"
    "v_readlane_b32 s40, %[La], s41\n s_lshr_b32 s41, s40, 24\n s_bitset1_b64 s[44:45], s41\n s_add_u32 s46, s46, s40\n"
    "v_readlane_b32 s40, %[La], s41\n s_lshr_b32 s41, s40, 24\n s_bitset1_b64 s[44:45], s41\n s_add_u32 s46, s46, s40\n"
    "v_readlane_b32 s40, %[La], s41\n s_add_u32 s46, s46, s40\n") OPS, "s44", "s45", "s46"); KEND(14)
template <typename K> void run(const char *name, K kern) {
    double *out; uint64_t *cyc, h[2];
    (void)hipMalloc(&out, 128 * 8); (void)hipMalloc(&cyc, 16);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, cyc, 1.5, 0.3); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    // (s_memtime ticks at the shader clock here: k_fp_chain13 reads the 4.44-cycle floor of tools/ubench/issue_floor.hip)
    printf("%-16s %8.1f cycles per block of %2d instructions = %5.2f cycles / instruction\n", name, (double)h[0] / (8.0 * ITER), (int)(h[1] / (8 * ITER)), (double)h[0] / (double)h[1]);
    (void)hipFree(out); (void)hipFree(cyc);
}
#define RUN(N) run(#N, N)
int main() {
    RUN(k_fp_chain13); RUN(k_step_nocmp); RUN(k_step_cmp); RUN(k_step_cmpx); RUN(k_step_cmpx_s);
    RUN(k_valid_salu); RUN(k_valid_valu); RUN(k_valid_cmpx); RUN(k_hop_direct); RUN(k_hop_bare); RUN(k_hop);
    RUN(k_step_cmpx64); RUN(k_step_cmp64); RUN(k_step_cmpu32); RUN(k_step_s42_cmpx); RUN(k_step_s42_cmpx64); RUN(k_chase_chained); RUN(k_chase_now);
    return 0;
}
