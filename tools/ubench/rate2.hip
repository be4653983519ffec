// Microbenchmark 11 (round 4): SIMD cycles per wave-instruction at 1 / 4 / 8 waves per SIMD for the opcodes rate.hip left out --
// VOP3-encoded 32-bit integer ops (v_add3_u32, v_lshl_add_u32, v_alignbit_b32, v_bfi_b32, v_and_or_b32, v_bitop3_b32, v_perm_b32),
// v_mbcnt, SDWA, carry pairs, compares that write an SGPR pair, v_readfirstlane, 64-bit moves / adds, packed f32.
// Prices the instruction diet of the normal generator (pte_normals.hpp): which replacements are cheaper in PIPE time, not in count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define REP 4096
#define R8(X) X X X X X X X X
#define KERN32(NAME, BODY)                                                                                         \
    __global__ __launch_bounds__(64) void k_##NAME(double *out, double b, int c) {                                  \
        int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;                                                               \
        _Pragma("unroll 1") for (int it = 0; it < REP; ++it) asm volatile(R8(BODY) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c), "v"(b), "s"(c) : "vcc", "s40", "s41", "s42", "s43"); \
        out[threadIdx.x + 64 * (blockIdx.x & 1)] = (double)(x0 + x1 + x2 + x3);                                    \
    }
#define KERN64(NAME, BODY)                                                                                         \
    __global__ __launch_bounds__(64) void k_##NAME(double *out, double b, int c) {                                  \
        unsigned long long x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, cc = c;                                        \
        _Pragma("unroll 1") for (int it = 0; it < REP; ++it) asm volatile(R8(BODY) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(cc), "v"(b), "s"(c) : "vcc", "s40", "s41", "s42", "s43"); \
        out[threadIdx.x + 64 * (blockIdx.x & 1)] = (double)(x0 + x1 + x2 + x3);                                    \
    }
KERN32(v_add3_u32, "v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4\n")
KERN32(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4\n")
KERN32(v_alignbit_b32, "v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7\n")
KERN32(v_bfi_b32, "v_bfi_b32 %0, %4, %0, %4\n v_bfi_b32 %1, %4, %1, %4\n v_bfi_b32 %2, %4, %2, %4\n v_bfi_b32 %3, %4, %3, %4\n")
KERN32(v_and_or_b32, "v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4\n")
KERN32(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %4\n v_lshl_or_b32 %1, %1, 3, %4\n v_lshl_or_b32 %2, %2, 3, %4\n v_lshl_or_b32 %3, %3, 3, %4\n")
KERN32(v_bitop3_b32, "v_bitop3_b32 %0, %0, %4, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %4 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %4 bitop3:0x96\n")
KERN32(v_perm_b32, "v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4\n")
KERN32(v_bfe_u32, "v_bfe_u32 %0, %0, 1, 9\n v_bfe_u32 %1, %1, 1, 9\n v_bfe_u32 %2, %2, 1, 9\n v_bfe_u32 %3, %3, 1, 9\n")
KERN32(v_xor_b32_e64, "v_xor_b32_e64 %0, %0, %4\n v_xor_b32_e64 %1, %1, %4\n v_xor_b32_e64 %2, %2, %4\n v_xor_b32_e64 %3, %3, %4\n")
KERN32(v_xor_b32_sgpr, "v_xor_b32 %0, %6, %0\n v_xor_b32 %1, %6, %1\n v_xor_b32 %2, %6, %2\n v_xor_b32 %3, %6, %3\n")
KERN32(v_xor_b32_lit, "v_xor_b32 %0, 0x12345678, %0\n v_xor_b32 %1, 0x12345678, %1\n v_xor_b32 %2, 0x12345678, %2\n v_xor_b32 %3, 0x12345678, %3\n")
KERN32(v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n")
KERN32(v_and_b32, "v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3\n")
KERN32(v_mov_b32, "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
KERN32(v_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %6, %0\n v_mbcnt_lo_u32_b32 %1, %6, %1\n v_mbcnt_lo_u32_b32 %2, %6, %2\n v_mbcnt_lo_u32_b32 %3, %6, %3\n")
KERN32(v_mbcnt_hi, "v_mbcnt_hi_u32_b32 %0, %6, %0\n v_mbcnt_hi_u32_b32 %1, %6, %1\n v_mbcnt_hi_u32_b32 %2, %6, %2\n v_mbcnt_hi_u32_b32 %3, %6, %3\n")
KERN32(v_lshl_sdwa, "v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
KERN32(v_add_co_pair, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n")
KERN32(v_cmp_u32_vcc, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4\n")
KERN32(v_cmp_u32_sgpr, "v_cmp_lt_u32_e64 s[40:41], %0, %4\n v_cmp_lt_u32_e64 s[42:43], %1, %4\n v_cmp_lt_u32_e64 s[40:41], %2, %4\n v_cmp_lt_u32_e64 s[42:43], %3, %4\n")
KERN32(v_cndmask_vcc, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERN32(v_readfirstlane, "v_readfirstlane_b32 s40, %0\n v_readfirstlane_b32 s41, %1\n v_readfirstlane_b32 s42, %2\n v_readfirstlane_b32 s43, %3\n")
KERN32(v_mul_lo_sgpr, "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %6\n")
KERN32(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4\n")
KERN32(v_mad_u32_u16, "v_mad_u32_u16 %0, %0, %4, %4\n v_mad_u32_u16 %1, %1, %4, %4\n v_mad_u32_u16 %2, %2, %4, %4\n v_mad_u32_u16 %3, %3, %4, %4\n")
KERN32(v_fma_f32, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")
KERN32(v_mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
KERN32(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
KERN32(v_mov_dpp_rowshr, "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERN32(v_add_u32_dpp, "v_add_u32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %2, %4, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERN64(v_mov_b64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n")
KERN64(v_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n")
KERN64(v_pk_add_f32, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
KERN64(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
KERN64(v_cmp_u64_sgpr, "v_cmp_lt_u64_e64 s[40:41], %0, %4\n v_cmp_lt_u64_e64 s[42:43], %1, %4\n v_cmp_lt_u64_e64 s[40:41], %2, %4\n v_cmp_lt_u64_e64 s[42:43], %3, %4\n")
KERN64(v_cmp_f64_sgpr, "v_cmp_lt_f64_e64 s[40:41], %0, %4\n v_cmp_lt_f64_e64 s[42:43], %1, %4\n v_cmp_lt_f64_e64 s[40:41], %2, %4\n v_cmp_lt_f64_e64 s[42:43], %3, %4\n")
KERN64(v_cmp_class_f64, "v_cmp_class_f64_e64 s[40:41], %0, 3\n v_cmp_class_f64_e64 s[42:43], %1, 3\n v_cmp_class_f64_e64 s[40:41], %2, 3\n v_cmp_class_f64_e64 s[42:43], %3, 3\n")
KERN32(v_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n v_cndmask_b32_e64 %2, %2, %4, s[42:43]\n v_cndmask_b32_e64 %3, %3, %4, s[42:43]\n")
KERN32(v_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERN32(v_cmp_cndmask_s, "v_cmp_lt_u32_e64 s[40:41], %0, %4\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n v_cmp_lt_u32_e64 s[42:43], %2, %4\n v_cndmask_b32_e64 %3, %3, %4, s[42:43]\n")
KERN32(v_addc_vcc, "v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n")
KERN32(v_add_u32_sgpr, "v_add_u32 %0, %6, %0\n v_add_u32 %1, %6, %1\n v_add_u32 %2, %6, %2\n v_add_u32 %3, %6, %3\n")
KERN32(v_add_u32_inl, "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n")
KERN32(v_or_b32, "v_or_b32 %0, %4, %0\n v_or_b32 %1, %4, %1\n v_or_b32 %2, %4, %2\n v_or_b32 %3, %4, %3\n")
KERN32(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n")
KERN32(v_sub_u32, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n")
KERN32(v_min_u32, "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n")
KERN32(v_add_f32, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
KERN32(v_fmac_f32, "v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4\n")
KERN32(v_bcnt, "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3\n")
// mixes: does a 2.4-cycle op hide behind a 4.2-cycle one?  (A B A B ...) and SALU interleaved with VALU
KERN32(mix_xor_mul, "v_xor_b32 %0, %4, %0\n v_mul_lo_u32 %1, %1, %4\n v_xor_b32 %2, %4, %2\n v_mul_lo_u32 %3, %3, %4\n")
KERN32(mix_valu_salu, "v_xor_b32 %0, %4, %0\n s_add_u32 s40, s40, 1\n v_xor_b32 %2, %4, %2\n s_add_u32 s41, s41, 1\n")
KERN32(mix_mul_salu, "v_mul_lo_u32 %0, %0, %4\n s_add_u32 s40, s40, 1\n v_mul_lo_u32 %2, %2, %4\n s_add_u32 s41, s41, 1\n")
KERN32(s_add_u32, "s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n")
KERN32(s_nop0, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
__global__ __launch_bounds__(64) void k_ds_read_b128(double *out, double b, int c) {
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = i;
    __syncthreads();
    double x0 = 0, x1 = 0, x2 = 0, x3 = 0; int a = (threadIdx.x * 16 + c * 16) & 4095;
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        asm volatile(R8("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n") "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(*(double2 *)&x0), "=&v"(*(double2 *)&x2) : "v"(a));
    }
    out[threadIdx.x + 64 * (blockIdx.x & 1)] = x0 + x1 + x2 + x3;
}
__global__ __launch_bounds__(64) void k_ds_write_b64(double *out, double b, int c) {
    __shared__ double s[1024];
    double x0 = b; int a = (threadIdx.x * 8 + c * 8) & 4095;
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
        asm volatile(R8("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:512\n ds_write_b64 %1, %0 offset:1024\n ds_write_b64 %1, %0 offset:1536\n") "s_waitcnt lgkmcnt(0)\n"
                     :: "v"(x0), "v"(a) : "memory");
    }
    out[threadIdx.x + 64 * (blockIdx.x & 1)] = s[threadIdx.x];
}
template <typename K> void run(const char *name, K kern, int per_iter = 32) {
    double *out; (void)hipMalloc(&out, 128 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int nb : {1024, 4096, 8192}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(64), 0, 0, out, 1.0000001, 3);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double instr_per_simd = (double)REP * per_iter * nb / 1024.0;
        printf("%-18s %d waves per SIMD: %7.3f ms  => %5.2f SIMD cycles per wave-instruction (2.39 GHz)\n", name, nb / 1024, best, best * 1e-3 * 2.39e9 / instr_per_simd);
    }
    (void)hipFree(out);
}
#define RUN(N) run(#N, k_##N)
int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int part = argc > 1 ? atoi(argv[1]) : 0;
    if (part == 0 || part == 1) {
    RUN(v_add3_u32); RUN(v_lshl_add_u32); RUN(v_alignbit_b32); RUN(v_bfi_b32); RUN(v_and_or_b32); RUN(v_lshl_or_b32); RUN(v_bitop3_b32); RUN(v_perm_b32); RUN(v_bfe_u32);
    RUN(v_xor_b32_e64); RUN(v_xor_b32_sgpr); RUN(v_xor_b32_lit); RUN(v_lshrrev_b32); RUN(v_and_b32); RUN(v_mov_b32); RUN(v_mbcnt_lo); RUN(v_mbcnt_hi); RUN(v_lshl_sdwa);
    RUN(v_add_co_pair); RUN(v_cmp_u32_vcc); RUN(v_cmp_u32_sgpr); RUN(v_cndmask_vcc); RUN(v_readfirstlane); RUN(v_mul_lo_sgpr); RUN(v_mad_u32_u24); RUN(v_mad_u32_u16);
    }
    if (part == 0 || part == 2) {
    RUN(v_cndmask_sgpr); RUN(v_cmp_cndmask); RUN(v_cmp_cndmask_s); RUN(v_addc_vcc); RUN(v_add_u32_sgpr); RUN(v_add_u32_inl); RUN(v_or_b32); RUN(v_lshlrev_b32); RUN(v_sub_u32); RUN(v_min_u32); RUN(v_add_f32); RUN(v_fmac_f32); RUN(v_bcnt);
    RUN(v_fma_f32); RUN(v_mul_f32); RUN(v_cvt_f32_u32);
    }
    if (part == 0 || part == 3) {
    RUN(v_mov_dpp_rowshr); RUN(v_add_u32_dpp);
    RUN(v_mov_b64); RUN(v_lshl_add_u64); RUN(v_pk_add_f32); RUN(v_pk_mul_f32); RUN(v_cmp_u64_sgpr); RUN(v_cmp_f64_sgpr); RUN(v_cmp_class_f64);
    }
    if (part == 0 || part == 4) {
    RUN(mix_xor_mul); RUN(mix_valu_salu); RUN(mix_mul_salu); RUN(s_add_u32); RUN(s_nop0);
    }
    if (part == 0 || part == 5) {
    run("ds_read_b128", k_ds_read_b128); run("ds_write_b64", k_ds_write_b64);
    }
    return 0;
}
