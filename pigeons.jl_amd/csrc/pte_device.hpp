// pte_device.hpp -- gfx950 device helpers: per-replica counter-based RNG streams,
// Julia-compatible samplers evaluated wave-parallel, and the fixed pairwise
// reduction tree shared by every kernel.
//
// One wavefront (64 lanes) owns one replica.  "Uniform" below means: the value is
// identical in all 64 lanes (the compiler may or may not keep it in SGPRs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZIG_TABLE_ATTR __device__
#include "zig_tables.h"
#include "../../include/pte_rng_policy.h"

namespace pte {

constexpr uint64_t MASK52 = 0x000fffffffffffffULL;

// include/pte_rng_policy.h: the conventions of Julia's Random that no fixture pins yet, one word per device, set by
// pte_set_rng_policy (hipMemcpyToSymbol).  Read on the ziggurat tails (~3e-4 of the draws) and by the Bernoulli refresh only.
static __device__ unsigned g_rng_policy = PTE_RNG_POLICY_DEFAULT;      // (one copy per translation unit: pte_set_rng_policy writes every one)
__device__ __forceinline__ double zig_tail_neglog(double u) { return (g_rng_policy & PTE_RNG_TAIL_LOG1P) ? -log1p(-u) : -log(u); }
__device__ __forceinline__ unsigned rng_bool_bit() { return PTE_RNG_POLICY_BOOL_BIT(g_rng_policy); }

// ---- SplittableRandom (SplittableRandoms.jl 0.1 == Java SplittableRandom) --------------------
// The k-th output of a stream is mix64(seed0 + k*gamma): counter based, so 64 lanes can
// evaluate 64 consecutive draws of ONE replica's stream in one shot.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t mix_gamma(uint64_t z) {
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL;
    z = (z ^ (z >> 33)) | 1ULL;
    int n = __popcll(z ^ (z >> 1));
    return (n < 24) ? (z ^ 0xaaaaaaaaaaaaaaaaULL) : z;
}
// rand(rng)::Float64 of Julia's generic AbstractRNG path: [1,2) from the low 52 bits, minus 1.
__device__ __forceinline__ double u52_to_unit(uint64_t u) {
    return __longlong_as_double((long long)((u & MASK52) | 0x3ff0000000000000ULL)) - 1.0;
}

// ---- cross-lane helpers -----------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ double readlane_f64(double v, int lane /*uniform*/) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane /*uniform*/) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double writelane_f64(double old, double v /*uniform*/, int lane /*uniform*/) {
    return (lane_id() == lane) ? v : old;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

// ---- the fixed reduction tree -----------------------------------------------------------------
// sqr_norm(x) (reference src/utils/misc.jl:10) is evaluated as the balanced binary tree over the
// leaves x_i^2 in natural order, zero padded to a power of two: node[i] = node[2i] + node[2i+1].
// With lane l holding leaf (64*b + l) of block b, the six in-block levels are an xor butterfly:
// after level k every lane holds the sum of its 2^(k+1)-aligned group (a+b == b+a bitwise, so both
// partners agree).  U[k] = node values of level k (U[0] = leaves), U[6] = block sum.
__device__ __forceinline__ void butterfly6(double leaf, double (&U)[7]) {
    U[0] = leaf;
#pragma unroll
    for (int k = 0; k < 6; ++k) U[k + 1] = U[k] + shfl_xor_f64(U[k], 1 << k);
}
__device__ __forceinline__ double wave_tree_sum64(double leaf) {
    double v = leaf;
#pragma unroll
    for (int k = 0; k < 6; ++k) v = v + shfl_xor_f64(v, 1 << k);
    return v;
}
// Root of the tree over the block sums: lane b holds block sum b (0 for b >= B).  NLU = log2 of
// the padded block count.  Result uniform.
template <int NLU>
__device__ __forceinline__ double upper_tree_root(double bs) {
    double v = bs;
#pragma unroll
    for (int q = 0; q < NLU; ++q) v = v + shfl_xor_f64(v, 1 << q);
    return readlane_f64(v, 0);
}
__device__ __forceinline__ double upper_tree_root_dyn(double bs, int nlu) {
    double v = bs;
    for (int q = 0; q < nlu; ++q) v = v + shfl_xor_f64(v, 1 << q);
    return readlane_f64(v, 0);
}

// ---- DPP wave reduction with the association of the fixed tree; result uniform -----------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double v) {
    // Full-mask permutations have a source for every lane: bound_ctrl tells the compiler that no lane keeps the `old`
    // operand, which saves initialising it.  The row-broadcast steps leave whole rows untouched: those must read +0.0.
    constexpr bool full = (ROW_MASK == 0xF);
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, full);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, full);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add_step<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]  : lanes l, l^1
    v = dpp_add_step<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]  : l, l^2
    v = dpp_add_step<0x141, 0xF>(v);    // row_half_mirror      : other group of 4 (uniform inside groups)
    v = dpp_add_step<0x140, 0xF>(v);    // row_mirror           : other group of 8
    v = dpp_add_step<0x142, 0xA>(v);    // row_bcast15 -> rows 1,3 (disabled rows add +0.0)
    v = dpp_add_step<0x143, 0xC>(v);    // row_bcast31 -> rows 2,3
    return readlane_f64(v, 63);
}
// M independent wave sums taken in lockstep: the same operations per sum as wave_sum_dpp (bit-identical), but every level of the tree
// is issued for all M sums before the next level, so that the wait states of one chain (a DPP move reads what the add before it
// wrote: s_nop, plus the dependent FP64 latency) are filled by the others.  One dependent sum costs a wave alone on its SIMD ~180
// cycles (tools/ubench/wave_sum_all.hip); hipcc issues consecutive wave_sum_dpp calls one after the other.
template <int M>
__device__ __forceinline__ void wave_sum_dpp_multi(double (&v)[M]) {
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0xB1, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x4E, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x141, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x140, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x142, 0xA>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x143, 0xC>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = readlane_f64(v[j], 63);
}
// gfx950 row / half exchanges between two registers:
//   permlane16_swap(a, b): a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]   (r = rows of 16 lanes)
//   permlane32_swap(a, b): a' = [a.lo, b.lo],             b' = [a.hi, b.hi]               (halves of 32 lanes)
// so that a' + b' is the next level of the fixed tree for BOTH operands at once, packed into one register.
__device__ __forceinline__ void permlane16_swap_f64(double &a, double &b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void permlane32_swap_f64(double &a, double &b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// M = 4 or 8 wave sums, v[2i] and v[2i+1] being two 64-blocks whose totals the tree adds next: out[i] = sum(v[2i]) + sum(v[2i+1]).
// Every addition is one the fixed tree makes (same operand pairs as wave_sum_dpp, then the block pair); what changes is how many
// registers carry the partial sums.  A level of the tree leaves the same value in both lanes (quads, ...) it combined, so after level k
// only one lane in 2^k needs to keep a chain's partial sum and the others can carry ANOTHER chain's:
//   level 1 (l, l^1), per chain                         then chains 2i / 2i+1 share a register: even lanes one, odd lanes the other (a select)
//   level 2 (l, l^2), per register (parity kept)        then two registers share one: a quad holds four chains [c0 c1 c2 c3] (a select)
//   level 3 / 4: row_shr:4 / row_shr:8 (lane & 3 kept)  -> the quad's four totals of a row in lanes 12..15 of the row
//   level 5 (rows 0+1, 2+3): permlane16_swap of the two registers (M = 8) / of the register with itself (M = 4)
//   level 6 (halves): permlane32_swap with itself;  block pair c0 + c1, c2 + c3: quad_perm [1,0,3,2]
// 73 vector instructions for eight chains (round 3, packing from the rows upwards only: 117; one chain at a time: 164), 43 for four (62).
#ifndef PTE_WSP_ROWS_ONLY
// lanes whose bit is set in `mask` take b, the others a: two v_cndmask_b32 on a scalar pair.  (Written as `odd ? v[2 * i + 1] : v[2 * i]`
// hipcc turns the select into an INDEXED access of the array -- through LDS / scratch -- and the reduction takes twice as long as before.)
__device__ __forceinline__ double select_lanes_f64(unsigned long long mask, double a, double b) {
    int lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(a)), "v"(__double2loint(b)), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(a)), "v"(__double2hiint(b)), "s"(mask));
    return __hiloint2double(hi, lo);
}
// the two 64-blocks of ONE pair: sum(a) + sum(b), uniform.  30 vector instructions (two wave_sum_dpp + the add: 37)
__device__ __forceinline__ double wave_sum_pair(double a, double b) {
    a = dpp_add_step<0xB1, 0xF>(a); b = dpp_add_step<0xB1, 0xF>(b);
    double m = select_lanes_f64(0xAAAAAAAAAAAAAAAAull, a, b);          // even lanes a, odd lanes b
    m = dpp_add_step<0x4E, 0xF>(m);
    m = dpp_add_step<0x114, 0xF>(m);
    m = dpp_add_step<0x118, 0xF>(m);
    double t = m;
    permlane16_swap_f64(m, t);
    double w = m + t, t2 = w;
    permlane32_swap_f64(w, t2);
    double z = w + t2;
    z = dpp_add_step<0xB1, 0xF>(z);
    return readlane_f64(z, 12);
}
template <int M>
__device__ __forceinline__ void wave_sum_pairs(double (&v)[M], double (&out)[M / 2]) {
    static_assert(M == 4 || M == 8, "wave_sum_pairs: 4 or 8 chains");
    constexpr unsigned long long ODD = 0xAAAAAAAAAAAAAAAAull, HI2 = 0xCCCCCCCCCCCCCCCCull;      // lanes with bit 0 / bit 1 set
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0xB1, 0xF>(v[j]);
    double m[M / 2];
#pragma unroll
    for (int i = 0; i < M / 2; ++i) m[i] = select_lanes_f64(ODD, v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < M / 2; ++i) m[i] = dpp_add_step<0x4E, 0xF>(m[i]);
    double n[M / 4];
#pragma unroll
    for (int i = 0; i < M / 4; ++i) n[i] = select_lanes_f64(HI2, m[2 * i], m[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < M / 4; ++i) n[i] = dpp_add_step<0x114, 0xF>(n[i]);     // row_shr:4  (quads 1 and 3 of a row hold q0 + q1, q2 + q3)
#pragma unroll
    for (int i = 0; i < M / 4; ++i) n[i] = dpp_add_step<0x118, 0xF>(n[i]);     // row_shr:8  (quad 3: the row's totals)
    double w;
    if constexpr (M == 8) {
        permlane16_swap_f64(n[0], n[1]);               // n0' = [a.r0, b.r0, a.r2, b.r2], n1' = [a.r1, b.r1, a.r3, b.r3]
        w = n[0] + n[1];                               // rows: a.r01, b.r01, a.r23, b.r23
    } else {
        double t = n[0];
        permlane16_swap_f64(n[0], t);                  // [r0, r0, r2, r2], [r1, r1, r3, r3]
        w = n[0] + t;
    }
    double t2 = w;
    permlane32_swap_f64(w, t2);                        // [w.lo, w.lo], [w.hi, w.hi]
    double z = w + t2;                                 // rows 0 / 1 (and 2 / 3): chains 0..3 / 4..7 in lanes 12..15
    z = dpp_add_step<0xB1, 0xF>(z);                    // the block pairs
    out[0] = readlane_f64(z, 12); out[1] = readlane_f64(z, 14);
    if constexpr (M == 8) { out[2] = readlane_f64(z, 28); out[3] = readlane_f64(z, 30); }
}
// FOUR independent wave sums, every one the fixed tree's (the operand pairs of wave_sum_dpp; a + b == b + a), packed as in wave_sum_pairs and without its
// last step: after level 1 two sums share a register (odd lanes the second), after level 2 four (lane & 3), so levels 3-6 are paid once instead of four times.
// 48 vector instructions against 80 for wave_sum_dpp_multi<4> (round 6, the four-wave Langevin kernels).  NZ: how many of v[] are not known zeros
// (NZ = 3: v[3] == +0.0 everywhere -- its first level is skipped: 0 + 0).
template <int NZ = 4>
__device__ __forceinline__ void wave_sum_packed4(double (&v)[4], double (&out)[4]) {
    constexpr unsigned long long ODD = 0xAAAAAAAAAAAAAAAAull, HI2 = 0xCCCCCCCCCCCCCCCCull;      // lanes with bit 0 / bit 1 set
#pragma unroll
    for (int j = 0; j < NZ; ++j) v[j] = dpp_add_step<0xB1, 0xF>(v[j]);
    double m0 = select_lanes_f64(ODD, v[0], v[1]), m1 = select_lanes_f64(ODD, v[2], v[3]);
    m0 = dpp_add_step<0x4E, 0xF>(m0); m1 = dpp_add_step<0x4E, 0xF>(m1);
    double n = select_lanes_f64(HI2, m0, m1);
    n = dpp_add_step<0x114, 0xF>(n);                   // row_shr:4
    n = dpp_add_step<0x118, 0xF>(n);                   // row_shr:8: lanes 12..15 of a row hold the row's totals of sums 0..3
    double t = n;
    permlane16_swap_f64(n, t);                         // [r0, r0, r2, r2], [r1, r1, r3, r3]
    double w = n + t, t2 = w;
    permlane32_swap_f64(w, t2);                        // [lo, lo], [hi, hi]
    const double z = w + t2;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = readlane_f64(z, 12 + k);
}
#else       // round 3's form (A/B builds): the partial sums of two chains share a register from the rows upwards only
template <int M>
__device__ __forceinline__ void wave_sum_pairs(double (&v)[M], double (&out)[M / 2]) {
    static_assert(M == 4 || M == 8, "wave_sum_pairs: 4 or 8 chains");
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0xB1, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x4E, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x141, 0xF>(v[j]);
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = dpp_add_step<0x140, 0xF>(v[j]);
    double w[M / 2];
#pragma unroll
    for (int i = 0; i < M / 2; ++i) { permlane16_swap_f64(v[2 * i], v[2 * i + 1]); w[i] = v[2 * i] + v[2 * i + 1]; }
    double u[M / 4];
#pragma unroll
    for (int i = 0; i < M / 4; ++i) { permlane32_swap_f64(w[2 * i], w[2 * i + 1]); u[i] = w[2 * i] + w[2 * i + 1]; }
    if constexpr (M == 8) {
        permlane16_swap_f64(u[0], u[1]);               // u0' = [A, E, C, G], u1' = [B, F, D, H]   (A..D = chains 0..3 of u0, E..H = 4..7)
        const double z = u[0] + u[1];                  // rows: pair 0, pair 2, pair 1, pair 3
        out[0] = readlane_f64(z, 0); out[2] = readlane_f64(z, 16);
        out[1] = readlane_f64(z, 32); out[3] = readlane_f64(z, 48);
    } else {
        double t = u[0];
        permlane16_swap_f64(u[0], t);                  // u0' = [A, A, C, C], t' = [B, B, D, D]
        const double z = u[0] + t;
        out[0] = readlane_f64(z, 0); out[1] = readlane_f64(z, 32);
    }
}
#endif
// ---- IEEE quotients by a divisor that stays the same for many divisions ---------------------------
// (the preconditioner's diagonal for a whole scan, the funnel's sigma for one evaluation: round 6)
// a / b as q = a r, q' = fma(fma(-q, b, a), r, q) with r = RN(1 / b): CORRECTLY ROUNDED (Markstein 1990: r the correctly rounded reciprocal, no over /
// underflow on the way, b's significand not all ones) -- the same bits as the division the reference makes, in 3 instructions instead of ~13 with ~10
// temporaries.  What the theorem does not cover takes the division itself, decided per vector by the caller (a uniform branch): a quotient estimate
// outside [2^-900, 2^901) -- zero, subnormal, infinite, NaN -- in a lane that holds a coordinate, or an excluded divisor (extreme exponent, all-ones
// significand).  pte_test_quotient holds the procedure to a / b on the host; the kernels that use it are held to kernels that divide
// (tests/test_gpu_langevin_mw.py) and to the oracle.
__device__ __forceinline__ bool markstein_divisor_ok(double b) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(b);
    const int ex = (int)((u >> 52) & 0x7FF);
    return (u & MASK52) != MASK52 && ex > 1023 - 500 && ex < 1023 + 500;
}
// (the estimate's exponent field inside [123, 1923] = magnitude in [2^-900, 2^901): zero and subnormals have field 0, infinities and NaN 2047 -- one v_bfe_u32)
__device__ __forceinline__ unsigned quotient_exponent_field(double q) { return __builtin_amdgcn_ubfe((unsigned)__double2hiint(q), 20u, 11u); }
__device__ __forceinline__ bool quotient_in_range(double q) { const unsigned ex = quotient_exponent_field(q); return ex >= 123u && ex <= 1923u; }
__device__ __forceinline__ double markstein_quotient(double a, double b, double rinv, double &q_estimate) {
    const double q = a * rinv;
    q_estimate = q;
    return __builtin_fma(__builtin_fma(-q, b, a), rinv, q);
}

// ---- sequential (uniform) stream: used on slow paths and for single draws ---------------------
struct SeqRng {
    uint64_t seed, gamma;   // uniform
    __device__ __forceinline__ uint64_t next() { seed += gamma; return mix64(seed); }
    __device__ __forceinline__ double rand() { return u52_to_unit(next()); }
};

// randn slow path (Julia Random/src/normal.jl `randn_unlikely`), given the failed first draw.
// (tables: global by default; a caller that staged them in LDS passes its copies -- a global gather is a memory round trip per event)
__device__ inline double randn_seq(SeqRng &r, const double *wi = ZIG_WI, const unsigned long long *ki = ZIG_KI, const double *fi = ZIG_FI);
__device__ inline double randn_unlikely(SeqRng &r, int idx, int64_t rabs, double x,
                                        const double *wi = ZIG_WI, const unsigned long long *ki = ZIG_KI, const double *fi = ZIG_FI) {
    if (idx == 0) {
        for (;;) {
            double xx = ZIG_NOR_INV_R * zig_tail_neglog(r.rand());
            double yy = zig_tail_neglog(r.rand());
            if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
        }
    } else if ((fi[idx - 1] - fi[idx]) * r.rand() + fi[idx] < exp(-0.5 * x * x)) {
        return x;
    }
    return randn_seq(r, wi, ki, fi);
}
__device__ inline double randn_seq(SeqRng &r, const double *wi, const unsigned long long *ki, const double *fi) {
    for (;;) {
        uint64_t u = r.next() & MASK52;
        int64_t rabs = (int64_t)(u >> 1);
        int idx = (int)(rabs & 0xFF);
        double x = (double)((u & 1) ? -rabs : rabs) * wi[idx];
        if ((uint64_t)rabs < ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = ZIG_NOR_INV_R * zig_tail_neglog(r.rand());
                double yy = zig_tail_neglog(r.rand());
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
            }
        } else if ((fi[idx - 1] - fi[idx]) * r.rand() + fi[idx] < exp(-0.5 * x * x)) {
            return x;
        }
    }
}
// randexp (Julia `randexp` + `randexp_unlikely`), sequential.
__device__ inline double randexp_from_raw(SeqRng &r, uint64_t raw) {
    for (;;) {
        uint64_t ri = raw & MASK52;
        int idx = (int)(ri & 0xFF);
        double x = (double)ri * ZIG_WE[idx];
        if (ri < ZIG_KE[idx]) return x;
        if (idx == 0) return ZIG_EXP_R + zig_tail_neglog(r.rand());
        if ((ZIG_FE[idx - 1] - ZIG_FE[idx]) * r.rand() + ZIG_FE[idx] < exp(-x)) return x;
        raw = r.next();
    }
}
__device__ inline double randexp_seq(SeqRng &r) { return randexp_from_raw(r, r.next()); }

// ---- wave-parallel block of normals in the reference's sequential draw order -------------------
// Produces randn #0..n_valid-1 of the stream (lane l gets output l) and advances `r` exactly as
// n_valid sequential randn(rng) calls would.  99.3 % of draws take the one-draw fast path, so all
// lanes draw speculatively at consecutive counters; the first lane that needs the slow path is
// resolved sequentially (it may consume extra draws) and the lanes after it are re-drawn.
__device__ inline double wave_randn_block(SeqRng &r, int lane, int n_valid /*uniform, <= 64*/,
                                          const double *wi = ZIG_WI, const unsigned long long *ki = ZIG_KI, const double *fi = ZIG_FI) {
    double out = 0.0;
    int start = 0;
    while (start < n_valid) {
        uint64_t u = mix64(r.seed + (uint64_t)(int64_t)(lane - start + 1) * r.gamma) & MASK52;
        int64_t rabs = (int64_t)(u >> 1);
        int idx = (int)(rabs & 0xFF);
        double x = (double)((u & 1) ? -rabs : rabs) * wi[idx];            // tables: global, or staged in LDS by the caller
        bool active = (lane >= start) && (lane < n_valid);
        bool ok = (uint64_t)rabs < ki[idx];
#ifdef PTE_MEASURE_ALWAYS_FAST      // measurement builds only (tools/bench_toy.py): what the fast path alone would cost -- changes the samples
        ok = true;
#endif
        uint64_t failmask = ballot64(active && !ok);
        int f = failmask ? (int)__builtin_ctzll(failmask) : n_valid;
        if (active && lane < f) out = x;
        if (f >= n_valid) {
            r.seed += (uint64_t)(n_valid - start) * r.gamma;
            break;
        }
        r.seed += (uint64_t)(f - start + 1) * r.gamma;        // up to and incl. lane f's first draw
        int idx_f = __builtin_amdgcn_readlane(idx, f);
        int64_t rabs_f = (int64_t)readlane_u64((uint64_t)rabs, f);
        double x_f = readlane_f64(x, f);
        double xf = randn_unlikely(r, idx_f, rabs_f, x_f, wi, ki, fi);
        if (lane == f) out = xf;
        start = f + 1;
    }
    return out;
}

// ---- 64 buffered raw draws for a uniform sequential consumer ----------------------------------
struct WaveDraws {
    uint64_t seed, gamma;   // uniform: stream state BEFORE draw #0 of the buffer
    uint64_t D;             // per lane: raw draw #lane of the buffer
    int p;                  // uniform: next unread draw
    __device__ __forceinline__ void init(uint64_t s, uint64_t g, int lane) {
        seed = s; gamma = g; p = 0;
        D = mix64(seed + (uint64_t)(lane + 1) * gamma);
    }
    __device__ __forceinline__ uint64_t next_raw(int lane) {
        if (p == 64) { seed += 64ull * gamma; p = 0; D = mix64(seed + (uint64_t)(lane + 1) * gamma); }
        uint64_t v = readlane_u64(D, p);
        p += 1;
        return v;
    }
    __device__ __forceinline__ double rand(int lane) { return u52_to_unit(next_raw(lane)); }
    __device__ __forceinline__ uint64_t final_seed() const { return seed + (uint64_t)p * gamma; }
    // hand the stream to a sequential consumer (slow paths) and take it back
    __device__ __forceinline__ SeqRng to_seq() const { return SeqRng{final_seed(), gamma}; }
    __device__ __forceinline__ void from_seq(const SeqRng &s, int lane) { init(s.seed, s.gamma, lane); }
};

}  // namespace pte
