// pte_slice_common.hpp -- pieces shared by the SliceSampler kernels: the 64-draw pre-converted buffer,
// mask selects, and the (debug-build only) section stopwatch.
#pragma once
#include "pte_kernels.hpp"

namespace pte {

struct DrawBuf {
    uint64_t seed, gamma;   // uniform: stream state before draw #0 of the buffer
    double unit;            // per lane: rand() of draw #lane
    double ex;              // per lane: randexp() fast-path value of draw #lane
    uint64_t exok;          // uniform: bit l set <=> draw #l passes the exponential ziggurat fast test
    int p;                  // uniform: next unread draw

    __device__ __forceinline__ void fill(int lane, const double *s_we, const unsigned long long *s_ke) {
        uint64_t r = mix64(seed + (uint64_t)(lane + 1) * gamma);
        unit = u52_to_unit(r);
        uint64_t ri = r & MASK52;
        int idx = (int)(ri & 0xFF);
        ex = (double)ri * s_we[idx];
        exok = ballot64(ri < s_ke[idx]);
        p = 0;
    }
    __device__ __forceinline__ void init(uint64_t s, uint64_t g, int lane, const double *s_we, const unsigned long long *s_ke) {
        seed = s; gamma = g;
        fill(lane, s_we, s_ke);
    }
    // make sure draws p .. p+k-1 are in the buffer
    __device__ __forceinline__ void ensure(int k, int lane, const double *s_we, const unsigned long long *s_ke) {
        if (p + k > 64) { seed += (uint64_t)p * gamma; fill(lane, s_we, s_ke); }
    }
    __device__ __forceinline__ double rand(int lane, const double *s_we, const unsigned long long *s_ke) {
        ensure(1, lane, s_we, s_ke);
        double u = readlane_f64(unit, p);
        p += 1;
        return u;
    }
    // randexp(rng): fast path from the buffer, slow path sequentially on the same stream
    __device__ __forceinline__ double randexp(int lane, const double *s_we, const unsigned long long *s_ke) {
        ensure(1, lane, s_we, s_ke);
        if ((exok >> p) & 1ull) {
            double v = readlane_f64(ex, p);
            p += 1;
            return v;
        }
        SeqRng s{seed + (uint64_t)(p + 1) * gamma, gamma};
        double v = randexp_from_raw(s, mix64(s.seed));
        seed = s.seed;
        fill(lane, s_we, s_ke);
        return v;
    }
    // randexp(rng) when the caller has already ensured the draw is buffered (k more are wanted after it)
    __device__ __forceinline__ double randexp_ensured(int k_after, int lane, const double *s_we, const unsigned long long *s_ke) {
        if (__builtin_expect((exok >> p) & 1ull, 1)) {
            double v = readlane_f64(ex, p);
            p += 1;
            return v;
        }
        SeqRng s{seed + (uint64_t)(p + 1) * gamma, gamma};
        double v = randexp_from_raw(s, mix64(s.seed));
        seed = s.seed;
        fill(lane, s_we, s_ke);          // p = 0: a full buffer of 64 draws >= k_after
        (void)k_after;
        return v;
    }
    __device__ __forceinline__ uint64_t final_seed() const { return seed + (uint64_t)p * gamma; }
};

// (m ? a : b) with m = 0 / -1.  Written as mask-and-merge; the optimiser canonicalises it to
// v_cmp + v_cndmask (measured 102 vs 94 cycles per proposal step against hand-placed v_bfi, but
// inline asm would make every derived value "divergent" for the compiler and spill the uniform
// control state into VGPRs / exec-mask loops -- a much larger loss).
__device__ __forceinline__ double bitsel(int m, double a, double b) {
    const int lo = (m & __double2loint(a)) | (~m & __double2loint(b));
    const int hi = (m & __double2hiint(a)) | (~m & __double2hiint(b));
    return __hiloint2double(hi, lo);
}
// -1 if x < 0 (sign bit set), else 0
__device__ __forceinline__ int neg_mask(double x) { return __double2hiint(x) >> 31; }

// tools/prof_sections*.py build with -DPTE_PROFILE_SECTIONS: cycle-counter stamps around kernel sections
#ifdef PTE_PROFILE_SECTIONS
#define PROF_T(v) __builtin_amdgcn_sched_barrier(0); const long long v = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define PROF_ADD(i, x) prof[i] += (x)
#else
#define PROF_T(v)
#define PROF_ADD(i, x)
#endif

}  // namespace pte
