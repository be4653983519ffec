// pte_automala_mw.hpp -- k_explore_langevin_mw: AutoMALA / MALA for 512 < d <= 1024 with FOUR wavefronts per replica (round 6;
// VERDICT r05 item 4).  Reference: src/explorers/AutoMALA.jl:84-275, src/explorers/MALA.jl:74-97,
// src/explorers/hamiltonian_dynamics.jl:40-84, src/explorers/Preconditioner.jl:57-77 -- the same procedure, draw for draw and
// addition for addition, as k_explore_automala (pte_automala.hpp), whose E = 16 instantiations (one wave holding sixteen 64-coordinate
// blocks of twelve vectors: 250-300 spilled VGPRs, one wave per SIMD) this kernel replaces in the product build.
//
// Leapfrog, gradient and log joint are elementwise + tree reductions, so a replica can own a 256-thread workgroup: wave w holds coordinates
// 256 w .. 256 w + 255, lane l FOUR CONSECUTIVE ones (its register j <-> coordinate 256 w + 4 l + j).  Why the results do not change by a bit:
//   * a reduction is the SAME fixed tree (pte_device.hpp: balanced, leaves in natural order).  With four consecutive leaves per lane its first two
//     levels are in-lane additions -- (t0 + t1) + (t2 + t3) -- the next six the xor butterfly over the 64 lanes (wave_sum_dpp: one register per
//     sum, where the one-wave layout -- one leaf per lane and block -- pays six levels for each of its blocks), which gives the 256-leaf node W_w;
//     the top two levels -- (W0 + W1) + (W2 + W3) -- are added by EVERY wave from the four partial sums exchanged through LDS: all waves hold
//     the same root, to the bit, so every branch of the algorithm (step-size search, finiteness checks, accept / reject) is taken identically
//     by the four waves without further talk;
//   * the replica's stream is consumed in the reference's order.  The momentum (d sequential randn) is the stream compaction of
//     pte_normals.hpp cut over the four waves: wave w evaluates the 320 stream positions of ITS segment (1280 = 1024 + slack) as if the
//     ziggurat's fast path applied, resolves the 1.5 % that leave it in one divergent pass, marks the positions those attempts consume;
//     the waves exchange how many positions each segment consumed, and every position that was not consumed is an output -- its index is
//     its position minus the consumed positions before it -- scattered to LDS in output order.  An attempt at a segment's last position
//     consumes the next segment's first (followed, when that position is no event itself); a longer spill, a tail that runs on, 32 or
//     more events in a segment: that refresh takes the plain sequential procedure instead (sixteen blocks, the stream handed from wave to
//     wave), which is what the fast path must equal anyway.  Every other draw (the preconditioner's, the two bounds, the accept uniform)
//     is taken by all four waves from identical copies of the stream;
//   * the funnel's first coordinate (its scale enters every term) is broadcast by the lane that owns it;
//   * a / M and x / sigma are the IEEE quotients (see div_M below): Markstein's correctly rounded q' = fma(fma(-q, b, a), r, q), the division
//     itself wherever the theorem's conditions are not met.
// Storage: a wave keeps x, p, g, g0, M, 1 / M and the forward search's restore copy of p in registers (7 vectors x 8 VGPRs); the start
// state, the kept trial's state and momentum (its conditioned gradient is RE-EVALUATED from the state: elementwise, plus one saved word for
// the funnel's first coordinate) and the conditioned gradient at the start live in LDS (32 KB per replica; each lane reads back only what
// it wrote: no barrier; the momentum's workspace uses 19 KB of the same bytes while no trial is kept).  <= 128 VGPRs and 38.4 KB of LDS:
// four workgroups per compute unit = 1024 replicas resident, four waves per SIMD.  The search is ONE loop with one leapfrog in it.
#pragma once
#include "pte_automala.hpp"
#include "pte_normals.hpp"

namespace pte {

constexpr int MW_NWV = 4;                 // waves per replica
constexpr int MW_EW = 4;                  // 64-coordinate blocks per wave: 16 blocks = d <= 1024
constexpr int MW_DMAX = 64 * MW_NWV * MW_EW;

constexpr int MW_SEG = 320;               // stream positions per wave of the momentum's compaction (4 x 320 = 1024 outputs + 256 of slack: ~25 are consumed)
constexpr int MW_SLOTS = MW_SEG / 64;
constexpr int MW_MAX_EV = 32;             // events a wave resolves lane-parallel, two lanes each
constexpr int MW_EV_CAP = 128;
struct MwMomentum {                       // (lives in the bytes of MwLds::v[0 .. 2]: no trial is kept while a momentum is drawn)
    double seg[MW_NWV][MW_SEG];           // a segment's values by stream position; consumed positions marked with a NaN
    double out[MW_DMAX];                  // the momentum in output order
    unsigned short ev[MW_NWV][MW_EV_CAP]; // positions that left the fast path, in stream order
};
enum { MW_XK = 0, MW_PK = 1, MW_GS = 2, MW_XS = 3 };      // MwLds::v: kept trial's state, its momentum, conditioned gradient at the start point, start state
struct MwLds {
    double wi[256]; unsigned long long ki[256];                      // the normal ziggurat's fast-path tables
#ifndef PTE_MW_FI_GLOBAL
    double fi[256];                                                  // ... and the wedge test's
#endif
    union { double v[4][MW_DMAX]; MwMomentum mom; };
    alignas(16) double part[2][4][MW_NWV]; // the waves' partial sums of up to four reductions taken in lockstep, double-buffered; [sum][wave]: a sum's four words are two 16-byte reads
    double ybuf[2][4];                    // the funnel's first coordinate y and what every term needs of it: sigma = exp(y / 2), log sigma, 1 / sigma (evaluated by wave 0 alone)
    double bounds[2];                     // log of the two uniforms that bound the step-size search (evaluated by wave 0 alone)
    unsigned long long seed[MW_NWV];      // sequential procedure: stream position after wave w's momentum blocks
    int mom_cons[MW_NWV], mom_fail[MW_NWV], mom_below[MW_NWV];      // compaction: positions a segment consumed; what it cannot follow; consumed by the attempts of the outputs below d
    int prio;                             // per-scan launches: the priority this replica's waves ask for in the refresh that begins (by its pace against the launch's mean)
#ifdef PTE_MW_LDS_PAD                     // development builds only: where does the fourth workgroup of a compute unit stop fitting
    char pad[PTE_MW_LDS_PAD];
#endif
};
static_assert(sizeof(MwMomentum) <= sizeof(double) * 3 * MW_DMAX, "the momentum's workspace must not reach the start state");
static_assert(sizeof(MwLds) <= 40 * 1024, "four workgroups per compute unit");

using MwLdsP = __attribute__((address_space(3))) MwLds *;

// ---- cold paths as CALLED functions: the code a refresh walks through stays small (the kernel's instruction footprint, not its arithmetic, is
// what a lone wave per SIMD waits for) and the register allocation of the hot path is not bent around them
__device__ __attribute__((noinline)) double mw_div_exact(double a, double b) { return a / b; }
// The funnel's scale (wave 0, once per evaluation) as a called function too, though it is anything but cold: inlined into the trial loop, the ~20 polynomial
// coefficients of exp and log (they initialise the accumulators of the Horner steps, so they must sit in vector registers) are hoisted out of the loop as
// loop invariants, spilled, and read back from SCRATCH one by one, each behind its own s_waitcnt -- 17 exposed memory round trips per trial leapfrog on
// the path the other three waves wait for at the barrier.  Called, the constants are materialised where they are used.
struct MwScale3 { double sigma, logsigma, rinv; };
__device__ __attribute__((noinline)) MwScale3 mw_funnel_scale_eval(double y) {
    MwScale3 r;
    r.sigma = exp(y / 2.0);
    r.logsigma = log(r.sigma);
    r.rinv = 1.0 / r.sigma;
    return r;
}
__device__ __attribute__((noinline)) double mw_exp_exact(double t) { return exp(t); }
// (the refresh level's exp / log -- the acceptance ratio, the two bounds of the search -- for the same reason as mw_funnel_scale_eval: inlined, their
// coefficients are loop invariants of the refresh loop: six vector registers held for the whole kernel and 6 + 19 reads from scratch per refresh)
__device__ __attribute__((noinline)) double mw_log_exact(double t) { return log(t); }
#ifndef PTE_MW_REFRESH_MATH_CALLED
#define PTE_MW_REFRESH_MATH_CALLED 1
#endif
#if PTE_MW_REFRESH_MATH_CALLED
#define MW_EXP(x) mw_exp_exact(x)
#define MW_LOG(x) mw_log_exact(x)
#else
#define MW_EXP(x) exp(x)
#define MW_LOG(x) log(x)
#endif
// tail of the normal ziggurat (Random/src/normal.jl randn_unlikely, idx == 0) for the two lanes of an event's pair, exactly, at stream position zt:
// trial k takes the draws 2k - 1 (-> xx) and 2k (-> yy) behind the event; the even lane evaluates the first, the odd lane the second.  Returns xx; *pairs_out = trials
__device__ __attribute__((noinline)) double mw_tail_event(uint64_t zt, const uint64_t gamma, const int role, const bool tail_log1p, const int max_pairs, int *pairs_out) {
    int pairs = 0;
    double xx, yy;
    do {
        const double ut = u52_to_unit(mix64(zt));
        const double v = tail_log1p ? -log1p(-ut) : -log(ut);
        zt += gamma + gamma;
        const double pv = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true),
                                           __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true));
        xx = ZIG_NOR_INV_R * (role ? pv : v);
        yy = role ? v : pv;
        pairs += 1;
    } while (!(yy + yy > xx * xx) && pairs < max_pairs);
    *pairs_out = pairs;
    return xx;
}
// the plain procedure for the momentum: sixteen blocks in stream order (wave_randn_block), each drawn by the wave that owns it, the stream handed on
// through LDS; leaves the momentum in L->mom.out, returns the stream's new position
template <bool FULL>
__device__ __attribute__((noinline)) uint64_t mw_draw_momentum_sequential(const MwLdsP L, const uint64_t seed0, const uint64_t gamma, const int64_t d) {
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SeqRng r{seed0, gamma};
    for (int ww = 0; ww < MW_NWV; ++ww) {
        if (w == ww) {
#pragma unroll 1
            for (int j = 0; j < MW_EW; ++j) {
                const int jg = w * MW_EW + j;
                const int nl = FULL ? 64 : (int)max((int64_t)0, min((int64_t)64, d - 64 * (int64_t)jg));
                if (nl > 0) { const double v = wave_randn_block(r, lane, nl); if (lane < nl) L->mom.out[64 * jg + lane] = v; }
            }
            if (lane == 0) L->seed[ww] = r.seed;
        }
        __syncthreads();
        r.seed = L->seed[ww];
    }
    return r.seed;
}

// The momentum of a refresh: d sequential randn(rng) of the replica's stream (seed, gamma), left in L->mom.out in output order by the four
// waves of the workgroup together; returns the stream's new position (the same in every wave).
//   fast path = the stream compaction of pte_normals.hpp (steps 1-3 as they stand there, per wave on its own segment of 320 positions), then
//     across the waves: consumed positions per segment exchanged, every position that was not consumed scattered to its output index;
//   plain path = sixteen blocks in stream order (wave_randn_block), each drawn by the wave that owns it, the stream handed on through LDS.
template <bool FULL>
__device__ __forceinline__ uint64_t mw_draw_momentum(const MwLdsP L, const uint64_t seed0, const uint64_t gamma, const int64_t d) {
    constexpr int NWV = MW_NWV;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int dn = FULL ? MW_DMAX : (int)d;
    __syncthreads();                                                // every wave has read its part of the kept trial: the bytes are the workspace's now
    const uint64_t g64 = gamma << 6;
    const uint64_t base = seed0 + (uint64_t)(MW_SEG * w) * gamma;  // this segment: positions base + 1 .. base + MW_SEG
    const double NRM_DEAD = __longlong_as_double(0x7ff8dead00000000LL);      // a consumed stream position (no fast-path value is a NaN)
    const bool tail_log1p = (g_rng_policy & PTE_RNG_TAIL_LOG1P) != 0;
    __attribute__((address_space(3))) double *const out = L->mom.seg[w];
    __attribute__((address_space(3))) unsigned short *const evl = L->mom.ev[w];
    // 1. the segment's positions as if the fast path applied; the others go to the event list.  Per position (pte_normals.hpp): the draw's 52 bits with
    //    bit 0 cleared are 2 rabs, (2 rabs)(wi / 2) is rabs wi bit for bit, and "rabs < ki" is one unsigned compare of the pattern of 2^52 + 2 rabs
    uint64_t zc = base + (uint64_t)(lane + 1) * gamma;
    int n_ev = 0;
    uint64_t raw_n = mix64(zc);
    double tw_n = L->wi[((uint32_t)raw_n >> 1) & 0xFFu]; unsigned long long tk_n = L->ki[((uint32_t)raw_n >> 1) & 0xFFu];
#pragma unroll
    for (int j = 0; j < MW_SLOTS; ++j) {
        const uint64_t raw = raw_n;
        const double tw = tw_n; const unsigned long long tk = tk_n;
        if (j + 1 < MW_SLOTS) { zc += g64; raw_n = mix64(zc); tw_n = L->wi[((uint32_t)raw_n >> 1) & 0xFFu]; tk_n = L->ki[((uint32_t)raw_n >> 1) & 0xFFu]; }
        const uint32_t lo = (uint32_t)raw, hi = (uint32_t)(raw >> 32);
        const uint64_t mb = ((uint64_t)((hi & 0x000FFFFFu) | 0x43300000u) << 32) | (lo & ~1u);
        const double wsigned = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(0.5 * tw) ^ ((unsigned long long)(lo & 1u) << 63)));
        // (+ 0.0: rabs = 0 with the sign bit set gives -0.0 here where randn gives +0.0)
        out[64 * j + lane] = (__longlong_as_double((long long)mb) - 4503599627370496.0) * wsigned + 0.0;
        const bool slow = !(mb < ((tk << 1) | 0x4330000000000000ull));
        const uint64_t m = ballot64(slow);
        if (m) {
            const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)n_ev));
            if (slow) evl[min(at, MW_EV_CAP - 1)] = (unsigned short)(64 * j + lane);
            n_ev += __popcll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int spill = 0;                                          // uniform: positions of the NEXT segment this one's attempts consume (99: more than the bookkeeping follows)
    if (n_ev >= MW_MAX_EV) { n_ev = MW_MAX_EV; spill = 99; }
    // 2. one divergent pass over the events, lanes 2 e and 2 e + 1 resolve event e together (pte_normals.hpp, step 2)
    const int role = lane & 1;
    const bool in_pass = (lane >> 1) < n_ev;
    const bool is_ev = in_pass && role == 0;
    int e_pos = 0x3fffffff, e_kind = 0, e_delta = 0;
    double e_tail = 0.0;
    if (in_pass) {
        e_pos = evl[lane >> 1];
        const double xv = out[e_pos];
        const uint64_t zr = base + (uint64_t)(e_pos + 1 + role) * gamma;
        const uint64_t mine = mix64(zr);
        const uint64_t theirs = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mine >> 32), 0xB1, 0xF, 0xF, true) << 32)
                                | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mine, 0xB1, 0xF, 0xF, true);
        const uint64_t raw = role ? theirs : mine, nxt = role ? mine : theirs;
        const int idx = (int)((raw >> 1) & 0xFF);
        if (idx == 0) {
            int pairs = 0;
            const double xx = mw_tail_event(zr + gamma, gamma, role, tail_log1p, MW_SEG, &pairs);
            e_kind = 3; e_delta = 2 * pairs;
            e_tail = ((raw >> 9) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
        } else {
            const double u1 = u52_to_unit(nxt);
#ifdef PTE_MW_FI_GLOBAL
            const double f1 = ZIG_FI[idx - 1], f0 = ZIG_FI[idx];
#else
            const double f1 = L->fi[idx - 1], f0 = L->fi[idx];
#endif
            const double t = -0.5 * xv * xv;
            const double y = (f1 - f0) * u1 + f0;
            const double ea = (double)__builtin_amdgcn_exp2f((float)t * 1.44269504088896341f);     // decided by the single-precision exp2 outside a band of 8e-6, exactly inside it
            e_kind = (y < ea * (1.0 - 8e-6)) ? 1 : ((y > ea * (1.0 + 8e-6)) ? 2 : 0);
            if (__builtin_expect(e_kind == 0, 0)) e_kind = (y < mw_exp_exact(t)) ? 1 : 2;
            e_delta = e_kind;
        }
    }
    // 3. which events start an attempt (pte_normals.hpp, step 3), the consumed positions
    const int e_cons = (e_kind == 3) ? e_delta : 1;
    const int prev_end = __shfl_up(e_pos + e_cons, 2, 64);
    const bool covered = is_ev && lane > 1 && e_pos <= prev_end;
    bool live = is_ev;
    if (ballot64(covered) != 0ull) {
        for (int itj = 0; itj < 3; ++itj) { const bool pl = __shfl_up((int)live, 2, 64) != 0; live = is_ev && !(covered && pl); }
        const bool pl = __shfl_up((int)live, 2, 64) != 0;
        const int prev2_end = __shfl_up(e_pos + e_cons, 4, 64);
        const bool bad = (live != (is_ev && !(covered && pl))) || (is_ev && lane > 3 && e_pos <= prev2_end);
        if (ballot64(bad) != 0ull) {
            uint64_t lm = 0ull; int cend = -1;
            for (int j = 0; j < 2 * n_ev; j += 2) {
                const int pq = __builtin_amdgcn_readlane(e_pos, j);
                if (pq <= cend) continue;
                lm |= 1ull << j;
                cend = pq + __builtin_amdgcn_readlane(e_cons, j);
            }
            live = __builtin_amdgcn_inverse_ballot_w64(lm);
        }
    }
    {   // an attempt that consumes positions of the NEXT segment: one position (a wedge at this segment's last position) is followed, more is not
        const uint64_t sm = ballot64(live && e_pos + e_cons >= MW_SEG);
        if (sm != 0ull) {
            const int sl = (int)__builtin_ctzll(sm);
            const int over = __builtin_amdgcn_readlane(e_pos, sl) + __builtin_amdgcn_readlane(e_cons, sl) - (MW_SEG - 1);
            spill = (spill == 0 && over == 1 && __popcll(sm) == 1) ? 1 : 99;
        }
    }
    const int e_d = live ? e_delta : 0;
    const int e_incl = wave_iscan_i32(e_d);
    const int e_kout = e_pos - (e_incl - e_d);          // the output (counted from the segment's first) the attempt belongs to
    if (live) {
        const int first = e_pos + ((e_kind == 2) ? 0 : 1), last = min(e_pos + ((e_kind == 3) ? e_delta : 1), MW_SEG - 1);
        for (int q = first; q <= last; ++q) out[q] = NRM_DEAD;
        if (e_kind == 3) out[e_pos] = e_tail;
    }
    const int consumed = __builtin_amdgcn_readlane(e_incl, 63);
    const int first_is_event = (n_ev > 0 && evl[0] == 0) ? 1 : 0;              // (a spill INTO this segment must not land on an event)
    if (lane == 0) { L->mom_cons[w] = consumed; L->mom_fail[w] = spill | (first_is_event << 8); }
    __syncthreads();
    // 4. across the waves: outputs before this segment = positions before it minus the positions consumed before it
    int before = 0, any_fail = 0, total = 0, spill_in = 0, prev_spill = 0;
#pragma unroll
    for (int u = 0; u < NWV; ++u) {
        const int cu = L->mom_cons[u], fu = L->mom_fail[u], su = fu & 0xFF, eu = fu >> 8;
        const int o0 = MW_SEG * u - (total - prev_spill);             // index of segment u's first output (a position spilled into it is counted before it, but lies in it)
        if (o0 < dn) {                                                  // (a segment past the last output wanted may be anything)
            if (su > 1 || (prev_spill == 1 && eu)) any_fail = 1;
            if (su == 1 && u == NWV - 1) any_fail = 1;
        }
        if (u == w) { before = total; spill_in = prev_spill; }
        total += cu;
        prev_spill = (su == 1) ? 1 : 0;
    }
    if (NWV * MW_SEG - total < dn + 2) any_fail = 1;
    if (__builtin_expect(any_fail, 0)) {                            // uniform over the workgroup: everybody read the same words
        __syncthreads();                                            // (the segments are done with)
        return mw_draw_momentum_sequential<FULL>(L, seed0, gamma, d);
    }
    if (spill_in) out[0] = NRM_DEAD;                                // consumed by the last attempt of the segment before (lane-uniform store of one word)
    __builtin_amdgcn_wave_barrier();
    const int out0 = MW_SEG * w - before + spill_in;               // index of this segment's first output
    {
        double vall[MW_SLOTS];
#pragma unroll
        for (int j = 0; j < MW_SLOTS; ++j) vall[j] = out[64 * j + lane];
        int cum = out0;
#pragma unroll
        for (int j = 0; j < MW_SLOTS; ++j) {
            const double vj = vall[j];
            const uint64_t keep = ballot64(vj == vj);
            const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, (unsigned)cum));
            if (vj == vj && at < MW_DMAX) L->mom.out[at] = vj;
            cum += __popcll(keep);
        }
    }
    {   // draws consumed by the attempts of the outputs below dn (the live events are sorted by output): the stream's new position
        const uint64_t below = ballot64(live && (MW_SEG * w - before) + e_kout < dn);
        const int cb = below ? __builtin_amdgcn_readlane(e_incl, 63 - (int)__builtin_clzll(below)) : 0;
        if (lane == 0) L->mom_below[w] = cb;
    }
    __syncthreads();
    return seed0 + (uint64_t)(dn + ((L->mom_below[0] + L->mom_below[1]) + (L->mom_below[2] + L->mom_below[3]))) * gamma;
}

#ifndef PTE_MW_PACE
#define PTE_MW_PACE 1
#endif
#ifndef PTE_MW_KEEP_GK
#define PTE_MW_KEEP_GK 1
#endif
#ifndef PTE_MW_BOUNDS_TWO_WAVES
#define PTE_MW_BOUNDS_TWO_WAVES 1
#endif
#ifndef PTE_MW_SCALE_CALLED
#define PTE_MW_SCALE_CALLED 1
#endif
#ifndef PTE_MW_PACKED_SUMS
#define PTE_MW_PACKED_SUMS 0            // development builds: wave_sum_packed4 for the leapfrog's three / four sums (32 fewer instructions, a longer dependent chain): funnel(1024)
#endif                                  // 1.593 -> 1.585, toy_mvn(1024) -1 %, toy_mvn(600) +10 % (0.447 -> 0.494) -- not taken
#ifndef PTE_MW_PACE_SCANS
#define PTE_MW_PACE_SCANS 1             // the scan loop too, beside its hand-shake rule (its launch zeroes the counter, a replica's own count runs over the scans of the call): toy_mvn(600) 0.471 -> 0.436, toy_mvn(1024) 0.552 -> 0.524-0.554
#endif
#ifndef PTE_MW_PACE_T0
#define PTE_MW_PACE_T0 0               // ahead of the mean: lowest; up to two refreshes behind: 1; up to four: 2; more: 3 -- measured flat around (0, 4, 8) and (2, 6, 10);
#define PTE_MW_PACE_T1 4               // finer steps -- (0, 1, 2), (0, 2, 4) -- and coarser ones -- (0, 8, 16) -- cost 1-4 % (profiles/r06_langevin_mw.txt)
#define PTE_MW_PACE_T2 8
#endif
// PACE (the per-scan launch): four replicas share a compute unit's SIMDs, the instruction arbiter serves the oldest wave first, so they END staggered and the
// last one runs alone, at a lone wave's poor issue rate -- and the launch is as long as the slowest replica of the most loaded compute unit
// (profiles/r06_langevin_mw.txt).  Every replica counts its refreshes into e.pace; one that has begun fewer than the launch's mean asks for more issue
// slots (s_setprio), one that is ahead gives way: the replicas of a compute unit finish together, with all four waves of a SIMD busy to the end.
template <int TGT, bool FULL, bool PACE = false>
__device__ __forceinline__ void langevin_mw_body(const EngineDev &e, const AmParams &ap, const int64_t wg, const int pace_base = 0) {      // pace_base: refreshes this replica has begun since e.pace was zeroed
    constexpr int EW = MW_EW, NWV = MW_NWV;
    static_assert(NWV == 4, "the cross-wave levels of the tree are written for four waves");
    __shared__ MwLds L;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = (int)threadIdx.x; i < 256; i += 64 * NWV) {
        L.wi[i] = ZIG_WI[i]; L.ki[i] = ZIG_KI[i];
#ifndef PTE_MW_FI_GLOBAL
        L.fi[i] = ZIG_FI[i];
#endif
    }
    __syncthreads();
    const int64_t cl = am_chain_of_workgroup(e.K, wg);
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    const int64_t d = e.d;
    double *xrow = e.x + (int64_t)slot * e.ld;
    const double nhp = e.nhp[c], nprec = e.nprec[c];
    const double beta = e.beta[c], omb = 1.0 - beta;
    const double ref_nhp = -0.5 * ap.ref_prec, ref_nprec = -ap.ref_prec, log3 = ap.log3;
#ifdef PTE_MW_NO_VR                         // development builds only: what the GaussianReference branches cost in registers
    const bool v_on = false;
#else
    const bool v_on = (TGT == TGT_FUNNEL) && e.v_use != nullptr;      // a GaussianReference is active on this engine
#endif
    const bool vr = v_on && e.v_use[c] != 0;
    int par = 0, ypar = 0;                                            // uniform: which half of the exchange buffers the next exchange uses
    const MwLdsP Lp = (MwLdsP)&L;
#ifdef PTE_PROFILE_AM                      // debug builds only (tools/prof_mw.py): shader-clock time per section of the refresh loop, wave 0's view
    uint64_t mw_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mw_sync = 0;
    uint64_t mw_tlast = __builtin_amdgcn_s_memtime();
    const uint64_t mw_rt0 = __builtin_amdgcn_s_memrealtime();
#define MW_STAMP(k) do { asm volatile("" ::: "memory"); const uint64_t t_ = __builtin_amdgcn_s_memtime(); mw_prof[k] += t_ - mw_tlast; mw_tlast = t_; asm volatile("" ::: "memory"); } while (0)
#define MW_X0() asm volatile("" ::: "memory"); const uint64_t tq1_ = __builtin_amdgcn_s_memtime()
#define MW_X1() asm volatile("" ::: "memory"); mw_sync += __builtin_amdgcn_s_memtime() - tq1_
#else
#define MW_STAMP(k) do { } while (0)
#define MW_X0() do { } while (0)
#define MW_X1() do { } while (0)
#endif

    // A value that is the same in all 64 lanes, MOVED to a scalar register pair: the sums, log densities, bounds and step sizes of the algorithm are
    // wave-uniform doubles that live across the whole search -- computed by vector instructions they would each hold two VGPRs of the 128
    auto U = [](double v) -> double {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    const int cbase = 64 * EW * w + EW * lane;                         // this lane's first coordinate: four consecutive ones per lane
    auto valid = [&](int j) -> bool { return FULL || (int64_t)(cbase + j) < d; };
    auto gidx = [&](int j) -> int { return cbase + j; };
    const bool owns_first = (w == 0 && lane == 0);                    // the lane that holds coordinate 0
    // (quotients: markstein_quotient, pte_device.hpp)
    // ---- sums: the wave's four blocks (tree_sum_regs*: one subtree of the fixed tree), then the two cross-wave levels from the partial sums of all four waves
    auto exchange = [&](const auto &mine, auto &out) {                 // K partial sums -> K roots, one barrier
        constexpr int K = (int)(sizeof(mine) / sizeof(double));
        MW_X0();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) L.part[par][k][w] = mine[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double2 a = *reinterpret_cast<const double2 *>(&L.part[par][k][0]), b = *reinterpret_cast<const double2 *>(&L.part[par][k][2]);
            out[k] = U((a.x + a.y) + (b.x + b.y));
        }
        par ^= 1;
        MW_X1();
    };
    // the wave's 256-leaf node of K sums in lockstep: two in-lane levels, six across the lanes
    auto wave_nodes = [&](auto &leaf4 /* [K][EW] */, auto &nodes /* [K] */) {
        constexpr int K = (int)(sizeof(nodes) / sizeof(double));
        static_assert(EW == 4, "two in-lane levels");
        double v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = (leaf4[k][0] + leaf4[k][1]) + (leaf4[k][2] + leaf4[k][3]);
        if constexpr (K == 1) nodes[0] = wave_sum_dpp(v[0]);
        else if constexpr ((K == 3 || K == 4) && PTE_MW_PACKED_SUMS) {      // levels 3-6 of the tree paid once for all of them (pte_device.hpp: 48 instructions against 60 / 80)
            double v4[4], o4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[k] = k < K ? v[k] : 0.0;
            wave_sum_packed4<K>(v4, o4);
#pragma unroll
            for (int k = 0; k < K; ++k) nodes[k] = o4[k];
        } else {
            wave_sum_dpp_multi<K>(v);
#pragma unroll
            for (int k = 0; k < K; ++k) nodes[k] = v[k];
        }
    };
    auto reduce1 = [&](const double (&t)[EW]) -> double {
        double leaf[1][EW], mine[1], out[1];
#pragma unroll
        for (int j = 0; j < EW; ++j) leaf[0][j] = t[j];
        wave_nodes(leaf, mine);
        exchange(mine, out);
        return out[0];
    };
    auto sqr_norm = [&](const double (&v)[EW]) -> double {
        double t[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) t[j] = v[j] * v[j];
        return reduce1(t);
    };
    // the wave's partial sums of |a|^2 and |b|^2, in lockstep
    auto partial_sqr2 = [&](const double (&a)[EW], const double (&b)[EW], double &sa, double &sb) {
        double t[2][EW], o[2];
#pragma unroll
        for (int j = 0; j < EW; ++j) { t[0][j] = a[j] * a[j]; t[1][j] = b[j] * b[j]; }
        wave_nodes(t, o);
        sa = o[0]; sb = o[1];
    };
    // The funnel's scale: y = x[coordinate 0], sigma = exp(y / 2), log sigma, 1 / sigma enter every term.  They are wave-uniform, and the vector
    // unit has no scalar double arithmetic: evaluated by all four waves they are ~300 instructions of exp and log per evaluation, three quarters of
    // them redundant -- and the kernel is bound by instruction issue (four waves per SIMD).  Wave 0 (which owns coordinate 0) evaluates them alone
    // and publishes the four words; the barrier is the one the broadcast of y needed anyway.  Same functions, same argument: the same bits.
    struct FunnelScale { double y, sigma, logsigma, rinv; };
    auto funnel_scale = [&](const double (&x)[EW]) -> FunnelScale {
        MW_X0();
        if (w == 0) {
            const double y = U(x[0]);                                 // (lane 0 is the first lane: coordinate 0)
#if PTE_MW_SCALE_CALLED
            const MwScale3 sc = mw_funnel_scale_eval(y);
            const double sigma = sc.sigma, logsigma = sc.logsigma, rinv = sc.rinv;
#else
            const double sigma = exp(y / 2.0);
            const double logsigma = log(sigma);
            const double rinv = 1.0 / sigma;
#endif
            if (lane == 0) { L.ybuf[ypar][0] = y; L.ybuf[ypar][1] = sigma; L.ybuf[ypar][2] = logsigma; L.ybuf[ypar][3] = rinv; }
        }
        __syncthreads();
        FunnelScale f{U(L.ybuf[ypar][0]), U(L.ybuf[ypar][1]), U(L.ybuf[ypar][2]), U(L.ybuf[ypar][3])};
        ypar ^= 1;
        MW_X1();
        return f;
    };
    // GaussianReference end of the path (variational leg), constants read where they are used (as the sixteen-block kernel does)
    auto VM = [&](int j) -> double { return valid(j) ? e.v_mean[gidx(j)] : 0.0; };
    auto VC0 = [&](int j) -> double { return valid(j) ? e.v_c0[gidx(j)] : 0.0; };
    auto VI2 = [&](int j) -> double { return valid(j) ? e.v_i2[gidx(j)] : 0.0; };
    auto VGF = [&](int j) -> double { return valid(j) ? e.v_gf[gidx(j)] : 0.0; };
    auto variational_lp = [&](const double (&x)[EW]) -> double {          // gaussian_logdensity (GaussianReference.jl:43-49), fixed tree
        double t[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) { const double dx = x[j] - VM(j); t[j] = valid(j) ? (VC0(j) - VI2(j) * (dx * dx)) : 0.0; }
        return reduce1(t);
    };
    auto ref_lp = [&](const double (&x)[EW], double S) -> double { if (__builtin_expect(vr, 0)) return variational_lp(x); return ref_nhp * S; };
    // the guard of a vector of quotient estimates: every live lane's four exponent fields inside [123, 1923] (2^-900 ... 2^900; zero and subnormals
    // have field 0, infinities and NaN 2047): one v_bfe_u32 per estimate, three min / max pairs, two compares per lane, one ballot per vector
    auto quotients_out_of_range = [&](const double (&q)[EW]) -> bool {
        unsigned lo = 0x7FFu, hi = 0u;
#pragma unroll
        for (int j = 0; j < EW; ++j) {
            const unsigned ex = valid(j) ? quotient_exponent_field(q[j]) : 1023u;
            lo = min(lo, ex); hi = max(hi, ex);
        }
        return ballot64(lo < 123u || hi > 1923u) != 0ull;
    };
    // a / sigma for the wave's coordinates (sigma uniform; rinv = 1 / sigma); `o` must not be `a`
    auto div_sigma = [&](const double (&a)[EW], double sigma, double rinv, bool divisor_ok, double (&o)[EW]) {
        double qe[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) o[j] = markstein_quotient(a[j], sigma, rinv, qe[j]);
        if (__builtin_expect(!divisor_ok || quotients_out_of_range(qe), 0)) {
#pragma unroll
            for (int j = 0; j < EW; ++j) o[j] = mw_div_exact(a[j], sigma);
        }
    };
    // the funnel's gradient without the first coordinate's sum (AmTarget::funnel: g2 = -(z / sigma) / sigma) blended with the reference's:
    // g[j] = (reference gradient) (1 - beta) + g2[j] beta; the owner of coordinate 0 overwrites its entry once the sum is known
    auto funnel_gradient_elementwise = [&](const double (&x)[EW], const double (&zi)[EW], double sigma, double rinv, bool sok, double (&g)[EW]) {
        double zs[EW];
        div_sigma(zi, sigma, rinv, sok, zs);
        if (__builtin_expect(vr, 0)) {               // BufferedAD{GaussianReference}: -1/s^2 (x - m)
#pragma unroll
            for (int j = 0; j < EW; ++j) g[j] = (VGF(j) * (x[j] - VM(j))) * omb + (valid(j) ? -zs[j] : 0.0) * beta;
        } else {
#pragma unroll
            for (int j = 0; j < EW; ++j) g[j] = (ref_nprec * x[j]) * omb + (valid(j) ? -zs[j] : 0.0) * beta;
        }
    };
    // funnel log density alone (test/supporting/dimensional-analysis.jl:36-48): AmTarget::funnel, term for term
    auto funnel_lp = [&](const double (&x)[EW]) -> double {
        const FunnelScale fs = funnel_scale(x);
        const double y = fs.y, sigma = fs.sigma, logsigma = fs.logsigma, rinv = fs.rinv;
        const bool sok = markstein_divisor_ok(sigma);
        const double LOG2PI = 1.8378770664093453;
        double t[EW], zi[EW];
        div_sigma(x, sigma, rinv, sok, zi);
#pragma unroll
        for (int j = 0; j < EW; ++j) t[j] = valid(j) ? (-(zi[j] * zi[j] + LOG2PI) / 2.0 - logsigma) : 0.0;
        const double zv = y / 3.0;
        if (owns_first) t[0] = -(zv * zv + LOG2PI) / 2.0 - log3;
        return reduce1(t);
    };
    // LogDensityProblems.logdensity_and_gradient of the interpolated funnel path with Q = sum q^2 taken alongside (AmTarget::logdensity_and_gradient_q<true>):
    // four independent fixed trees, their in-wave parts taken two at a time (registers), ONE exchange for all four
    auto funnel_logdensity_and_gradient_q = [&](const double (&x)[EW], double (&g)[EW], const double (&q)[EW], double &Q) -> double {
        const FunnelScale fs = funnel_scale(x);
        const double y = fs.y, sigma = fs.sigma, logsigma = fs.logsigma, rinv = fs.rinv;
        const bool sok = markstein_divisor_ok(sigma);
        const double LOG2PI = 1.8378770664093453;
        double mine[4], out[4];
#if PTE_MW_PACKED_SUMS
        {
            // |x|^2, the log-density terms, the first coordinate's gradient terms, |q|^2: their two in-lane levels one after the other (four doubles
            // stay), then ONE packed butterfly (levels 3-6 of the tree paid once for the four sums)
            double v4[4];
            v4[0] = (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
            v4[3] = (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]);
            double zi[EW];
            div_sigma(x, sigma, rinv, sok, zi);
            {
                double t[2][EW];
#pragma unroll
                for (int j = 0; j < EW; ++j) {
                    t[0][j] = valid(j) ? (-(zi[j] * zi[j] + LOG2PI) / 2.0 - logsigma) : 0.0;
                    t[1][j] = valid(j) ? (zi[j] * zi[j] - 1.0) / 2.0 : 0.0;
                }
                const double zv = y / 3.0;
                if (owns_first) { t[0][0] = -(zv * zv + LOG2PI) / 2.0 - log3; t[1][0] = -(y / 9.0); }
                v4[1] = (t[0][0] + t[0][1]) + (t[0][2] + t[0][3]);
                v4[2] = (t[1][0] + t[1][1]) + (t[1][2] + t[1][3]);
            }
            wave_sum_packed4<4>(v4, mine);
            funnel_gradient_elementwise(x, zi, sigma, rinv, sok, g);
        }
#else
        partial_sqr2(x, q, mine[0], mine[3]);
        {
            double zi[EW];
            div_sigma(x, sigma, rinv, sok, zi);
            {
                double t[2][EW], o[2];
#pragma unroll
                for (int j = 0; j < EW; ++j) {
                    t[0][j] = valid(j) ? (-(zi[j] * zi[j] + LOG2PI) / 2.0 - logsigma) : 0.0;
                    t[1][j] = valid(j) ? (zi[j] * zi[j] - 1.0) / 2.0 : 0.0;
                }
                const double zv = y / 3.0;
                if (owns_first) { t[0][0] = -(zv * zv + LOG2PI) / 2.0 - log3; t[1][0] = -(y / 9.0); }
                wave_nodes(t, o);
                mine[1] = o[0]; mine[2] = o[1];
            }
            funnel_gradient_elementwise(x, zi, sigma, rinv, sok, g);
        }
#endif
        exchange(mine, out);
        const double S = out[0], l2 = out[1];
        Q = out[3];
        if (owns_first) {                            // g2[0] = the sum; the same blend as the other coordinates
            if (__builtin_expect(vr, 0)) g[0] = (VGF(0) * (x[0] - VM(0))) * omb + out[2] * beta;
            else g[0] = (ref_nprec * x[0]) * omb + out[2] * beta;
        }
        const double l1 = ref_lp(x, S);
        double logdens = 0.0;
        logdens += l1 * omb;
        logdens += l2 * beta;
        return logdens;
    };

    double x[EW];
    if (is_ref_chain(e, c)) {
        if (e.compose_phase == 2) return;
        // sample_iid! at the reference (pigeons.jl:104-105): d sequential normals, drawn and stored by wave 0 as the one-wave kernel does
        double lp0r = 0.0, S0 = 0.0;
        if (w == 0) {
            lp0r = lp_before_explore(e, c, slot);
            if (vr) {      // sample_iid!(::GaussianReference) (GaussianReference.jl:33-40): x_i = randn * sd_i + mean_i, in draw order
                SeqRng r0{e.rng[2 * slot], e.rng[2 * slot + 1]};
                for (int jg = 0; jg < NWV * EW; ++jg) {
                    const int nl = (int)max((int64_t)0, min((int64_t)64, d - 64 * (int64_t)jg));
                    if (nl > 0) {
                        const double z = wave_randn_block(r0, lane, nl);
                        if (lane < nl) xrow[64 * jg + lane] = z * e.v_std[64 * jg + lane] + e.v_mean[64 * jg + lane];
                    }
                }
                if (lane == 0) e.rng[2 * slot] = r0.seed;
            } else {
                S0 = iid_refresh<4>(e, slot, e.sd[c], lane);
            }
        }
        __syncthreads();                                              // wave 0's row is visible to the workgroup (one compute unit: one L1)
        if (TGT == TGT_FUNNEL || vr) {
#pragma unroll
            for (int j = 0; j < EW; ++j) x[j] = valid(j) ? xrow[gidx(j)] : 0.0;
        }
        if (vr) { S0 = sqr_norm(x); if (owns_first) e.suff[slot] = S0; }
        double l20 = 0.0, l30 = 0.0;
        if (TGT == TGT_FUNNEL) {
            l20 = funnel_lp(x);
            if (owns_first) e.suff2[slot] = l20;
            if (v_on) { l30 = variational_lp(x); if (owns_first) e.suff3[slot] = l30; }
        }
        if (w == 0) record_after_explore_impl(e, cl, c, slot, lane, lp0r, S0, l20, l30);
        return;
    }
    const double lp_before = (w == 0) ? lp_before_explore(e, c, slot) : 0.0;
#pragma unroll
    for (int j = 0; j < EW; ++j) x[j] = valid(j) ? xrow[gidx(j)] : 0.0;

    SeqRng r{e.rng[2 * slot], e.rng[2 * slot + 1]};                   // every wave holds the stream; they advance it identically
    // build_preconditioner! (Preconditioner.jl:57-77)
    double M[EW];
#pragma unroll
    for (int j = 0; j < EW; ++j) M[j] = 1.0;
    if (ap.target_std != nullptr && ap.precond != 0) {
        double sdv[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) sdv[j] = valid(j) ? ap.target_std[gidx(j)] : 1.0;
        if (ap.precond == 1) {
#pragma unroll
            for (int j = 0; j < EW; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : 1.0 / sdv[j];
        } else {
            const double u = r.rand();
            if (u <= ap.p0) {
#pragma unroll
                for (int j = 0; j < EW; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : 1.0 / sdv[j];
            } else if (u <= ap.p0 + ap.p1) {
                // ones
            } else {
                const double mix = r.rand(), rmix = 1.0 - mix;
#pragma unroll
                for (int j = 0; j < EW; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : mix + rmix / sdv[j];
            }
        }
    }
    double Minv[EW];
    bool m_not_one = false, m_excluded = false;
#pragma unroll
    for (int j = 0; j < EW; ++j) {
        Minv[j] = 1.0 / M[j];
        m_not_one = m_not_one || (valid(j) && M[j] != 1.0);
        m_excluded = m_excluded || (valid(j) && !markstein_divisor_ok(M[j]));
    }
    const bool m_one = ballot64(m_not_one) == 0ull;                    // the identity (round 1; a third of the scans of MixDiagonalPreconditioner): a / 1.0 is a
    const bool m_ok = ballot64(m_excluded) == 0ull;
    // a[j] / M[j] for the wave's coordinates; `o` must not be `a` (the rare branch reads `a` again)
    auto div_M = [&](const double (&a)[EW], double (&o)[EW]) {
        if (m_one) {
#pragma unroll
            for (int j = 0; j < EW; ++j) o[j] = a[j];
            return;
        }
        double qe[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) o[j] = markstein_quotient(a[j], M[j], Minv[j], qe[j]);
        if (__builtin_expect(!m_ok || quotients_out_of_range(qe), 0)) {
#pragma unroll
            for (int j = 0; j < EW; ++j) o[j] = mw_div_exact(a[j], M[j]);
        }
    };

    double p[EW], g[EW], pb[EW], g0[EW];
    long long steps_sum = 0; int steps_n = 0;
    double fac_sum = 0.0; int fac_n = 0;
    int rev_sum = 0, rev_n = 0;
    double acc_sum = 0.0; int acc_n = 0;
    int err = 0;
    // (as in the one-wave kernel: each point's log density / conditioned gradient is evaluated once and its bits reused where the
    // reference recomputes them -- pure functions of the state)
    double lp0 = 0.0, pp0 = 0.0;
    // log density and CONDITIONED gradient at x, with Q = sum q^2
    auto density_and_conditioned_gradient = [&](double (&gout)[EW], const double (&q)[EW], double &Q) -> double {
        double lp, gr[EW];
        if constexpr (TGT == TGT_MVN) {
            double mine[2], out[2];
            partial_sqr2(x, q, mine[0], mine[1]);
#pragma unroll
            for (int j = 0; j < EW; ++j) gr[j] = nprec * x[j];
            exchange(mine, out);
            Q = out[1];
            lp = nhp * out[0];
        } else lp = funnel_logdensity_and_gradient_q(x, gr, q, Q);
        div_M(gr, gout);
        return lp;
    };
    // the conditioned gradient at x alone, given the first coordinate's entry (kept from the evaluation that made the sum): elementwise
    auto conditioned_gradient_again = [&](double (&gout)[EW], double first_entry) {
        double gr[EW];
        if constexpr (TGT == TGT_MVN) {
#pragma unroll
            for (int j = 0; j < EW; ++j) gr[j] = nprec * x[j];
        } else {
            const FunnelScale fs = funnel_scale(x);
            const double sigma = fs.sigma, rinv = fs.rinv;
            const bool sok = markstein_divisor_ok(sigma);
            double zi[EW];
            div_sigma(x, sigma, rinv, sok, zi);
            funnel_gradient_elementwise(x, zi, sigma, rinv, sok, gr);
        }
        div_M(gr, gout);
        if (TGT != TGT_MVN && owns_first) gout[0] = first_entry;
    };
    auto kinetic = [&]() -> double { return U(0.5 * sqr_norm(p)); };
    // hamiltonian_dynamics! with n_steps = 1 FROM the point (state = LDS row xsrc, momentum = psrc, conditioned gradient = g0): every trial of a search
    // starts from the same point, so the "restore" of the reference is reading it again -- no copies back into x and p after a trial
    auto leap_frog = [&](double eps, int xsrc, const double (&psrc)[EW], double &logp_out, double &ke_out) -> bool {
        const double half = eps / 2;
#pragma unroll
        for (int j = 0; j < EW; ++j) p[j] = psrc[j] + half * g0[j];
        {
            double pm[EW];
            div_M(p, pm);
#pragma unroll
            for (int j = 0; j < EW; ++j) x[j] = L.v[xsrc][gidx(j)] + eps * pm[j];
        }
        if constexpr (TGT == TGT_MVN) {
            // the gradient is elementwise here (-prec x), so the momentum after the second half kick is known before any sum is: |x|^2, |p|^2
            // after the first kick and |p|^2 after the second are three independent fixed trees -- ONE exchange per leapfrog instead of two,
            // the same bits.  The second kick is made in place; where the reference returns BEFORE it (a non-finite joint: rare) the momentum
            // after the first kick is evaluated again -- the same expression, the same bits.
            double mine[3], out[3], gr[EW];
#pragma unroll
            for (int j = 0; j < EW; ++j) gr[j] = nprec * x[j];
            div_M(gr, g);
            {
                double t[3][EW];
#pragma unroll
                for (int j = 0; j < EW; ++j) {
                    t[0][j] = x[j] * x[j]; t[1][j] = p[j] * p[j];
                    p[j] = p[j] + half * g[j];
                    t[2][j] = p[j] * p[j];
                }
                wave_nodes(t, mine);
            }
            exchange(mine, out);
            const double logp = U(nhp * out[0]);
            logp_out = logp;
            const double ke_mid = U(0.5 * out[1]);
            ke_out = ke_mid;
            const double cur = logp - ke_mid;
            if (__builtin_expect(!isfinite(cur), 0)) {
#pragma unroll
                for (int j = 0; j < EW; ++j) p[j] = psrc[j] + half * g0[j];
                return false;
            }
            const double sq = out[2];
            ke_out = U(0.5 * sq);
            if (__builtin_expect(!isfinite(sq), 0)) return false;
            return true;
        } else {
            double pp_mid;
            const double logp = U(density_and_conditioned_gradient(g, p, pp_mid));
            logp_out = logp;
            const double ke_mid = U(0.5 * pp_mid);
            ke_out = ke_mid;
            const double cur = logp - ke_mid;
            if (__builtin_expect(!isfinite(cur), 0)) return false;
#pragma unroll
            for (int j = 0; j < EW; ++j) p[j] = p[j] + half * g[j];
            const double sq = sqr_norm(p);
            ke_out = U(0.5 * sq);
            if (__builtin_expect(!isfinite(sq), 0)) return false;
            return true;
        }
    };
    double *gk_row = (TGT == TGT_FUNNEL && PTE_MW_KEEP_GK) ? e.mw_gk + ((int64_t)cl * NWV + w) * (64 * EW) : nullptr;      // this wave's 256 words of the kept trial's conditioned gradient
    double lpk = 0.0, kek = 0.0, gk_first = 0.0; bool okk = true;   // the kept trial's scalars (gk_first: its conditioned gradient's first entry, per lane); its vectors: L.v[MW_XK], L.v[MW_PK]
    // auto_step_size (AutoMALA.jl:184-214) as ONE loop with one trial leapfrog in it (mode 0: the first trial at the current step size; 1: halving
    // until the joint's change rises above `lower`; 2: doubling until it falls below `upper` or stops being finite) -- the sequence of trials,
    // of kept trials and of restores is the reference's; one copy of the leapfrog in the kernel instead of six.  `forward`: the search from the
    // refresh's start point (it keeps the trial the proposal would repeat; every trial starts from the start state in LDS); otherwise from the
    // proposed point (the kept trial's state in LDS).  The momentum every trial starts from is pb.  x and p are NOT restored behind the last trial:
    // the callers load what they go on with.
    auto auto_step_size = [&](double &lower, double &upper, double h_before, bool forward) -> int {
        const int xsrc = forward ? MW_XS : MW_XK;
#pragma unroll
        for (int j = 0; j < EW; ++j) pb[j] = p[j];
        double eps = ap.step_size;
        int mode = 0, n = 0, n_steps = 0, exponent = 0;
        for (;;) {
            double t_lp = 0.0, t_ke = 0.0;
            const bool t_ok = leap_frog(eps, xsrc, pb, t_lp, t_ke);
            if (forward && mode == 0) { lower = U(L.bounds[0]); upper = U(L.bounds[1]); }      // (behind the leapfrog's exchange barrier: wave 0 has published them)
            const double diff = U((t_lp - t_ke) - h_before);
            const bool stop_growing = mode == 2 && (!isfinite(diff) || diff < upper);
            if (forward && !stop_growing) {     // the trial the proposal would repeat: the last one when shrinking or not moving, the last but one when growing
#pragma unroll
                for (int j = 0; j < EW; ++j) { L.v[MW_XK][gidx(j)] = x[j]; L.v[MW_PK][gidx(j)] = p[j]; }
                gk_first = g[0];
                if constexpr (TGT == TGT_FUNNEL && PTE_MW_KEEP_GK) {      // ... and its conditioned gradient (8 KB per replica: no room beside the four LDS rows; it stays in the L2)
                    double2 *gk = reinterpret_cast<double2 *>(gk_row + 4 * lane);
                    gk[0] = make_double2(g[0], g[1]); gk[1] = make_double2(g[2], g[3]);
                }
                lpk = t_lp; kek = t_ke; okk = t_ok;
            }
            if (mode == 0) {
                if (!isfinite(diff) || diff < lower) mode = 1;
                else if (diff > upper) mode = 2;
                else break;
            } else if (mode == 1) {
                if (eps == 0.0) { err = ERR_AM_STEP; break; }
                if (diff > lower) { n_steps = n; exponent = -n; break; }
            } else if (stop_growing) { n_steps = n; exponent = n - 1; break; }
            n += 1;
            eps = U((mode == 1) ? eps / 2.0 : eps * 2.0);
        }
        steps_sum += 1 + n_steps; steps_n += 1;
        if (e.am_log != nullptr && owns_first && fac_n < e.am_log_cap) e.am_log[(e.trace_idx * e.K + cl) * e.am_log_cap + fac_n] = (int16_t)exponent;
        fac_sum = U(fac_sum + ldexp(1.0, exponent)); fac_n += 1;
        return exponent;
    };

    for (int it = 0; it < ap.n_refresh && !err; ++it) {
        MW_STAMP(7);
        unsigned int begun = 0;
        const bool paced = PACE && PTE_MW_PACE && ap.pace && !ap.mala;             // (MALA: a refresh is one leapfrog -- equal work, and 37 k increments of one word per 0.3 ms cost more than they steer: 0.335 -> 0.49 ms)
        if (paced && owns_first) begun = __hip_atomic_fetch_add(e.pace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (consumed behind the momentum)
        // the refresh's start state (read back by the forward search's restores and on a rejection)
#pragma unroll
        for (int j = 0; j < EW; ++j) L.v[MW_XS][gidx(j)] = x[j];
        // the momentum: d sequential randn(rng), left in LDS in output order by the four waves together (mw_draw_momentum)
        r.seed = mw_draw_momentum<FULL>(Lp, r.seed, r.gamma, d);
#pragma unroll
        for (int j = 0; j < EW; ++j) p[j] = valid(j) ? L.mom.out[gidx(j)] : 0.0;
        if (paced) {
            if (owns_first) {                                           // refreshes begun by the K workgroups before this one: mean = begun / K against this replica's own count
                const long long ahead = (long long)begun - (long long)(pace_base + it) * (long long)e.K;
                const long long hk = (long long)e.K / 2;                  // thresholds in half refreshes of the mean
                L.prio = ahead < PTE_MW_PACE_T0 * hk ? 0 : ahead < PTE_MW_PACE_T1 * hk ? 1 : ahead < PTE_MW_PACE_T2 * hk ? 2 : 3;
            }
        }
        __syncthreads();                                                // (the workspace's bytes are the kept trial's and the start gradient's again from here on)
        if (paced) {
            switch (L.prio) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
        }
        MW_STAMP(0);
        if (it == 0) lp0 = U(density_and_conditioned_gradient(g0, p, pp0));
        else pp0 = sqr_norm(p);
        const double lp_s = lp0;
#pragma unroll
        for (int j = 0; j < EW; ++j) L.v[MW_GS][gidx(j)] = g0[j];
        MW_STAMP(1);
        const double init_joint = U(lp0 - 0.5 * pp0);
        if (!isfinite(init_joint)) { err = ERR_AM_DENSITY; break; }
        if (ap.mala) {                                   // mala! (MALA.jl:79-96)
            double lpn, ken;
#pragma unroll
            for (int j = 0; j < EW; ++j) pb[j] = p[j];
            leap_frog(ap.step_size, MW_XS, pb, lpn, ken);
#pragma unroll
            for (int j = 0; j < EW; ++j) p[j] = p[j] * -1.0;
            const double ex = MW_EXP((lpn - ken) - init_joint);
            const double probability = ex < 1.0 ? ex : (isnan(ex) ? ex : 1.0);
            acc_sum = U(acc_sum + probability); acc_n += 1;
            if (!(r.rand() < probability)) {
#pragma unroll
                for (int j = 0; j < EW; ++j) x[j] = L.v[MW_XS][gidx(j)];                 // (lp0, g0 stay those of the start point)
            } else {
                lp0 = lpn;
#pragma unroll
                for (int j = 0; j < EW; ++j) g0[j] = g[j];
            }
            steps_sum += 1; steps_n += 1;
            continue;
        }
        // the bounds of the search: log of two uniforms (every wave draws them: the streams stay identical); the two logarithms -- ~250 instructions of
        // wave-uniform arithmetic -- by wave 0 alone, published before the first trial leapfrog's exchange and read by everybody behind its barrier
        const double ua = r.rand(), ub = r.rand();
#if PTE_MW_BOUNDS_TWO_WAVES                          // ... one logarithm each by waves 0 and 1, side by side (the other two go ahead to the barrier)
        if (w < 2) {
            const double b_ = MW_LOG((w == 0) == (ua < ub) ? ua : ub);      // wave 0: the smaller uniform's, wave 1: the larger's
            if (lane == 0) L.bounds[w] = b_;
        }
#else
        if (w == 0) {
            const double lo_ = MW_LOG(ua < ub ? ua : ub), hi_ = MW_LOG(ua < ub ? ub : ua);
            if (lane == 0) { L.bounds[0] = lo_; L.bounds[1] = hi_; }
        }
#endif
        double lower = 0.0, upper = 0.0;                 // (set by the forward search behind its first exchange)
        MW_STAMP(2);
        // forward search from the start point, then (scan != 1) the reversed search from the proposed point: one copy of the search
        int proposed = 0, reversed = 0;
        double h_rev = 0.0;
        for (int dir = 0; dir < 2 && !err; ++dir) {
            if (dir == 1) {
                MW_STAMP(3);
                // leap_frog!(..., step_size * 2^proposed) from the start point == the trial the search kept: its state and momentum from LDS, its
                // conditioned gradient evaluated again (elementwise; the same bits)
#pragma unroll
                for (int j = 0; j < EW; ++j) { x[j] = L.v[MW_XK][gidx(j)]; p[j] = L.v[MW_PK][gidx(j)]; }
                lp0 = lpk;                                   // log density at the proposed point: the leapfrog computed it
                if constexpr (TGT == TGT_FUNNEL && PTE_MW_KEEP_GK) {      // the funnel's evaluation is ~450 instructions and a barrier: read the bits back instead (0.8 of a trial leapfrog per refresh)
                    const double2 *gk = reinterpret_cast<const double2 *>(gk_row + 4 * lane);
                    const double2 a = gk[0], b = gk[1];
                    g0[0] = a.x; g0[1] = a.y; g0[2] = b.x; g0[3] = b.y;
                } else conditioned_gradient_again(g0, gk_first);
                MW_STAMP(4);
                if (!ap.use_mh) break;                       // no MH step: the chain stays where the proposal leapfrog ended
#pragma unroll
                for (int j = 0; j < EW; ++j) p[j] = p[j] * -1.0;
                h_rev = U(lp0 - (okk ? kek : kinetic()));
            }
            const int ex = auto_step_size(lower, upper, dir == 0 ? init_joint : h_rev, dir == 0);
            if (dir == 0) proposed = ex; else reversed = ex;
        }
        if (err) break;
        if (ap.use_mh) {
            MW_STAMP(5);
            const bool passed = reversed == proposed;
            rev_sum += passed ? 1 : 0; rev_n += 1;
            double probability = 0.0;
            if (passed) {
                const double ex = MW_EXP(h_rev - init_joint);   // final_joint_log == log_joint at the proposed point
                probability = ex < 1.0 ? ex : (isnan(ex) ? ex : 1.0);
            }
            acc_sum = U(acc_sum + probability); acc_n += 1;
            if (!(r.rand() < probability)) {
                lp0 = lp_s;
#pragma unroll
                for (int j = 0; j < EW; ++j) { x[j] = L.v[MW_XS][gidx(j)]; g0[j] = L.v[MW_GS][gidx(j)]; }
            } else {
#pragma unroll
                for (int j = 0; j < EW; ++j) x[j] = L.v[MW_XK][gidx(j)];              // the proposed point (lp0, g0 are its already)
            }
            MW_STAMP(6);
        }
    }
#ifdef PTE_PROFILE_AM
    if (owns_first) {
        double *o = e.on_m2 + 2 * (d + 1) + 12 * cl;
        for (int k = 0; k < 7; ++k) o[k] = (double)mw_prof[k];
        o[0] += (double)mw_prof[7];            // (loop head + the start state's store: with the momentum)
        o[7] = (double)(mw_rt0 & 0xFFFFFFFFFFFull); o[10] = (double)mw_sync;       // (o[7]: the workgroup's start on the 100 MHz clock)
        o[8] = (double)(__builtin_amdgcn_s_memrealtime() - mw_rt0); o[9] = (double)steps_sum;
        const uint32_t hw_ = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_ = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // HW_ID (cu_id 11:8, sh_id 12, se_id 15:13), XCC_ID
        o[11] = (double)ap.n_refresh + 4096.0 * (double)(((xcc_ & 15u) << 8) | ((hw_ >> 8) & 0xFFu));                          // (which compute unit ran this replica)
    }
#endif
    if (err) { if (owns_first) set_error(e, err, (int)c, -1); return; }
#pragma unroll
    for (int j = 0; j < EW; ++j) if (valid(j)) xrow[gidx(j)] = x[j];
    const double S = sqr_norm(x);
    double l2 = 0.0, l3 = 0.0;
    if (TGT == TGT_FUNNEL) l2 = funnel_lp(x);
    if (v_on) l3 = variational_lp(x);
    if (owns_first) {
        e.suff[slot] = S;
        if (TGT == TGT_FUNNEL) e.suff2[slot] = l2;
        if (v_on) e.suff3[slot] = l3;
        e.rng[2 * slot] = r.seed;
        e.expl_steps_sum[cl] += (double)steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += acc_sum;             e.expl_acc_n[cl] += acc_n;
        e.am_fac_sum[cl] += fac_sum;               e.am_fac_n[cl] += fac_n;
        e.am_rev_sum[cl] += (double)rev_sum;       e.am_rev_n[cl] += rev_n;
    }
    __syncthreads();                                                  // every wave's part of the row is stored before wave 0's recorders read it
    if (w == 0) record_after_explore(e, cl, c, slot, lane, lp_before, S, l2, l3);
}

// Waves per SIMD the register allocation is held to: 4 = 128 VGPRs, four workgroups per compute unit, 1024 replicas resident.  Scaled-precision MVN
// path: the trial loop is free of scratch traffic (54-68 values are spilled around it, at refresh level).  Funnel path: its evaluation (two quotient
// passes, four sums, the blended gradient) does not fit beside the six vectors: 128-182 values spilled, a few reloaded per trial -- and still the
// fastest setting once the trial loop stopped copying registers (funnel(1024), N = 1024: 1.86 ms per scan at 4, 2.05 at 3, 1.98 at 2 waves per
// SIMD = two generations of 512 replicas; before that change 4.1 / 3.1 / 2.2: profiles/r06_langevin_mw.txt).
#ifndef PTE_MW_OCC
#define PTE_MW_OCC 4
#endif
#ifndef PTE_MW_OCC_FUNNEL
#define PTE_MW_OCC_FUNNEL 4
#endif
// The body as a CALLED function in the scan loop (as automala_body_called: inlined, the loop's long-lived values push the spilled values from 54-182
// to 147-365).  What it must NOT get is a reference to the caller's copies of the engine and the parameters: every e.* / ap.* would be a load from the
// caller's stack, to be repeated after any store that might alias it -- 4x the vector memory reads, 0.65 -> 0.97 ms per scan measured.  It gets the
// KERNEL-ARGUMENT segment instead (constant address space: scalar loads, invariant, re-loadable instead of spilled -- what the per-scan kernel sees)
// plus the two words the scan loop changes per scan.
// Nor may they arrive as ordinary arguments: a called function's arguments live in VECTOR registers and count as divergent -- every value derived from them
// would be a vector value, every branch on them a divergent one.  The function reads the kernel-argument segment pointer and the workgroup id itself (implicit
// scalar inputs of any function of the call graph) and takes the two per-scan words through v_readfirstlane.
// (llvm.amdgcn.kernarg.segment.ptr is null outside a kernel; the IMPLICIT-argument pointer is an input of every function, and the implicit arguments
// follow the explicit ones: the segment starts sizeof(explicit arguments), rounded up to 8, before it.)
typedef const char __attribute__((address_space(4))) *MwKernargP;
constexpr size_t mw_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t MW_KERNARG_AP = mw_align_up(sizeof(EngineDev), alignof(AmParams));                                         // k_*_langevin_mw(EngineDev, AmParams[, ScanLoop])
constexpr size_t MW_KERNARG_END_EXPLORE = mw_align_up(MW_KERNARG_AP + sizeof(AmParams), 8);
constexpr size_t MW_KERNARG_END_SCANS = mw_align_up(mw_align_up(MW_KERNARG_AP + sizeof(AmParams), alignof(ScanLoop)) + sizeof(ScanLoop), 8);
__device__ __forceinline__ int mw_kernarg_check(const EngineDev &e, const AmParams &ap) { return (int)e.K * 31 + (int)e.d * 7 + ap.n_refresh; }
template <int TGT, bool FULL, bool SCANS>
__device__ __attribute__((noinline))        // (amdgpu_waves_per_eu is a kernel attribute; the AMDGPU attributor hands the caller's bound down to this function)
void langevin_mw_body_called(const int trace_idx_lo, const int trace_idx_hi, const int use_mh, const int pace_base, const int check) {
    const MwKernargP ka = (MwKernargP)__builtin_amdgcn_implicitarg_ptr() - (SCANS ? MW_KERNARG_END_SCANS : MW_KERNARG_END_EXPLORE);
    EngineDev e = *(const EngineDev *)ka;                               // (cast to generic for the copy constructor's sake: InferAddressSpaces takes the loads back to address space 4)
    AmParams ap = *(const AmParams *)(ka + MW_KERNARG_AP);
    // the layout this relies on (implicit arguments right behind the explicit ones) is the code-object ABI's, not the language's: should a toolchain ever
    // move them, the words read here are not the caller's -- stop the launch (hipErrorLaunchFailure) instead of running on them
    if (mw_kernarg_check(e, ap) != __builtin_amdgcn_readfirstlane(check)) __builtin_trap();
    e.trace_idx = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(trace_idx_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(trace_idx_lo));
    ap.use_mh = __builtin_amdgcn_readfirstlane(use_mh);
    langevin_mw_body<TGT, FULL, !SCANS || PTE_MW_PACE_SCANS>(e, ap, blockIdx.x, __builtin_amdgcn_readfirstlane(pace_base));
}

template <int TGT, bool FULL>
__global__ __launch_bounds__(64 * MW_NWV)
__attribute__((amdgpu_waves_per_eu(TGT == TGT_FUNNEL ? PTE_MW_OCC_FUNNEL : PTE_MW_OCC, TGT == TGT_FUNNEL ? PTE_MW_OCC_FUNNEL : PTE_MW_OCC)))
void k_explore_langevin_mw(EngineDev e, AmParams ap) {
#ifdef PTE_MW_EXPLORE_CALLED               // development builds only: what the call costs the per-scan kernel
    langevin_mw_body_called<TGT, FULL, false>((int)(uint32_t)e.trace_idx, (int)(uint32_t)((uint64_t)e.trace_idx >> 32), ap.use_mh, 0, mw_kernarg_check(e, ap));
#else
    langevin_mw_body<TGT, FULL, true>(e, ap, blockIdx.x);
#endif
}

#ifndef PTE_MW_PRIO
#define PTE_MW_PRIO 1
#endif
// One launch per pte_run_scans (pte_kernels.hpp "ScanLoop"), as k_scans_automala: the refreshes of a scan, then the pairwise swap hand-shake
// (thread 0; the other 255 wait at the workgroup barrier and take no issue slots), for all the scans of the call.  Why here, where the loop was
// held to d <= 512 through round 5: with four workgroups per compute unit and an age-ordered instruction arbiter the per-scan launch is as long
// as the most loaded compute unit's LAST replica (profiles/r06_langevin_mw.txt: ends p10 / p50 / max 1180 / 1430 / 1950 us on the funnel,
// correlation of a replica's duration with its own work 0.73, 0.28 on the MVN path; the sum of a compute unit's four replicas varies by +- 30 %
// from scan to scan, at random) -- without a launch boundary the next scan's work fills those tails.
template <int TGT, bool FULL>
__global__ __launch_bounds__(64 * MW_NWV)
__attribute__((amdgpu_waves_per_eu(TGT == TGT_FUNNEL ? PTE_MW_OCC_FUNNEL : PTE_MW_OCC, TGT == TGT_FUNNEL ? PTE_MW_OCC_FUNNEL : PTE_MW_OCC)))
void k_scans_langevin_mw(EngineDev e, AmParams ap, ScanLoop sl) {
    __shared__ int wg_word, wg_prio;
    const int64_t cl = am_chain_of_workgroup(e.K, blockIdx.x);
    if (threadIdx.x == 0) { wg_word = scan_loop_gate(sl) ? 1 : 0; wg_prio = 1; }       // every workgroup of the launch is resident, or nobody starts (pte_kernels.hpp)
    __syncthreads();
    if (!wg_word) return;
    for (int64_t i = 0; i < sl.n_scans; ++i) {
        e.trace_idx = sl.scan_idx0 + i;
        if (!ap.mala) ap.use_mh = (sl.first_scan + i != 1) ? 1 : 0;
        langevin_mw_body_called<TGT, FULL, true>((int)(uint32_t)e.trace_idx, (int)(uint32_t)((uint64_t)e.trace_idx >> 32), ap.use_mh, (int)i * ap.n_refresh, mw_kernarg_check(e, ap));
        __syncthreads();                                               // every thread's stores of the explore step happen before thread 0's release
        if (threadIdx.x == 0) {
#if PTE_MW_PRIO
            // Who waits for whom?  A chain whose partner has already published this scan's epoch is on the ladder's critical path: its waves ask the
            // instruction arbiter for more (s_setprio); one that is there first gives way.  (Four waves share a SIMD and keep it ~96 % busy.)
            const int64_t c = e.c0 + cl, pc = deo_partner(e.N, ((sl.first_scan + i) % 2 == 0) ? 1 : 0, c);
            if (pc != c) {
                const bool late = __hip_atomic_load(&sl.flag[pc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= sl.epoch0 + 1ull + (unsigned long long)i;
                int p = wg_prio;
                p = late ? (p < 3 ? p + 1 : 3) : (p > 0 ? p - 1 : 0);
                wg_prio = p;
            }
#endif
            wg_word = swap_handshake(e, sl, i, cl, e.slot_of_chain[cl]);
        }
        __syncthreads();                                               // ... and thread 0's acquire before every thread's loads of the next one
        if (wg_word < 0) return;
#if PTE_MW_PRIO
        switch (wg_prio) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
#endif
    }
}

}  // namespace pte
