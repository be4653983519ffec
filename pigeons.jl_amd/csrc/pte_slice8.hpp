// pte_slice8.hpp -- k_explore_slice8: SliceSampler kernel, speculation over stream offsets.
//
// Same draws, decisions and states as every other slice kernel here (bit-for-bit, tests compare
// them), organised around the only true sequential dependence of one Gibbs sweep on the
// scaled-precision MVN path.  With the filtered predicate of pte_slice5.hpp the update of
// coordinate c is a pure function of (x_c, position o_c of the replica's stream at which the update
// starts): the other coordinates enter only the error margin.  The sweep is therefore the pointer
// chase  o_{c+1} = o_c + n_c(x_c, o_c),  and n_c(x_c, .) can be tabulated for every plausible o_c
// BEFORE o_c is known.
//
// One round handles G = 5 consecutive coordinates with the 64 lanes as hypotheses (g, o):
// lane 0 is coordinate l at the known position; 14 / 16 / 17 / 16 lanes cover the positions at
// which coordinates l+1 .. l+4 can start (their windows cover ~3 sigma of the consumed-draw
// distribution).  Every lane runs the complete scalar procedure of the reference
// (SliceSampler.jl:97-237: doubling, shrinkage; the acceptance check of the doubling scheme cannot reject on this path and is
// not executed since round 3 -- the proof is at its place in the round) on its own
// hypothesis, reading pre-converted draws from a 512-draw LDS window of the stream.  Every lane also
// names the lane that follows it on the true path (its draw count fixes where the next coordinate
// starts), so the chase is one v_readlane per level, without branches; the true lanes then store
// their results into the LDS copy of the block.  A hypothesis that meets anything inexact (ambiguous
// filter outcome, ziggurat slow path, window overflow, a budget) is marked invalid and ends the
// chase: its coordinate is lane 0 of the next round, where the certain hypothesis alone may run past
// the budgets and resolves a slow-path exponential exactly; what is still inexact then goes to the
// exact sequential procedure (fixed-tree recompute available), which is also how errors are raised.
// A round retires ~3.5 coordinates for ~450 instructions of one wave.
#pragma once
#include "pte_slice7.hpp"

namespace pte {

// Straight-line variant of k_explore_slice7: the budgeted part of every stage is fully unrolled and
// predicated (no exec-mask loops, no taken branches: a lone wave pays ~35 cycles of refetch per taken
// branch); only the certain hypothesis (lane 0) can continue beyond the budgets, in rarely entered loops.
#ifndef PTE_S8_BD
#define PTE_S8_BD 3
#endif
#ifndef PTE_S8_BS                        // shrinkage budget instantiated by pte.hip (6..10; tuning builds override it)
#define PTE_S8_BS 9
#endif
constexpr int S8_BD = PTE_S8_BD;                 // doubling budget of a speculative hypothesis (2 / 3 / 4: 0.870 / 0.849 / 0.869 ms/scan after the acceptance check went); the shrinkage budget S8_BS is a template parameter

#ifndef PTE_S8_WAVES                     // occupancy hint to the register allocator (waves per SIMD)
#define PTE_S8_WAVES 4
#endif
// The body is shared by the kernels below, which differ in the size of the LDS draw window and in the form of the doubling steps:
// 512 draws convert fewer draws twice (1.2 % faster), 256 draws keep the block at 10 KB of LDS when a GPU holds more than ~2800 replicas.
// The 512-draw kernel needs ~145 VGPRs whatever the occupancy hint says (the hand-scheduled blocks pin v96-v123): 3 resident waves per SIMD,
// all its LDS allows.  The many-replica twin is CAPPED at 128 VGPRs by amdgpu_waves_per_eu(4, 4) and spills up to 16 VGPRs at the deeper
// trees (12-20 B of scratch per lane at d >= 1024, profiles/r04_kernel_resources.txt; 14-20 VGPRs in round 3); the capped, spilling build is the FASTER one (round 3 A/B at N = 3072 / 4096 / 8192: 1.868 /
// 2.311 / 4.211 ms against 1.864 / 2.500 / 4.448 ms with three waves per SIMD and no spill, profiles/r03_twin_kernel_ab.txt).
// None of the spills of either kernel is executed per round: the round loop of k_explore_slice8<4, 9> holds no v_writelane and no
// spill reload (profiles/r04_slice8_round_loop.txt lists every lane instruction of the loop: the five of the chase) -- the 134 spilled
// SGPRs of the resource table live in the prologue, the window refill and the exact sequential procedure.
template <int NLU, int S8_BS, int WINDOW, int DBL_MODE>      // DBL_MODE: form of the budgeted doubling steps (0 selects, 1 EXEC masks, 2 v_cmpx + selects)
__device__ __forceinline__ void slice8_body(EngineDev e, SliceParams sp, const int64_t cl) {      // cl: the local chain this workgroup explores (blockIdx.x in the per-scan kernels)
    using namespace s7;
    constexpr int WIN = WINDOW, REFILL_AT = WINDOW - PTE_S7_MARGIN;
    // FAST: the instantiation for S8_BD < sp.p <= 20 and sp.max_iter >= S8_BS (the launcher checks; SliceSampler's defaults are p = 20,
    // max_iter = 1024), which lets the round drop per-step / per-round tests that cannot fail under these conditions
    constexpr bool FAST = (DBL_MODE == 2);
    static_assert(CAP_ITERS >= S8_BS, "lane 0 must be able to run past the speculative budget");
#ifndef PTE_S8_BLK
#define PTE_S8_BLK 256
#endif
    constexpr int BLK = PTE_S8_BLK, CPB = BLK / 64;  // coordinates per block (rounds do not reach across a block end) = CPB 64-leaf chunks of the tree
    // One struct so that the uniforms sit at LDS offset 0: the nine shrinkage draws of a hypothesis (ds_read2_b64: 8-bit offsets in units
    // of 8 bytes) and its head draws then need no address arithmetic beyond the lane's own index -- seven v_add_u32 per round
    // less, five of them between the doubling steps and the shrinkage block (round 4)
    __shared__ struct {
        double u[WIN];
        double e[WIN];                       // randexp fast-path value; NaN <=> slow path needed
        double x[BLK];                       // the current block: start-of-pass values, overwritten as coordinates retire
        double we[256];
        unsigned long long ke[256];
    } sm;
    double (&s_u)[WIN] = sm.u; double (&s_e)[WIN] = sm.e; double (&s_x)[BLK] = sm.x; double (&s_we)[256] = sm.we; unsigned long long (&s_ke)[256] = sm.ke;
    const int lane = lane_id();
#ifdef PTE_PROFILE_WAVES
    const uint64_t wave_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = lane; i < 256; i += 64) { s_we[i] = ZIG_WE[i]; s_ke[i] = ZIG_KE[i]; }
    __syncthreads();
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    if (is_ref_chain(e, c)) {
        iid_refresh_recorded<NLU>(e, cl, c, slot, e.sd[c], lane);
        return;
    }
    const double lp_before = lp_before_explore(e, c, slot);
    const int64_t d = e.d;
    double *xrow = e.x + (int64_t)slot * e.ld;
    const int B = (int)((d + 63) >> 6);
    const double nhp = e.nhp[c];
    const double inv_nhp = 1.0 / nhp;
    const double inv_abs_nhp = -inv_nhp * (1.0 + 1e-6);
    const double w = sp.w;
    const double w11 = 1.1 * sp.w;
    const int cap_iters = min(sp.max_iter, CAP_ITERS);
    const int kcap = min(sp.p, 20);                    // window headroom: 2 + 20 + 24 draws per hypothesis

    // hypothesis (g, rel) of this lane
    const int hg = s7_level(lane);
    const int hrel = s7_pick(LO, hg, 0) + lane - s7_pick(BASE, hg, 0);

    // the hypothesis that follows this one on the true path starts `cnt` draws later, one coordinate further:
    // lane  succ_base + (cnt + succ_off)  if that falls into the next level's window (never for the last level)
    const int succ_lo = s7_pick(LO, hg + 1, 0), succ_wd = s7_pick(WD, hg + 1, 0), succ_base = s7_pick(BASE, hg + 1, 0);
    const int succ_off = hrel - succ_lo;
    const int word_c = 0x10000 - succ_off;           // chase word of a valid hypothesis = (draws consumed + succ_off) + word_c + successor lane << 24
    const int word_c9 = (1 << 20) - succ_off * 512;   // ... and in the layout of the hand-written tail: (draws consumed + succ_off) << 9 + word_c9 + successor lane
    const uint64_t seed_in = e.rng[2 * slot];

    double BS = 0.0;                                   // lane b: exact fixed-tree sum of block b
    for (int b = 0; b < B; ++b) {
        int64_t i = 64 * (int64_t)b + lane;
        double v = (i < d) ? xrow[i] : 0.0;
        double s = wave_sum_dpp(v * v);
        if (lane == b) BS = s;
    }
    double S = upper_tree_root<NLU>(BS);
    if (nhp * S == -INFINITY) { if (lane == 0) set_error(e, ERR_SLICE_SUPPORT, (int)c, -1); return; }

    // ---- the stream window: s_u[i], s_e[i] = draw #i after `wseed`
    // (Round 4 built a window that resolves the ziggurat's slow-path exponentials when it is FILLED and keeps, per stream position, the whole
    // head of a coordinate starting there -- exponential, the extra draws its slow path consumed, the uniforms behind them: 48 B per position,
    // one LDS round trip as before, every hypothesis passes a slow-path exponential (7 % of the rounds end at one), no per-round test for
    // lane 0's.  Bit-identical, 3.2 % fewer rounds (4.01 -> 4.14 coordinates per round) -- and 3 % SLOWER, 0.789 against 0.765 ms: the
    // 16-byte head reads cost the round 60 cycles and the refill, which now evaluates ~12 slow paths and builds 512 records, the rest.)
    uint64_t wseed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];
    int p = 0;                                         // uniform: next unread draw of the window
    uint64_t gamma_inv = gamma;                        // gamma^-1 mod 2^64 (gamma is odd): Newton, 5 steps
    for (int k = 0; k < 5; ++k) gamma_inv *= 2ull - gamma * gamma_inv;
    // Upper bound on sum x^2 for the filter margins.  An accepted coordinate grows sum x^2 by less than E_c / |nhp| (its
    // new value satisfies v^2 < x_c^2 + E_c / |nhp| + margin), and every E_c taken on the fast path is one of the window's
    // exponentials: so  S(now) <= S(at the last exact point) + (sum of ALL exponentials of the windows used since) / |nhp|.
    // `wsum` = that sum for the current window; Sest is bumped by it at every refill and re-based at block boundaries.
    double wsum = 0.0;
    auto fill_window = [&]() __attribute__((always_inline)) {
        __syncthreads();                               // one wave per block: orders the LDS accesses
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < WIN / 64; ++k) {
            const int i = 64 * k + lane;
            const uint64_t r = mix64(wseed + (uint64_t)(i + 1) * gamma);
            const uint64_t ri = r & MASK52;
            const int idx = (int)(ri & 0xFF);
            const bool fast = ri < s_ke[idx];
            const double ev = (double)ri * s_we[idx];
            s_u[i] = u52_to_unit(r);
            s_e[i] = fast ? ev : __longlong_as_double(0x7ff8000000000000LL);
            acc += fast ? ev : 0.0;
        }
        wsum = wave_sum_dpp(acc);
        p = 0;
        __syncthreads();
    };
    fill_window();

    // recorder sums of the coordinates done by the exact sequential procedure; those of the speculative rounds are
    // derived at the end from the draws consumed (every coordinate draws E and u0, every further draw is one step)
    long long steps_sum = 0, fb_draws = 0, ex_total = 0;     // ex_total: extra draws of slow-path exponentials taken inside rounds
    int steps_n = 0, acc_sum = 0, acc_n = 0, n_fb = 0;
    int err = 0, err_coord = -1;
#ifdef PTE_PROFILE_SECTIONS
    long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int pass = 0; pass < sp.n_passes && !err; ++pass) {
        for (int b = 0; BLK * (int64_t)b < d && !err; ++b) {
            const int64_t base = BLK * (int64_t)b;
            const int nl = (int)min((int64_t)BLK, d - base);
#pragma unroll
            for (int k = 0; k < CPB; ++k) s_x[64 * k + lane] = (64 * k + lane < nl) ? xrow[base + 64 * k + lane] : 0.0;
            __builtin_amdgcn_wave_barrier();
            double Sest = S * (1.0 + 1e-6) + wsum * inv_abs_nhp;       // upper bound on sum x^2 while this window lasts
            int l = 0, l_end = nl;
            // (Requesting the head's LDS loads of the NEXT round as soon as the chase has produced (l, p), so that the stores, exit tests
            // and back edge run in their shadow, was built and measured in round 4: 0.793 against 0.777 ms -- the two extra not-taken
            // branches of the restructured tail cost more than the ~55 exposed cycles of the round trip.)
            do {                                                   // (nl >= 1; one back edge, the refill out of line: a taken branch costs a lone wave ~35 cycles)
                PROF_T(t0);
                if (__builtin_expect(p > REFILL_AT, 0)) { wseed += (uint64_t)p * gamma; fill_window(); Sest = Sest * (1.0 + 1e-6) + wsum * inv_abs_nhp; }
                // ================= speculative round: lane = hypothesis (l + hg, p + hrel) ==========
                // Decisions are sign tests of d(v) = v^2 - Q; every tested |d| is folded into dmin and the
                // hypothesis is valid only if dmin clears the margin at the end (so the loops carry no
                // validity state).  NaNs never pass: they fail the final interval-width test.
                // ---- head: slice level and initial interval (SliceSampler.jl:97-113)
                const bool active = (l + hg) < nl;
                const double xold = s_x[(l + hg) & (BLK - 1)];   // not yet updated in this pass
                int idx0 = p + hrel;
                double E = s_e[idx0];
                // every draw the head and the budgeted doubling steps can need is requested in the same LDS round trip as E
                // (the positions only move for lane 0, in the rare branch below, which re-reads them)
                double u0 = s_u[idx0 + 1];
                double Vd[S8_BD];
#pragma unroll
                for (int it = 0; it < S8_BD; ++it) Vd[it] = s_u[idx0 + 2 + it];
                int ex0 = 0;
                int cnt_base = 2;                            // draws of the head: E and u0 (+ lane 0's extra draws of a slow-path exponential)
                if (__builtin_expect((ballot64(E != E) & 1ull) != 0ull, 0)) {
                    // the certain hypothesis needs the ziggurat's slow path for its exponential (2.3 % of the coordinates):
                    // evaluate it exactly at its stream position; its other draws follow `ex0` positions later
                    SeqRng rs{wseed + (uint64_t)p * gamma, gamma};
                    const double Ex = randexp_seq(rs);
                    const int used = (int)((rs.seed - (wseed + (uint64_t)p * gamma)) * gamma_inv);
                    if (used <= 17) {
                        ex0 = used - 1;
                        if (lane == 0) { E = Ex; idx0 += ex0; cnt_base = 2 + ex0; }
                        Sest += Ex * inv_abs_nhp;            // not among the window's fast-path exponentials summed into Sest
                        ex_total += ex0;
                        u0 = s_u[idx0 + 1];
#pragma unroll
                        for (int it = 0; it < S8_BD; ++it) Vd[it] = s_u[idx0 + 2 + it];
                    }
                }
                const double Q = xold * xold - E * inv_nhp;
                const double Bq = Sest + fabs(Q);
                double dmin = fabs(E * inv_nhp);             // |x_old^2 - Q|: the old position is inside the slice by the margin too (see the acceptance check)
                auto test = [&](double v) __attribute__((always_inline)) -> double {
                    const double d = v * v - Q;
                    dmin = fmin(dmin, fabs(d));
                    return d;                                // inside the slice <=> d < 0
                };
                double LL = xold - w * u0;
                double RR = LL + w;
                double dL = test(LL), dR = test(RR);
                // ---- doubling (:115-139): S8_BD predicated steps for every lane ...
                int kd = 0;
                double dmin_lr = 0.0;                    // min(dL, dR) after the budgeted steps (DBL_MODE 2: left by the hand-written block)
#if PTE_S8_BD >= 1 && PTE_S8_BD <= 4 && !defined(PTE_S8_DOUBLING_SELECTS)
                if constexpr (DBL_MODE == 2) {           // (the launcher picks this instantiation only for sp.p >= S8_BD: no second body in the loop)
                    // Round 4, the one-wave-per-SIMD kernel: a hypothesis that needs no (further) doubling drops out of EXEC by v_cmpx --
                    // written by the VECTOR side, no trip to the scalar side (what made the EXEC-mask form below 2 % slower for a lone
                    // wave) -- which freezes its interval, end values, step count and dmin without a select; only the side (left / right,
                    // per lane) is selected: 20 VALU instructions per step, none scalar, where the pure select form needs 28.  "Needs
                    // doubling" is monotone (a lane that stops never resumes: its values do not change), so EXEC only narrows and is
                    // restored once.  Same operations in the same order per lane: L - (R - L) or R + (R - L) (SliceSampler.jl:123-131),
                    // d = v * v - Q, |d| folded into dmin.  (>= 4 instructions between a VALU write of VCC and its use as a lane mask.)
                    // Fixed registers because the 64-bit selects address register halves; LL / RR / dmin / Q sit where the shrinkage block
                    // below wants Lbar / Rbar / dmin / Q.
                    double t_, wd_, cl_, cr_, cd_, dc_;
                    uint64_t sv_;
#define PTE_S8_CSTEP(V) \
                    "v_min_f64 v[102:103], v[114:115], v[116:117]\n" \
                    "v_cmpx_gt_f64 vcc, 0, v[102:103]\n" \
                    "v_cmp_ge_f64 vcc, 0.5, " V "\n" \
                    "v_add_f64 v[112:113], v[98:99], -v[96:97]\n" \
                    "v_add_u32 %[kd], 1, %[kd]\n" \
                    "v_add_f64 v[118:119], v[96:97], -v[112:113]\n" \
                    "v_add_f64 v[120:121], v[98:99], v[112:113]\n" \
                    "v_cndmask_b32 v100, v120, v118, vcc\n" \
                    "v_cndmask_b32 v101, v121, v119, vcc\n" \
                    "v_cndmask_b32 v96, v96, v118, vcc\n" \
                    "v_cndmask_b32 v97, v97, v119, vcc\n" \
                    "v_mul_f64 v[102:103], v[100:101], v[100:101]\n" \
                    "v_cndmask_b32 v98, v120, v98, vcc\n" \
                    "v_cndmask_b32 v99, v121, v99, vcc\n" \
                    "v_add_f64 v[122:123], v[102:103], -v[110:111]\n" \
                    "v_cndmask_b32 v114, v114, v122, vcc\n" \
                    "v_cndmask_b32 v115, v115, v123, vcc\n" \
                    "v_cndmask_b32 v116, v122, v116, vcc\n" \
                    "v_cndmask_b32 v117, v123, v117, vcc\n" \
                    "v_min_f64 v[104:105], v[104:105], |v[122:123]|\n"
                    asm volatile("s_mov_b64 %[sv], exec\n"
                                 PTE_S8_CSTEP("%[V0]")
#if PTE_S8_BD >= 2
                                 PTE_S8_CSTEP("%[V1]")
#endif
#if PTE_S8_BD >= 3
                                 PTE_S8_CSTEP("%[V2]")
#endif
#if PTE_S8_BD >= 4
                                 PTE_S8_CSTEP("%[V3]")
#endif
                                 "s_mov_b64 exec, %[sv]\n"
                                 "v_min_f64 v[102:103], v[114:115], v[116:117]\n"      // (all lanes: what "still needs doubling" is decided on below)
                                 : "+{v[96:97]}"(LL), "+{v[98:99]}"(RR), "+{v[114:115]}"(dL), "+{v[116:117]}"(dR), "+{v[104:105]}"(dmin), [kd] "+v"(kd),
                                   "=&{v[102:103]}"(t_), "=&{v[112:113]}"(wd_), "=&{v[118:119]}"(cl_), "=&{v[120:121]}"(cr_), "=&{v[100:101]}"(cd_),
                                   "=&{v[122:123]}"(dc_), [sv] "=&s"(sv_)
                                 : "{v[110:111]}"(Q), [V0] "v"(Vd[0]), [V1] "v"(Vd[S8_BD > 1 ? 1 : 0]), [V2] "v"(Vd[S8_BD > 2 ? 2 : 0]), [V3] "v"(Vd[S8_BD > 3 ? 3 : 0])
                                 : "vcc", "scc");
#undef PTE_S8_CSTEP
                    dmin_lr = t_;
                } else
                if (DBL_MODE == 1 && sp.p >= S8_BD) {    // (uniform; compile-time per kernel)
                    // Hand-written: a step runs under EXEC = "this hypothesis still needs doubling", the left / right extension under
                    // EXEC = need & left / need & ~left -- 13 VALU instructions per step where the select form below needs 28 (sixteen of
                    // them v_cndmask halves).  Same operations in the same order per lane: L - (R - L) or R + (R - L)
                    // (SliceSampler.jl:123-131), d = v * v - Q, |d| folded into dmin.  Which form is faster depends on what else the SIMD
                    // has to do: each step of this form crosses twice from the vector to the scalar side (v_cmp -> s_and_saveexec), ~40
                    // cycles each for a wave ALONE on its SIMD, nothing when other waves fill the gap -- measured at N = 1024 / 2048 /
                    // 4096 / 8192 replicas per GPU: 0.868 / 1.141 / 1.853 / 3.395 ms per scan in this form against 0.851 / 1.167 / 1.950 /
                    // 3.566 in the select form.  Choosing at RUN time inside one kernel costs more than either (0.897 at N = 1024: both bodies in the
                    // loop), so the choice is per kernel: selects in k_explore_slice8 (up to 2816 replicas), EXEC masks in the many-replica twin.
                    double t_, wd_;
                    uint64_t sv_, sv2_;
#define PTE_S8_DSTEP(V) \
                    "v_min_f64 %[t], %[dL], %[dR]\n" \
                    "v_cmp_gt_f64 vcc, 0, %[t]\n" \
                    "s_and_saveexec_b64 %[sv], vcc\n" \
                    "v_add_u32 %[kd], 1, %[kd]\n" \
                    "v_add_f64 %[wd], %[RR], -%[LL]\n" \
                    "v_cmp_ge_f64 vcc, 0.5, " V "\n" \
                    "s_and_saveexec_b64 %[sv2], vcc\n" \
                    "v_add_f64 %[LL], %[LL], -%[wd]\n" \
                    "v_mul_f64 %[t], %[LL], %[LL]\n" \
                    "v_add_f64 %[dL], %[t], -%[Q]\n" \
                    "v_min_f64 %[dmin], %[dmin], |%[dL]|\n" \
                    "s_andn2_b64 exec, %[sv2], exec\n" \
                    "v_add_f64 %[RR], %[RR], %[wd]\n" \
                    "v_mul_f64 %[t], %[RR], %[RR]\n" \
                    "v_add_f64 %[dR], %[t], -%[Q]\n" \
                    "v_min_f64 %[dmin], %[dmin], |%[dR]|\n" \
                    "s_mov_b64 exec, %[sv]\n"
                    asm volatile(PTE_S8_DSTEP("%[V0]")
#if PTE_S8_BD >= 2
                                 PTE_S8_DSTEP("%[V1]")
#endif
#if PTE_S8_BD >= 3
                                 PTE_S8_DSTEP("%[V2]")
#endif
#if PTE_S8_BD >= 4
                                 PTE_S8_DSTEP("%[V3]")
#endif
                                 : [LL] "+v"(LL), [RR] "+v"(RR), [dL] "+v"(dL), [dR] "+v"(dR), [dmin] "+v"(dmin), [kd] "+v"(kd),
                                   [t] "=&v"(t_), [wd] "=&v"(wd_), [sv] "=&s"(sv_), [sv2] "=&s"(sv2_)
                                 : [Q] "v"(Q), [V0] "v"(Vd[0]), [V1] "v"(Vd[S8_BD > 1 ? 1 : 0]), [V2] "v"(Vd[S8_BD > 2 ? 2 : 0]), [V3] "v"(Vd[S8_BD > 3 ? 3 : 0])
                                 : "vcc", "scc");        // (s_and_saveexec / s_andn2 write SCC)
#undef PTE_S8_DSTEP
                } else
#endif
#pragma unroll
                for (int it = 0; it < S8_BD; ++it) {
                    const bool need = (fmin(dL, dR) < 0.0) && it < sp.p;
                    const double V = Vd[it];                  // (a step is needed only if every earlier one was: draw #it)
                    kd += need ? 1 : 0;
                    const bool left = V <= 0.5;
                    const double wd = RR - LL;
                    const double cand = left ? (LL - wd) : (RR + wd);
                    const double dc = cand * cand - Q;
                    dmin = need ? fmin(dmin, fabs(dc)) : dmin;
                    const bool nl_ = need && left, nr_ = need && !left;
                    LL = nl_ ? cand : LL;
                    RR = nr_ ? cand : RR;
                    dL = nl_ ? dc : dL;
                    dR = nr_ ? dc : dR;
                }
#if !(PTE_S8_BD >= 1 && PTE_S8_BD <= 4) || defined(PTE_S8_DOUBLING_SELECTS)
                // the hand-written block is compiled out in this variant build: DBL_MODE 2 would run the select loop above and leave dmin_lr at 0.0
                // (dbl_ok always true, lane 0's further doublings never run -- silently wrong samples); such builds must ask for DBL_MODE 0 or 1
                static_assert(DBL_MODE != 2, "PTE_S8_DBL_MODE == 2 needs the v_cmpx doubling block: 1 <= PTE_S8_BD <= 4 and no PTE_S8_DOUBLING_SELECTS");
#endif
                if constexpr (DBL_MODE != 2) dmin_lr = fmin(dL, dR);
                // ... and the rest for the certain hypothesis only.  (Round 4 built the alternative -- no test inside the round, a round whose
                // lane 0 ran out of a budget or met a slow-path exponential redoes lane 0 without budgets out of line and runs the tail again --
                // and measured 0.948 against 0.762 ms: lane 0 IS the hypothesis that ended the chase before, a quarter of the rounds need it.)
                if (__builtin_expect(ballot64(lane == 0 && (dmin_lr < 0.0) && kd < kcap) != 0ull, 0)) {
                    bool need = (lane == 0);
                    while (need) {
                        const double V = s_u[idx0 + 2 + kd];
                        kd += 1;
                        const bool left = V <= 0.5;
                        const double wd = RR - LL;
                        const double cand = left ? (LL - wd) : (RR + wd);
                        const double dc = test(cand);
                        LL = left ? cand : LL;
                        RR = left ? RR : cand;
                        dL = left ? dc : dL;
                        dR = left ? dR : dc;
                        need = (kd < kcap) && (fmin(dL, dR) < 0.0);
                    }
                    asm volatile("" : "+v"(dL), "+v"(dR), "+v"(kd));
                    dmin_lr = fmin(dL, dR);
                    if constexpr (FAST) dmin_lr = (kd >= sp.p) ? 0.0 : dmin_lr;   // (the reference's own limit p ended it: not a budget)
                }
                // ended by itself, not by a budget.  FAST (S8_BD < p <= 20): a speculative lane has kd <= S8_BD < p, and lane 0 leaves
                // the loop above either satisfied or at kd = p, so "still needs doubling" alone decides
                const bool dbl_ok = FAST ? !(dmin_lr < 0.0) : !((kd < sp.p) && (dmin_lr < 0.0));
                double thr2 = 1e-6 * fmax(fabs(LL), fabs(RR));
                if constexpr (DBL_MODE == 2) asm volatile("" : "+v"(thr2));       // (taken here: LL / RR then live on only as the shrinkage block's bracket, in place)
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(LL), "v"(RR), "v"(kd), "v"(thr2));
#endif
                PROF_T(t1); PROF_ADD(0, t1 - t0);
                // ---- shrinkage (:141-190) up to the first proposal inside the slice: S8_BS predicated steps,
                //      their draws loaded up front (consecutive stream positions of this hypothesis)
                const double *us = &s_u[idx0 + 2 + kd];
                double u[S8_BS];
#pragma unroll
                for (int k = 0; k < S8_BS; ++k) u[k] = us[k];
                // In the shadow of that LDS round trip (a lone wave has nothing else to put there): everything the validity test and the
                // chase word need that does not depend on the shrinkage -- the margin, the draws consumed so far relative to the
                // successor's window, "is this coordinate in the block", "did the doubling end by itself" as lane masks
                double mthr = 2e-12 * Bq;
                int kn0 = cnt_base + kd + succ_off;          // + n: offset of the successor in the next level's window
                uint64_t pre_ok = ballot64(active && dbl_ok);
                if constexpr (FAST) asm volatile("" : "+v"(mthr), "+v"(kn0), "+s"(pre_ok));
                double Lbar = LL, Rbar = RR, xf = xold, W = 0.0;
                int n = 0;
                bool fin = false;
                uint64_t fin_m = 0ull;                       // `fin` as a lane mask, for the hand-written tail (hipcc moves a mask through v_cndmask + v_cmp otherwise)
                if constexpr (S8_BS == PTE_S8_BS && S8_BS >= 6 && S8_BS <= 10) {
                    // Hand-scheduled: a lane whose proposal lands inside the slice drops out of EXEC (v_cmpx), which freezes
                    // its result, step count and bracket -- no per-step selects, no mask arithmetic, no branches.  Fixed
                    // registers because the 64-bit selects address register halves.  (>= 2 instructions between a VALU
                    // write of VCC and its use as a lane mask; EXEC restored before the block ends.)
                    double t_;
                    uint64_t fin_mask, exec_save;
#define PTE_S8_STEP(U) \
                    "v_add_f64 v[112:113], v[98:99], -v[96:97]\n" \
                    "v_mul_f64 v[102:103], " U ", v[112:113]\n" \
                    "v_add_f64 v[100:101], v[96:97], v[102:103]\n" \
                    "v_cmp_lt_f64 vcc, v[100:101], v[108:109]\n" \
                    "v_mul_f64 v[102:103], v[100:101], v[100:101]\n" \
                    "v_add_u32 v106, 1, v106\n" \
                    "v_add_f64 v[102:103], v[102:103], -v[110:111]\n" \
                    "v_cndmask_b32 v96, v96, v100, vcc\n" \
                    "v_cndmask_b32 v97, v97, v101, vcc\n" \
                    "v_cndmask_b32 v98, v100, v98, vcc\n" \
                    "v_cndmask_b32 v99, v101, v99, vcc\n" \
                    "v_min_f64 v[104:105], v[104:105], |v[102:103]|\n" \
                    "v_cmpx_ngt_f64 vcc, 0, v[102:103]\n"
                    asm volatile(".p2align 3\n"           // hand-written stream at an 8-byte phase (MI355X_MICROARCH.md: the 4 mod 8 phase costs ~1 % here)
                                 "s_mov_b64 %[sv], exec\n"
                                 PTE_S8_STEP("%[u0]") PTE_S8_STEP("%[u1]") PTE_S8_STEP("%[u2]") PTE_S8_STEP("%[u3]")
                                 PTE_S8_STEP("%[u4]") PTE_S8_STEP("%[u5]")
#if PTE_S8_BS >= 7
                                 PTE_S8_STEP("%[u6]")
#endif
#if PTE_S8_BS >= 8
                                 PTE_S8_STEP("%[u7]")
#endif
#if PTE_S8_BS >= 9
                                 PTE_S8_STEP("%[u8]")
#endif
#if PTE_S8_BS >= 10
                                 PTE_S8_STEP("%[u9]")
#endif
                                 "s_andn2_b64 %[fin], %[sv], exec\n"
                                 "s_mov_b64 exec, %[sv]\n"
                                 "s_nop 3\n"
                                 : "+{v[96:97]}"(Lbar), "+{v[98:99]}"(Rbar), "=&{v[100:101]}"(xf), "=&{v[102:103]}"(t_),
                                   "+{v[104:105]}"(dmin), "+{v106}"(n), "=&{v[112:113]}"(W), [fin] "=&s"(fin_mask), [sv] "=&s"(exec_save)
                                 : "{v[108:109]}"(xold), "{v[110:111]}"(Q), [u0] "v"(u[0]), [u1] "v"(u[1]), [u2] "v"(u[2]), [u3] "v"(u[3]),
                                   [u4] "v"(u[4]), [u5] "v"(u[5]), [u6] "v"(u[S8_BS > 6 ? 6 : 0]), [u7] "v"(u[S8_BS > 7 ? 7 : 0]),
                                   [u8] "v"(u[S8_BS > 8 ? 8 : 0]), [u9] "v"(u[S8_BS > 9 ? 9 : 0])
                                 : "vcc", "scc");        // (s_andn2 writes SCC)
#undef PTE_S8_STEP
                    fin = __builtin_amdgcn_inverse_ballot_w64(fin_mask);
                    fin_m = fin_mask;
                } else {
#pragma unroll
                    for (int k = 0; k < S8_BS; ++k) {
                        W = Rbar - Lbar;
                        const double v = Lbar + u[k] * W;
                        const double dv = v * v - Q;
                        dmin = fmin(dmin, fabs(dv));              // (after `fin` too: only ever makes the filter more conservative)
                        xf = fin ? xf : v;
                        n += fin ? 0 : 1;
                        const bool below = v < xold;
                        Lbar = below ? v : Lbar;                 // (after `fin` these only shrink further: harmless)
                        Rbar = below ? Rbar : v;
                        fin = fin || (dv < 0.0);
                    }
                    fin_m = ballot64(fin);
                }
                if (__builtin_expect(ballot64(lane == 0 && !fin && n < cap_iters) != 0ull, 0)) {
                    // the certain hypothesis continues from its state after S8_BS rejected proposals
                    double dx = 1.0;
                    bool go = (lane == 0);
                    while (go) {
                        W = Rbar - Lbar;
                        xf = Lbar + us[n] * W;
                        n += 1;
                        dx = test(xf);
                        const bool below = xf < xold;
                        Lbar = below ? xf : Lbar;
                        Rbar = below ? Rbar : xf;
                        go = !(dx < 0.0) && n < cap_iters;
                    }
                    asm volatile("" : "+v"(dx), "+v"(n), "+v"(W));
                    fin = fin || (lane == 0 && dx < 0.0);
                    fin_m = ballot64(fin);
                }
                // W > thr2 at the last step (widths only shrink) rules out isapprox(Lbar, Rbar) at every step
                // FAST drops two tests that cannot fail there: a NaN exponential makes Q, every d, dmin and Bq NaN, so it fails the margin
                // test below (and `fin`); n <= S8_BS <= max_iter for a speculative lane, and lane 0's loop above stops at cap_iters
                PROF_T(t2); PROF_ADD(1, t2 - t1); PROF_ADD(3, 1);      // (validity + chase are one section since round 4: slot 5)
                int gdone;
#ifdef PTE_PROFILE_SECTIONS
                constexpr bool ASM_TAIL = false;
#else
                constexpr bool ASM_TAIL = FAST && G == 5;
#endif
                if constexpr (ASM_TAIL) {
                    // ================= validity + chase, hand-written (round 4) ===========================
                    // What tools/ubench/round_cost.hip measured for a lone wave: the validity test as hipcc writes it -- compares into
                    // scalar pairs combined by s_and -- costs 86 cycles for 8 instructions (every compare -> s_and is a trip from the
                    // vector to the scalar side), the same conditions applied as a chain of selects on the word itself 40; and a hop of
                    // the chase costs 34-42 cycles when scalar instructions stand between the v_readlane that produces a lane select and
                    // the one that consumes it, 27.5 when the word read IS the next lane select (successor in bits 5:0, which is all
                    // v_readlane and s_bitset1 look at) and the four wait states the hardware demands there (tools/ubench/hop_check.hip:
                    // without them the chase reads the wrong lanes) are filled by the bookkeeping of the hop before.
                    // word of a valid hypothesis: bits 0-5 its successor lane (0: none), bits 9-17 its draw count, bit 20 "one more
                    // coordinate done" -- the five words of the path are summed as they stand (successors pile up below bit 9, counts
                    // below bit 18).  An invalid hypothesis has word 0; so has lane 0 in `pk`, where a broken path ends up.
                    const int kn = kn0 + n;
                    const int wsucc = ((unsigned)kn < (unsigned)succ_wd) ? kn + succ_base : 0;
                    const int word_v = (kn << 9) + word_c9 + wsucc;
                    unsigned acc; uint64_t tmask, sA_, sB_; int w_, pk_, s0_, s1_, s2_, s3_, s4_;
                    asm volatile("v_cmp_gt_f64 vcc, %[W], %[thr]\n"
                                 "v_cndmask_b32_e64 %[w], 0, %[wv], %[pre]\n"          // in the block, doubling ended by itself
                                 "v_cmp_gt_f64_e64 %[sA], %[dm], %[mthr]\n"
                                 "v_cndmask_b32_e64 %[w], 0, %[w], %[fin]\n"           // a proposal inside the slice within the budget
                                 "v_cndmask_b32_e32 %[w], 0, %[w], vcc\n"              // interval still wider than the isapprox threshold
                                 "s_nop 0\n"
                                 "v_cndmask_b32_e64 %[w], 0, %[w], %[sA]\n"            // every decision clears the filter's margin
                                 "v_cndmask_b32_e64 %[pk], 0, %[w], %[nz]\n"
                                 : [w] "=&v"(w_), [pk] "=&v"(pk_), [sA] "=&s"(sA_)
                                 : [wv] "v"(word_v), [pre] "s"(pre_ok), [fin] "s"(fin_m), [nz] "s"(0xFFFFFFFFFFFFFFFEull),
                                   [W] "v"(W), [thr] "v"(thr2), [dm] "v"(dmin), [mthr] "v"(mthr)
                                 : "vcc");
                    // (a second statement: with vector outputs in the same one hipcc takes the scalar results for divergent and routes
                    // p and l through the vector side -- v_bfe, v_add, v_readfirstlane -- on the way to the next round)
                    asm volatile("v_cmp_ne_u32_e64 %[sB], 0, %[w]\n"
                                 "s_mov_b64 %[tm], 1\n"
                                 "v_readlane_b32 %[s0], %[w], 0\n"
                                 "s_nop 3\n"
                                 "v_readlane_b32 %[s1], %[pk], %[s0]\n"
                                 "s_bitset1_b64 %[tm], %[s0]\n"
                                 "s_mov_b32 %[acc], %[s0]\n"
                                 "s_nop 1\n"
                                 "v_readlane_b32 %[s2], %[pk], %[s1]\n"
                                 "s_bitset1_b64 %[tm], %[s1]\n"
                                 "s_add_u32 %[acc], %[acc], %[s1]\n"
                                 "s_nop 1\n"
                                 "v_readlane_b32 %[s3], %[pk], %[s2]\n"
                                 "s_bitset1_b64 %[tm], %[s2]\n"
                                 "s_add_u32 %[acc], %[acc], %[s2]\n"
                                 "s_nop 1\n"
                                 "v_readlane_b32 %[s4], %[pk], %[s3]\n"
                                 "s_bitset1_b64 %[tm], %[s3]\n"
                                 "s_add_u32 %[acc], %[acc], %[s3]\n"
                                 "s_and_b64 %[tm], %[tm], %[sB]\n"                      // (a path ends AT an invalid lane: its bit was set above)
                                 "s_add_u32 %[acc], %[acc], %[s4]\n"
                                 : [sB] "=&s"(sB_), [tm] "=&s"(tmask), [acc] "=&s"(acc),
                                   [s0] "=&s"(s0_), [s1] "=&s"(s1_), [s2] "=&s"(s2_), [s3] "=&s"(s3_), [s4] "=&s"(s4_)
                                 : [w] "v"(w_), [pk] "v"(pk_)
                                 : "scc");
                    if (__builtin_amdgcn_inverse_ballot_w64(tmask)) s_x[(l + hg) & (BLK - 1)] = xf;
                    __builtin_amdgcn_wave_barrier();
                    p += (int)((acc >> 9) & 0x1FFu);
                    gdone = (int)((acc >> 20) & 7u);
                    l += gdone;
                } else {
                bool valid = __builtin_amdgcn_inverse_ballot_w64(pre_ok) && fin && (W > thr2);
                if constexpr (!FAST) valid = valid && !(E != E) && n <= cap_iters;                  // (max_iter < S8_BS: exact path raises)
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(xf), "v"(n), "v"(dmin));
#endif
                // ---- acceptance check of the doubling scheme (:192-237): provably a no-op on this path, so it is not executed.
                //      slice_accept rejects iff at some halving the old and the new position lie on different sides of the midpoint
                //      (D) and BOTH ends of the halved interval are outside the slice.  Once D holds, the end on the old position's
                //      side is a midpoint strictly between the new and the old position.  Here the slice { v : z < lp(v) } is
                //      { v : v^2 < Q' } for the reference's own floating-point predicate too (the fixed-tree sum is monotone in
                //      |v|: every rounding step is), an interval -- and both positions are inside it with the filter's margin to
                //      spare (the new one by its own test, the old one because |x_old^2 - Q| = E / |nhp| was folded into dmin
                //      above), so that end is inside and the test cannot fire.  Measured before the removal: 0 rejections in every
                //      profile (profiles/r03_slice8_sections_by_chain.txt), while the region cost the hot chains (precision near 1,
                //      where most intervals are doubled) ~8 % of their time -- they were the launch's slowest waves
                //      (profiles/r03_slice8_per_wave.txt).  The exact sequential procedure below still runs the reference's test.
#ifdef PTE_PROFILE_SECTIONS
                {   // why the true path ends where it ends (slots 8..15: all 5 levels done, then the causes)
                    const bool mg = dmin > 2e-12 * Bq;
                    const int rc = !active ? 7 : (E != E) ? 1 : !dbl_ok ? 2 : !fin ? 3 : !valid ? 4 : !mg ? 5 : 0;
                    const int cnt_ = cnt_base + kd + n;
                    int o_ = 0, g_ = 0, why = 0;
                    for (g_ = 0; g_ < G; ++g_) {
                        if (l + g_ >= nl) { why = 7; break; }
                        const int k_ = o_ - LO[g_];
                        if ((unsigned)k_ >= (unsigned)WD[g_]) { why = 6; break; }
                        const int ln_ = BASE[g_] + k_;
                        const int r_ = __builtin_amdgcn_readlane(rc, ln_);
                        if (r_) { why = r_; break; }
                        o_ += __builtin_amdgcn_readlane(cnt_, ln_);
                    }
                    PROF_ADD(8 + why, 1);
                }
#endif
                valid = valid && (dmin > mthr);
                // ================= chase the true path through the hypotheses =======================
                // Branch free: every lane names its successor, the chase is one v_readlane per level.  An invalid
                // hypothesis packs 0, so a broken path falls back to lane 0, which never carries the level >= 1 flag.
                const int kn = kn0 + n;
                const bool inw = (unsigned)kn < (unsigned)succ_wd;
                const uint64_t vmask = ballot64(valid);
                // word of a valid hypothesis: bits 0-7 its draw count, bit 16 "one more coordinate done", bits 24-29 its successor lane
                // (0: none) -- ONE add per level accumulates the draws consumed (low half) and the coordinates retired (bits 16-18) of
                // the true path (the successor fields pile up above bit 24, out of the way), one shift yields the next lane select
                // (v_readlane reads six bits of it).  Lane 0 is read through `word`; in `packed` it is 0, because a broken path (an
                // invalid hypothesis has word 0) falls back to lane 0, which must then contribute nothing.  Round 4: the level-0 hop
                // used to be re-derived on the scalar side (window test, two selects), the true lanes' mask built by shift + or, the
                // fields masked before every add: 16 scalar instructions per round less.
                int word_v = kn + word_c + (inw ? (int)((unsigned)(kn + succ_base) << 24) : 0);     // = 0x10000 | cnt | successor << 24  (cnt = kn - succ_off < 256)
                asm("" : "+v"(word_v));                       // (computed for every lane and selected: hipcc otherwise wraps it into an EXEC-masked branch)
                const int word = valid ? word_v : 0;
                const int packed = (lane != 0) ? word : 0;
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(packed));
#endif
                {
                    int pk = __builtin_amdgcn_readlane(word, 0);
                    unsigned acc = 0u;
                    uint64_t tmask = 1ull;
#pragma unroll
                    for (int g = 1; g < G; ++g) {
                        const int cur = (int)((unsigned)pk >> 24);
                        // tmask |= 1 << (cur & 63); the sum taken here, between the hops (hipcc leaves all the adds behind the last hop, on the path to the next round's loads)
                        asm volatile("s_bitset1_b64 %0, %2\n\ts_add_u32 %1, %1, %3" : "+s"(tmask), "+s"(acc) : "s"(cur), "s"(pk) : "scc");
                        pk = __builtin_amdgcn_readlane(packed, cur);
                    }
                    acc += (unsigned)pk;
                    tmask &= vmask;                           // (a path ends AT an invalid lane: its bit was set above)
                    if (__builtin_amdgcn_inverse_ballot_w64(tmask)) s_x[(l + hg) & (BLK - 1)] = xf;
                    __builtin_amdgcn_wave_barrier();
                    p += (int)(acc & 0xFFFFu);
                    gdone = (int)((acc >> 16) & 0xFFu);
                    l += gdone;
                }
                }
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "s"(p), "s"(l));
#endif
                PROF_T(t4); PROF_ADD(5, t4 - t2); PROF_ADD(4, gdone);
                if (__builtin_expect(gdone == 0, 0)) {
                    // ================= exact sequential procedure for coordinate l ===================
                    ex_total -= ex0;                              // (its exponential is drawn again below)
                    SeqRng rs{wseed + (uint64_t)p * gamma, gamma};
                    // (the block's chunk sums have not been refreshed since the block started: do it now, for the exact tree)
                    const int ck = l >> 6, li = l & 63, bc = CPB * b + ck;
#pragma unroll
                    for (int k = 0; k < CPB; ++k) {
                        const double xk = s_x[64 * k + lane];
                        const double sk = wave_sum_dpp(xk * xk);
                        if (lane == CPB * b + k) BS = sk;
                    }
                    const double X = s_x[64 * ck + lane];
                    const double xo = readlane_f64(X, li);
                    const double E = randexp_seq(rs);
                    const double u0 = rs.rand();
                    const double Q = xo * xo - E * inv_nhp;
                    const double mg = 1e-12 * (Sest + fabs(Q));
                    const double Qlo = Q - mg, Qhi = Q + mg;
                    auto inside_exact = [&](double v) __attribute__((always_inline)) -> bool {
                        const double Xv = (lane == li) ? v : X;
                        const double sv = wave_sum_dpp(Xv * Xv), s0 = wave_sum_dpp(X * X);
                        const double Sv = upper_tree_root<NLU>((lane == bc) ? sv : BS);
                        const double S0 = upper_tree_root<NLU>((lane == bc) ? s0 : BS);
                        const double zz = nhp * S0 - E;
                        return zz < nhp * Sv;
                    };
                    auto inside = [&](double v) __attribute__((always_inline)) -> bool {
                        const double q = v * v;
                        const bool in = q < Qlo;
                        const bool out = q > Qhi;
                        if (__builtin_expect(!(in || out), 0)) return inside_exact(v);
                        return in;
                    };
                    double LL = xo - w * u0;
                    double RR = LL + w;
                    bool in_L = inside(LL), in_R = inside(RR);
                    int K = sp.p;
                    while (K > 0 && (in_L || in_R)) {
                        const double V = rs.rand();
                        if (V <= 0.5) { LL = LL - (RR - LL); in_L = inside(LL); }
                        else { RR = RR + (RR - LL); in_R = inside(RR); }
                        K -= 1;
                    }
                    steps_sum += (sp.p - K); steps_n += 1;
                    const bool doubled = (RR - LL) > w11;
                    double Lbar = LL, Rbar = RR;
                    double xn = xo;
                    bool fin = false;
                    for (int n = 1; n <= sp.max_iter; ++n) {
                        const double newpos = Lbar + rs.rand() * (Rbar - Lbar);
                        if (inside(newpos)) {
                            bool ok = true;
                            if (doubled) {
                                double Lhat = LL, Rhat = RR;
                                bool oL = !in_L, oR = !in_R;
                                bool Rstale = false, Lstale = false, D = false;
                                while (Rhat - Lhat > w11) {
                                    const double Mid = (Lhat + Rhat) / 2.0;
                                    if ((xo < Mid && newpos >= Mid) || (xo >= Mid && newpos < Mid)) D = true;
                                    if (newpos < Mid) { Rhat = Mid; Rstale = true; }
                                    else { Lhat = Mid; Lstale = true; }
                                    if (D) {
                                        if (Lstale) { oL = !inside(Lhat); Lstale = false; }
                                        if (Rstale) { oR = !inside(Rhat); Rstale = false; }
                                        if (oL && oR) { ok = false; break; }
                                    }
                                }
                            }
                            acc_n += 1;
                            if (ok) { acc_sum += 1; xn = newpos; steps_sum += n; steps_n += 1; fin = true; break; }
                        }
                        if (newpos < xo) Lbar = newpos; else Rbar = newpos;
                        if (jl_isapprox(Lbar, Rbar)) { steps_sum += n; steps_n += 1; fin = true; break; }
                    }
                    if (!fin) { err = ERR_SLICE_MAX_ITER; err_coord = (int)(base + l); l_end = 0; continue; }   // (leaves through the loop's ONE exit test: a `break` makes hipcc carry a "no error" flag over the back edge of every round)
                    Sest = Sest + fabs(xn * xn - xo * xo);
                    if (lane == li) s_x[l] = xn;
                    __builtin_amdgcn_wave_barrier();
                    l += 1;
                    const int used = (int)((rs.seed - (wseed + (uint64_t)p * gamma)) * gamma_inv);   // draws consumed; window stays
                    p += used;
                    fb_draws += used; n_fb += 1;
                    PROF_T(t5); PROF_ADD(7, t5 - t4); PROF_ADD(6, 1);
                }
            } while (l < l_end);
            if (err) break;
            {   // write the block back and re-establish the exact fixed-tree values at the block boundary
#pragma unroll
                for (int k = 0; k < CPB; ++k) {
                    const double X = s_x[64 * k + lane];
                    if (64 * k + lane < nl) xrow[base + 64 * k + lane] = X;
                    const double s = wave_sum_dpp(X * X);
                    if (lane == CPB * b + k) BS = s;
                }
                S = upper_tree_root<NLU>(BS);
                if (__builtin_expect(!isfinite(nhp * S), 0)) { err = ERR_SLICE_INVALID_LP; err_coord = (int)base; }
            }
        }
    }
    if (err) { if (lane == 0) set_error(e, err, (int)c, err_coord); return; }
    if (lane == 0) {
        const uint64_t seed_out = wseed + (uint64_t)p * gamma;
        const long long n_spec = (long long)sp.n_passes * d - n_fb;                               // coordinates retired by speculative rounds
        const long long spec_draws = (long long)((seed_out - seed_in) * gamma_inv) - fb_draws;
        steps_sum += spec_draws - ex_total - 2 * n_spec; steps_n += (int)(2 * n_spec); acc_sum += (int)n_spec; acc_n += (int)n_spec;
        e.suff[slot] = S;
        e.rng[2 * slot] = seed_out;
        e.expl_steps_sum[cl] += (double)steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += (double)acc_sum;     e.expl_acc_n[cl] += acc_n;
#ifdef PTE_PROFILE_SECTIONS
        for (int i = 0; i < 16; ++i) e.on_m2[16 * cl + i] += (double)prof[i];   // debug builds only (needs d >= 16K)
#endif
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
#ifdef PTE_PROFILE_WAVES
    if (lane == 0) {
        double *o = e.on_m2 + 2 * (e.d + 1) + 4 * cl;
        o[0] = (double)wave_t0; o[1] = (double)__builtin_amdgcn_s_memrealtime();
        o[2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4); o[3] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

#ifndef PTE_S8_DBL_MODE
#define PTE_S8_DBL_MODE 2
#endif
template <int NLU, int S8_BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTE_S8_WAVES, PTE_S8_WAVES))) void k_explore_slice8(EngineDev e, SliceParams sp) {
    slice8_body<NLU, S8_BS, PTE_S7_WIN, PTE_S8_DBL_MODE>(e, sp, blockIdx.x);        // PTE_S8_DBL_MODE == 2 requires S8_BD < sp.p <= 20 and sp.max_iter >= S8_BS (launch_explore checks)
}
template <int NLU, int S8_BS>           // any p / max_iter (non-default SliceSampler(p = 1 .. 3, p > 20, max_iter < 9)): the select form tests it < p per step, the round keeps every validity test
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTE_S8_WAVES, PTE_S8_WAVES))) void k_explore_slice8_generic(EngineDev e, SliceParams sp) {
    slice8_body<NLU, S8_BS, PTE_S7_WIN, 0>(e, sp, blockIdx.x);
}
#ifndef PTE_S8_TWIN_WAVES
#define PTE_S8_TWIN_WAVES PTE_S8_WAVES
#endif
#ifndef PTE_S8_TWIN_FROM                 // more local replicas than this run the 10 KB-LDS twin
// Two waves of the 512-draw kernel per SIMD: scheduled for instruction-level parallelism (-amdgpu-sched-strategy=max-ilp: 1.3-2.3 % faster with one
// wave per SIMD) it holds 189 VGPRs, and a third wave per SIMD would wait for the first two -- 2304 .. 2816 replicas took 1.58 ms per scan at
// d = 1024 where the twin takes 1.45 (its LDS allowed 11 replicas per CU = 2816, the bound of rounds 2-3).
#define PTE_S8_TWIN_FROM (256 * 8)
#endif
template <int NLU, int S8_BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTE_S8_TWIN_WAVES, PTE_S8_TWIN_WAVES))) void k_explore_slice8_lds10k(EngineDev e, SliceParams sp) {
    slice8_body<NLU, S8_BS, 256, 1>(e, sp, blockIdx.x);
}

// ---- one launch per pte_run_scans (pte_kernels.hpp, "ScanLoop"): workgroup c explores chain c and then takes part in the swap of its own
// pair, for all the scans of the call.  Same body, same launch bounds as the per-scan kernels; grid = K workgroups, all resident (the
// launcher checks the occupancy: pte.hip, fused_scans_limit).
template <int NLU, int S8_BS, int WINDOW, int DBL_MODE>
__device__ __forceinline__ void slice8_scan_loop(EngineDev e, const SliceParams &sp, const ScanLoop &sl) {
    const int lane = lane_id();
    const int64_t cl = scan_loop_chain(e.K);               // XCD-aware: consecutive chains share an L2 (pte_kernels.hpp)
    int go = 1;
    if (lane == 0) go = scan_loop_gate(sl) ? 1 : 0;        // every workgroup of the launch is resident, or nobody starts (pte_kernels.hpp)
    if (!__builtin_amdgcn_readfirstlane(go)) return;
    for (int64_t i = 0; i < sl.n_scans; ++i) {
        e.trace_idx = sl.scan_idx0 + i;
        slice8_body<NLU, S8_BS, WINDOW, DBL_MODE>(e, sp, cl);
        __syncthreads();                                   // every lane's stores of the explore step happen before lane 0's release
        int slot = 0;
        if (lane == 0) slot = swap_handshake(e, sl, i, cl, e.slot_of_chain[cl]);
        slot = __builtin_amdgcn_readfirstlane(slot);
        __syncthreads();                                   // ... and lane 0's acquire before every lane's loads of the next one
        if (slot < 0) return;                              // (time-out: the error word is set, the host reports it)
    }
}
template <int NLU, int S8_BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTE_S8_WAVES, PTE_S8_WAVES))) void k_scans_slice8(EngineDev e, SliceParams sp, ScanLoop sl) {
    slice8_scan_loop<NLU, S8_BS, PTE_S7_WIN, PTE_S8_DBL_MODE>(e, sp, sl);
}
template <int NLU, int S8_BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTE_S8_WAVES, PTE_S8_WAVES))) void k_scans_slice8_generic(EngineDev e, SliceParams sp, ScanLoop sl) {
    slice8_scan_loop<NLU, S8_BS, PTE_S7_WIN, 0>(e, sp, sl);
}
}  // namespace pte
