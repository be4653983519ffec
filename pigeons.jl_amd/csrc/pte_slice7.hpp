// pte_slice7.hpp -- k_explore_slice7: SliceSampler kernel, speculation over stream offsets.
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
// (SliceSampler.jl:97-237: doubling, shrinkage, acceptance check of the doubling scheme) on its own
// hypothesis, reading pre-converted draws from a 512-draw LDS window of the stream.  The chase then
// walks g = 0,1,.. through the lanes that turned out to be true and applies their results.  A
// hypothesis that meets anything inexact (ambiguous filter outcome, ziggurat slow path, window
// overflow, too many iterations) is marked invalid; if the chase hits it the coordinate is done by
// the exact sequential procedure (fixed-tree recompute available), which is also how errors are
// raised.  So a round retires ~4.5 coordinates for the latency of the slowest of 64 scalar updates.
#pragma once
#include "pte_slice5.hpp"

namespace pte {

namespace s7 {
#ifndef PTE_S7_G
#define PTE_S7_G 5
#endif
constexpr int G = PTE_S7_G;              // coordinates (levels) per round
#ifndef PTE_S7_WIN
#define PTE_S7_WIN 512
#endif
constexpr int WIN = PTE_S7_WIN;          // draws in the LDS window (a multiple of 64)
#ifndef PTE_S7_MARGIN
#define PTE_S7_MARGIN 80
#endif
constexpr int REFILL_AT = WIN - PTE_S7_MARGIN;   // hypotheses start <= p + 29 and read <= 46 draws (lane 0: <= 62): 76 would do
constexpr int CAP_ITERS = 24;            // speculative shrinkage cap (beyond: exact path)
#ifndef PTE_S7_TABLES                     // (a tuning build can substitute its own window layout: LO, WD, BASE initialisers)
#define PTE_S7_TABLES {0, 3, 6, 10, 14}, {1, 14, 16, 17, 16}, {0, 1, 15, 31, 48}
#endif
struct S7Tables { int lo[G], wd[G], base[G]; };
constexpr S7Tables S7T = {PTE_S7_TABLES};
struct S7Row { int v[G]; };
constexpr S7Row s7_row(const int (&a)[G]) { S7Row r{}; for (int i = 0; i < G; ++i) r.v[i] = a[i]; return r; }
__device__ constexpr S7Row LO_ = s7_row(S7T.lo), WD_ = s7_row(S7T.wd), BASE_ = s7_row(S7T.base);
#define LO LO_.v
#define WD WD_.v
#define BASE BASE_.v
// level and offset (relative to the round's start) of the hypothesis of a lane
__device__ __forceinline__ int s7_level(int lane) { int g = 0; for (int k = 1; k < G; ++k) g += (lane >= BASE[k]) ? 1 : 0; return g; }
__device__ __forceinline__ int s7_pick(const int (&a)[G], int g, int dflt) { int r = dflt; for (int k = 0; k < G; ++k) r = (g == k) ? a[k] : r; return r; }
constexpr int VALID = 1 << 30;
}  // namespace s7

// speculation budgets: a hypothesis other than the certain one (lane 0) that needs more doubling /
// shrinkage / halving steps than this is dropped (its coordinate becomes lane 0 of the next round)
struct S7Tune { int bud_d, bud_s, bud_a; };
namespace s7 {
}  // namespace s7

template <int NLU>
__global__ __launch_bounds__(64) void k_explore_slice7(EngineDev e, SliceParams sp, S7Tune tn) {
    using namespace s7;
    __shared__ double s_we[256];
    __shared__ unsigned long long s_ke[256];
    __shared__ double s_u[WIN];
    __shared__ double s_e[WIN];              // randexp fast-path value; NaN <=> slow path needed
    const int lane = lane_id();
    for (int i = lane; i < 256; i += 64) { s_we[i] = ZIG_WE[i]; s_ke[i] = ZIG_KE[i]; }
    __syncthreads();
    const int64_t cl = blockIdx.x;
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
    const double w = sp.w;
    const double w11 = 1.1 * sp.w;
    const int cap_iters = min(sp.max_iter, CAP_ITERS);
    const int kcap = min(sp.p, 20);                    // window headroom: 2 + 20 + 24 draws per hypothesis

    // hypothesis (g, rel) of this lane
    const int hg = s7_level(lane);
    const int hrel = s7_pick(LO, hg, 0) + lane - s7_pick(BASE, hg, 0);

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
    uint64_t wseed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];
    int p = 0;                                         // uniform: next unread draw of the window
    uint64_t gamma_inv = gamma;                        // gamma^-1 mod 2^64 (gamma is odd): Newton, 5 steps
    for (int k = 0; k < 5; ++k) gamma_inv *= 2ull - gamma * gamma_inv;
    auto fill_window = [&]() __attribute__((always_inline)) {
        __syncthreads();                               // one wave per block: orders the LDS accesses
#pragma unroll
        for (int k = 0; k < WIN / 64; ++k) {
            const int i = 64 * k + lane;
            const uint64_t r = mix64(wseed + (uint64_t)(i + 1) * gamma);
            const uint64_t ri = r & MASK52;
            const int idx = (int)(ri & 0xFF);
            s_u[i] = u52_to_unit(r);
            s_e[i] = (ri < s_ke[idx]) ? (double)ri * s_we[idx] : __longlong_as_double(0x7ff8000000000000LL);
        }
        p = 0;
        __syncthreads();
    };
    fill_window();

    long long steps_sum = 0;
    int steps_n = 0, acc_sum = 0, acc_n = 0;
    int err = 0, err_coord = -1;
#ifdef PTE_PROFILE_SECTIONS
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int pass = 0; pass < sp.n_passes && !err; ++pass) {
        for (int b = 0; b < B && !err; ++b) {
            const int64_t base = 64 * (int64_t)b;
            const int nl = (int)min((int64_t)64, d - base);
            double X = (lane < nl) ? xrow[base + lane] : 0.0;
            double Sest = S;
            int l = 0;
            while (l < nl) {
                PROF_T(t0);
                if (p > REFILL_AT) { wseed += (uint64_t)p * gamma; fill_window(); }
                // ================= speculative round: lane = hypothesis (l + hg, p + hrel) ==========
                // Decisions are sign tests of d(v) = v^2 - Q; every tested |d| is folded into dmin and the
                // hypothesis is valid only if dmin clears the margin at the end (so the loops carry no
                // validity state).  NaNs never pass: they fail the final interval-width test.
                // ---- head: slice level and initial interval (SliceSampler.jl:97-113)
                const bool active = (l + hg) < nl;
                const double xold = __shfl(X, (l + hg) & 63, 64);
                const int idx0 = p + hrel;
                const double E = s_e[idx0];
                const double u0 = s_u[idx0 + 1];
                const double *up = &s_u[idx0 + 2];           // next unread draw of this hypothesis
                double Vn = *up;                             // software-pipelined: always one load in flight
                const double Q = xold * xold - E * inv_nhp;
                const double Bq = Sest + fabs(Q);
                double dmin = INFINITY;
                auto test = [&](double v) __attribute__((always_inline)) -> double {
                    const double d = v * v - Q;
                    dmin = fmin(dmin, fabs(d));
                    return d;                                // inside the slice <=> d < 0
                };
                double LL = xold - w * u0;
                double RR = LL + w;
                double dL = test(LL), dR = test(RR);
                // ---- doubling (:115-139), branch-free body, per-lane trip count
                int kd = 0;
                const int kbud = (lane == 0) ? kcap : min(kcap, tn.bud_d);
                const int nmax = (lane == 0) ? cap_iters : min(cap_iters, tn.bud_s);
                bool need = active && (fmin(dL, dR) < 0.0) && kbud > 0;
                while (need) {
                    const double V = Vn;
                    up += 1;
                    Vn = *up;
                    const bool left = V <= 0.5;
                    const double wd = RR - LL;
                    const double cand = left ? (LL - wd) : (RR + wd);
                    const double dc = test(cand);
                    LL = left ? cand : LL;
                    RR = left ? RR : cand;
                    dL = left ? dc : dL;
                    dR = left ? dR : dc;
                    kd += 1;
                    need = (kd < kbud) && (fmin(dL, dR) < 0.0);
                }
                const bool dbl_ok = !((kd < sp.p) && (fmin(dL, dR) < 0.0));     // ended by itself, not by a budget
                const bool doubled = (RR - LL) > w11;
                const double thr2 = 1e-6 * fmax(fabs(LL), fabs(RR));
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(LL), "v"(RR), "v"(kd), "v"(thr2), "v"(Vn));
#endif
                PROF_T(t1); PROF_ADD(0, t1 - t0);
                // ---- shrinkage (:141-190) up to the first proposal inside the slice
                double Lbar = LL, Rbar = RR, xf = xold, dx = 1.0, W = 0.0;
                int n = 0;
                bool go = active && dbl_ok;
                while (go) {
                    W = Rbar - Lbar;
                    xf = Lbar + Vn * W;
                    up += 1;
                    Vn = *up;
                    n += 1;
                    dx = test(xf);
                    const bool below = xf < xold;
                    Lbar = below ? xf : Lbar;
                    Rbar = below ? Rbar : xf;
                    go = !(dx < 0.0) && n < nmax;
                }
                // (opaque copy: the compiler otherwise reuses the in-loop compare, which only holds the lanes
                //  of the LAST wave iteration -- lanes that left the loop earlier would read 0)
                asm volatile("" : "+v"(dx));
                // W > thr2 at the last step (widths only shrink) rules out isapprox(Lbar, Rbar) at every step
                bool valid = active && !(E != E) && dbl_ok && (dx < 0.0) && (W > thr2);
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(xf), "v"(n), "v"(dmin));
#endif
                PROF_T(t2); PROF_ADD(1, t2 - t1);
                // ---- acceptance check of the doubling scheme (:192-237) for the proposal found; a
                //      proposal that fails it sends the hypothesis to the exact path
                if (valid && doubled) {
                    double Lhat = LL, Rhat = RR, oL = dL, oR = dR;
                    bool D = false, ok = true;
                    int na = (lane == 0) ? 64 : tn.bud_a;
                    while (ok && (Rhat - Lhat > w11)) {
                        const double Mid = (Lhat + Rhat) * 0.5;
                        const bool right = xf < Mid;
                        D = D || ((xold < Mid) != right);
                        const double dm = test(Mid);
                        Rhat = right ? Mid : Rhat;
                        Lhat = right ? Lhat : Mid;
                        oR = right ? dm : oR;
                        oL = right ? oL : dm;
                        ok = !(D && !(oL < 0.0) && !(oR < 0.0)) && (na > 0);
                        na -= 1;
                    }
                    valid = ok;
                }
#ifdef PTE_DEBUG_S7
                {
                    const bool c1 = !(E != E), c2 = dbl_ok, c3 = dx < 0.0, c4 = W > thr2, c5 = valid, c6 = dmin > 2e-12 * Bq;
                    prof[0] += (ballot64(!c1) & 1) ; prof[1] += (ballot64(!c2) & 1); prof[2] += (ballot64(!c3) & 1);
                    prof[5] += (ballot64(!c4) & 1); prof[6] += (ballot64(!c5) & 1); prof[7] += (ballot64(!c6) & 1);
                }
#endif
                valid = valid && (dmin > 2e-12 * Bq);
                const int packed = (valid ? VALID : 0) | (2 + kd + n) | ((kd + n) << 8);
                const double dS = fabs(xf * xf - xold * xold);
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(packed), "v"(dS));
#endif
                PROF_T(t3); PROF_ADD(2, t3 - t2); PROF_ADD(3, 1);
                // ================= chase the true path through the hypotheses =======================
                int gdone = 0;
                {
                    const double X0 = X, Sest0 = Sest;
                    int o = 0, st = 0;
                    uint64_t tmask = 0;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (l + g >= nl) break;
                        const int k = o - LO[g];
                        if ((unsigned)k >= (unsigned)WD[g]) break;
                        const int ln = BASE[g] + k;
                        const int pk = __builtin_amdgcn_readlane(packed, ln);
                        if (!(pk & VALID)) break;
                        tmask |= 1ull << ln;
                        Sest = Sest + readlane_f64(dS, ln);
                        X = writelane_f64(X, readlane_f64(xf, ln), l + g);
                        o += pk & 0xFF;
                        st += (pk >> 8) & 0xFFF;
                        gdone += 1;
                    }
                    // every applied hypothesis' margin (computed from the round's Sest) must still cover S
                    if (__builtin_expect((ballot64(!(Sest <= 99.0 * Bq)) & tmask) != 0ull, 0)) {
                        const int pk = __builtin_amdgcn_readlane(packed, 0);        // keep the certain one only
                        X = writelane_f64(X0, readlane_f64(xf, 0), l);
                        Sest = Sest0 + readlane_f64(dS, 0);
                        o = pk & 0xFF; st = (pk >> 8) & 0xFFF; gdone = 1;
                    }
                    steps_sum += st; steps_n += 2 * gdone; acc_n += gdone; acc_sum += gdone;
                    p += o;
                    l += gdone;
                }
#ifdef PTE_PROFILE_SECTIONS
                asm volatile("" :: "v"(X), "s"(p), "s"(l));
#endif
                PROF_T(t4); PROF_ADD(5, t4 - t3); PROF_ADD(4, gdone);
                if (__builtin_expect(gdone == 0, 0)) {
                    // ================= exact sequential procedure for coordinate l ===================
                    SeqRng rs{wseed + (uint64_t)p * gamma, gamma};
                    const double xo = readlane_f64(X, l);
                    const double E = randexp_seq(rs);
                    const double u0 = rs.rand();
                    const double Q = xo * xo - E * inv_nhp;
                    const double mg = 1e-12 * (Sest + fabs(Q));
                    const double Qlo = Q - mg, Qhi = Q + mg;
                    auto inside_exact = [&](double v) __attribute__((always_inline)) -> bool {
                        const double Xv = (lane == l) ? v : X;
                        const double sv = wave_sum_dpp(Xv * Xv), s0 = wave_sum_dpp(X * X);
                        const double Sv = upper_tree_root<NLU>((lane == b) ? sv : BS);
                        const double S0 = upper_tree_root<NLU>((lane == b) ? s0 : BS);
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
                    if (!fin) { err = ERR_SLICE_MAX_ITER; err_coord = (int)(base + l); break; }
                    Sest = Sest + fabs(xn * xn - xo * xo);
                    if (lane == l) X = xn;
                    l += 1;
                    p += (int)((rs.seed - (wseed + (uint64_t)p * gamma)) * gamma_inv);   // draws consumed; window stays
                    PROF_T(t5); PROF_ADD(7, t5 - t4); PROF_ADD(6, 1);
                }
            }
            if (err) break;
            if (lane < nl) xrow[base + lane] = X;
            {   // re-establish the exact fixed-tree values at the block boundary
                const double s = wave_sum_dpp(X * X);
                if (lane == b) BS = s;
                S = upper_tree_root<NLU>(BS);
                if (__builtin_expect(!isfinite(nhp * S), 0)) { err = ERR_SLICE_INVALID_LP; err_coord = (int)base; }
            }
        }
    }
    if (err) { if (lane == 0) set_error(e, err, (int)c, err_coord); return; }
    if (lane == 0) {
        e.suff[slot] = S;
        e.rng[2 * slot] = wseed + (uint64_t)p * gamma;
        e.expl_steps_sum[cl] += (double)steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += (double)acc_sum;     e.expl_acc_n[cl] += acc_n;
#ifdef PTE_PROFILE_SECTIONS
        for (int i = 0; i < 8; ++i) e.on_m2[8 * cl + i] += (double)prof[i];   // debug builds only (needs d >= 8K)
#endif
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

}  // namespace pte
