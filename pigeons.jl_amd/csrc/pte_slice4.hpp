// pte_slice4.hpp -- k_explore_slice4: SliceSampler kernel tuned to the measured gfx950 cost model
// (DESIGN.md 5): a lone wavefront pays ~4 cycles per VALU instruction but 20-40 cycles for every
// VALU->SALU hand-off (v_cmp -> branch / v_cndmask, v_readlane -> use).  Same algorithm, draws and
// results as k_explore_slice / slice2 / slice3 (tests compare all of them bit for bit).
//
//  * selects on data-dependent conditions are done in the VALU data path: the sign bit of a
//    difference is smeared into a mask (v_ashrrev) and merged with v_bfi -- no VCC round trip;
//  * candidates are evaluated M at a time by the lanes (as slice2), and a coordinate whose first M
//    proposals are all rejected simply runs another batch (no switch to scalar code);
//  * the scalar fallback (doubling needed) uses the filtered predicates of slice3, so it pays the
//    log2(P)-add tree path only for the accepted point.
#pragma once
#include "pte_slice3.hpp"

namespace pte {

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

template <int NLU, int M>
__global__ __launch_bounds__(64) void k_explore_slice4(EngineDev e, SliceParams sp) {
    constexpr int NL = 6 + NLU;
    __shared__ double s_we[256];
    __shared__ unsigned long long s_ke[256];
    const int lane = lane_id();
    for (int i = lane; i < 256; i += 64) { s_we[i] = ZIG_WE[i]; s_ke[i] = ZIG_KE[i]; }
    __syncthreads();
    const int64_t cl = blockIdx.x;
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    if (c == 0 && e.N > 1) {
        iid_refresh_recorded<NLU>(e, cl, c, slot, e.sd[0], lane);
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
    // lane masks of the candidate slots: lane 0 <- L, lane 1 <- R (first batch), lanes 2..M+1 <- proposals
    int lm[M + 2];
#pragma unroll
    for (int n = 0; n < M + 2; ++n) lm[n] = (lane == n) ? -1 : 0;

    double BS = 0.0;
    for (int b = 0; b < B; ++b) {
        int64_t i = 64 * (int64_t)b + lane;
        double v = (i < d) ? xrow[i] : 0.0;
        double s = wave_tree_sum64(v * v);
        if (lane == b) BS = s;
    }
    double S = upper_tree_root<NLU>(BS);
    double lp = nhp * S;
    if (lp == -INFINITY) { if (lane == 0) set_error(e, ERR_SLICE_SUPPORT, (int)c, -1); return; }

    DrawBuf dr;
    dr.init(e.rng[2 * slot], e.rng[2 * slot + 1], lane, s_we, s_ke);
    long long steps_sum = 0;
    int steps_n = 0, acc_sum = 0, acc_n = 0;
    int err = 0, err_coord = -1;
    double sib[NL];
    double z = 0.0, Qlo = 0.0, Qhi = 0.0;
#ifdef PTE_PROFILE_SECTIONS
    // cycles: [0] head (draws, L/R) [1] batches [2] scalar fallback [3] tail; counts: [4] coords, [5] fallback coords, [6] batches, [7] block setup cycles
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_T(v) __builtin_amdgcn_sched_barrier(0); const long long v = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define PROF_ADD(i, x) prof[i] += (x)
#else
#define PROF_T(v)
#define PROF_ADD(i, x)
#endif

    auto evalS = [&](double v) __attribute__((always_inline)) -> double {
        double t = v * v;
#pragma unroll
        for (int k = 0; k < NL; ++k) t = t + sib[k];
        return t;
    };
    auto inside = [&](double v) __attribute__((always_inline)) -> bool {       // filtered predicate of slice3 (scalar fallback only)
        const double q = v * v;
        const bool in = q < Qlo;
        const bool out = q > Qhi;
        if (__builtin_expect(!(in || out), 0)) return z < nhp * evalS(v);
        return in;
    };

    for (int pass = 0; pass < sp.n_passes && !err; ++pass) {
        for (int b = 0; b < B && !err; ++b) {
            const int64_t base = 64 * (int64_t)b;
            const int nl = (int)min((int64_t)64, d - base);
            PROF_T(tb0);
            double X = (lane < nl) ? xrow[base + lane] : 0.0;
            double U[7];
            butterfly6(X * X, U);
            {
                double V = BS;
#pragma unroll
                for (int q = 0; q < NLU; ++q) {
                    sib[6 + q] = readlane_f64(V, b ^ (1 << q));
                    V = V + shfl_xor_f64(V, 1 << q);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) sib[k] = readlane_f64(U[k], 1 << k);
            double xf = 0.0;
            PROF_T(tb1); PROF_ADD(7, tb1 - tb0);
            for (int l = 0; l < nl; ++l) {
                PROF_T(t0);
                const double xold = readlane_f64(X, l);
                dr.ensure(2 + M, lane, s_we, s_ke);
                const double E = dr.randexp(lane, s_we, s_ke);
                dr.ensure(1 + M, lane, s_we, s_ke);
                z = lp - E;
                const double u0 = readlane_f64(dr.unit, dr.p);
                dr.p += 1;
                const double L = xold - w * u0;
                const double R = L + w;
                const double thr = 1e-6 * fmax(fabs(L), fabs(R));   // isapprox pre-filter, valid for every nested bracket
                double Lb = L, Rb = R;
                double cand = bitsel(lm[0], L, R);
                bool done = false;
                int n_base = 0;                      // proposals consumed by earlier batches of this coordinate
                bool first = true;
                PROF_T(t1); PROF_ADD(0, t1 - t0); PROF_ADD(4, 1);
                while (true) {
                    PROF_ADD(6, 1);
                    // ---- one speculative batch: M proposals from the current bracket, VALU only
                    const double Lb0 = Lb, Rb0 = Rb;
                    double u[M];
#pragma unroll
                    for (int n = 0; n < M; ++n) u[n] = readlane_f64(dr.unit, dr.p + n);
#pragma unroll
                    for (int n = 0; n < M; ++n) {
                        const double v = Lb + u[n] * (Rb - Lb);
                        cand = bitsel(lm[n + 2], v, cand);
                        const int below = neg_mask(v - xold);          // v < xold
                        Lb = bitsel(below, v, Lb);
                        Rb = bitsel(below, Rb, v);
                    }
                    const double Sc = evalS(cand);
                    const double lpc = nhp * Sc;
                    uint64_t ins = ballot64(z < lpc);
                    // (inline-asm results count as divergent for the compiler: re-uniformise what steers control flow)
                    const bool risk = ballot64(!((Rb - Lb) > thr)) != 0ull;
                    if (first) {
                        if (__builtin_expect((ins & 3ull) != 0ull || risk, 0)) break;      // doubling (or degenerate bracket)
                        steps_n += 1;                                                      // explorer_n_steps += p - K = 0
                    } else if (__builtin_expect(risk, 0)) {
                        Lb = Lb0; Rb = Rb0;          // the scalar code redoes this batch with exact isapprox tests
                        break;
                    }
                    const uint64_t acc = (ins >> 2) & ((1ull << M) - 1ull);
                    if (acc != 0ull) {
                        const int n = (int)__builtin_ctzll(acc);
                        xf = readlane_f64(cand, n + 2);
                        S = readlane_f64(Sc, n + 2);
                        lp = nhp * S;
                        dr.p += n + 1;
                        steps_sum += n_base + n + 1; steps_n += 1;
                        acc_sum += 1; acc_n += 1;          // slice_accept: no doubling => accept
                        done = true;
                        break;
                    }
                    // all M rejected: continue the shrinkage with the next batch
                    dr.p += M;
                    n_base += M;
                    first = false;
                    if (__builtin_expect(n_base + M > sp.max_iter, 0)) break;
                    dr.ensure(M, lane, s_we, s_ke);
                }
                PROF_T(t2); PROF_ADD(1, t2 - t1);
                if (__builtin_expect(!done, 0)) {
                    PROF_ADD(5, 1);
                    // ---- scalar procedure of the reference, restarted after the u0 draw of this coordinate
                    //      (first batch) or continued from the current bracket (later batches)
                    {
                        const double T = z * inv_nhp;
                        const double Q = T - (S - xold * xold);
                        const double m = 1e-11 * (fabs(T) + S);
                        Qlo = Q - m; Qhi = Q + m;
                    }
                    double LL = L, RR = R;
                    bool in_L = false, in_R = false;
                    int n0 = n_base + 1;
                    double Lbar = readlane_f64(Lb, 0), Rbar = readlane_f64(Rb, 0);
                    if (first) {
                        in_L = inside(LL); in_R = inside(RR);
                        int K = sp.p;
                        while (K > 0 && (in_L || in_R)) {
                            const double V = dr.rand(lane, s_we, s_ke);
                            if (V <= 0.5) { LL = LL - (RR - LL); in_L = inside(LL); }
                            else { RR = RR + (RR - LL); in_R = inside(RR); }
                            K -= 1;
                        }
                        steps_sum += (sp.p - K); steps_n += 1;
                        Lbar = LL; Rbar = RR; n0 = 1;
                    }
                    const bool doubled = (RR - LL) > w11;
                    const double thr2 = 1e-6 * fmax(fabs(LL), fabs(RR));
                    bool fin = false;
                    xf = xold;
                    for (int n = n0; n <= sp.max_iter; ++n) {
                        const double W = Rbar - Lbar;
                        if (__builtin_expect(n > 1 && !(W > thr2), 0)) {
                            if (jl_isapprox(Lbar, Rbar)) { steps_sum += (n - 1); steps_n += 1; fin = true; break; }
                        }
                        const double newpos = Lbar + dr.rand(lane, s_we, s_ke) * W;
                        if (inside(newpos)) {
                            bool ok = true;
                            if (doubled) {          // slice_accept (:192-237)
                                double Lhat = LL, Rhat = RR;
                                bool oL = !in_L, oR = !in_R;
                                bool Rstale = false, Lstale = false, D = false;
                                while (Rhat - Lhat > w11) {
                                    const double Mid = (Lhat + Rhat) / 2.0;
                                    if ((xold < Mid && newpos >= Mid) || (xold >= Mid && newpos < Mid)) D = true;
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
                            if (ok) {
                                acc_sum += 1;
                                xf = newpos;
                                S = evalS(newpos);
                                lp = nhp * S;
                                steps_sum += n; steps_n += 1;
                                fin = true;
                                break;
                            }
                        }
                        if (newpos < xold) Lbar = newpos; else Rbar = newpos;
                        if (__builtin_expect(n == sp.max_iter, 0)) {
                            if (jl_isapprox(Lbar, Rbar)) { steps_sum += n; steps_n += 1; fin = true; }
                        }
                    }
                    if (!fin) { err = ERR_SLICE_MAX_ITER; err_coord = (int)(base + l); break; }
                }
                PROF_T(t3); PROF_ADD(2, t3 - t2);
                if (__builtin_expect(!isfinite(lp), 0)) { err = ERR_SLICE_INVALID_LP; err_coord = (int)(base + l); break; }
                if (lane == l) X = xf;
                // ---- siblings of coordinate l+1
                if (l + 1 < nl) {
                    const int l1 = l + 1;
                    const int r = __builtin_ctz((unsigned)l1);
                    double t = xf * xf;
                    if (r == 0) {
                        sib[0] = t;
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            if (k < r) { t = t + sib[k]; sib[k] = readlane_f64(U[k], l1 ^ (1 << k)); }
                            else if (k == r) sib[k] = t;
                        }
                    }
                }
                PROF_T(t4); PROF_ADD(3, t4 - t3);
            }
            if (err) break;
            if (lane < nl) xrow[base + lane] = X;
            {
                double t = xf * xf;
#pragma unroll
                for (int k = 0; k < 6; ++k) t = t + sib[k];
                if (lane == b) BS = t;
            }
        }
    }
    if (err) { if (lane == 0) set_error(e, err, (int)c, err_coord); return; }
    if (lane == 0) {
        e.suff[slot] = S;
        e.rng[2 * slot] = dr.final_seed();
        e.expl_steps_sum[cl] += (double)steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += (double)acc_sum;     e.expl_acc_n[cl] += acc_n;
#ifdef PTE_PROFILE_SECTIONS
        for (int i = 0; i < 8; ++i) e.on_m2[8 * cl + i] += (double)prof[i];   // debug builds only (needs d >= 8K)
#endif
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

}  // namespace pte
