// pte_slice6.hpp -- k_explore_slice6: slice5 + lane-parallel doubling and acceptance check.
//
// Bit-identical decisions, draws and states to k_explore_slice (v1) and the oracle, but no
// log-density evaluation on the sequential path at all.  For the scaled-precision MVN path
//
//   [ z < lp_fl(x with x_c = v) ],   z = fl(lp_fl(x) - E),   lp_fl(y) = fl(nhp * S_fl(y)),
//
// S_fl the fixed-tree sum of squares (non-negative terms => S_fl(y) = S(y)(1+theta), |theta| <=
// gamma_{NL+1}).  Dividing by nhp < 0 and cancelling the common part R = S(x) - x_c^2 gives
//
//   decision  <=>  v^2 (1+e1) < x_c^2 - E/nhp + e2,   |e1| <= 3e-15,  |e2| <= 3e-15 (S(x) + |E/nhp|),
//
// i.e. the threshold Q = x_c^2 - E/nhp does not involve the other coordinates; they enter only the
// error term.  With the margin m = 1e-12 (S~ + |Q|), S~ any estimate of S(x) within a factor 2, the
// test is:  v^2 < Q - m  -> inside;  v^2 > Q + m -> outside;  otherwise (probability ~1e-9 per test,
// and whenever something is not finite) the decision is taken EXACTLY: both fixed-tree roots are
// recomputed from the register-resident block and the block sums.  The exact tree root is also
// re-established at every block boundary and written out as the swap statistic.
//
// slice6: when an end point lies inside the slice (10-31 % of coordinates) the doubling steps are
// generated speculatively from the draws (they do not depend on densities), their end points are
// tested by the lanes in one shot and the stopping step is found with scalar bit logic; the following
// shrinkage batches run Neal's acceptance check of the doubling scheme per lane (each lane bisects
// for its own proposal with the O(1) predicate).  Anything unusual (sliver, degenerate bracket, more
// than J doublings) restores the coordinate's counters and takes the scalar procedure.
//
// Per coordinate the wave therefore runs: 3 draws read from the pre-converted buffer, the
// threshold, and batches of M shrinkage proposals evaluated by M lanes (as slice2/slice4).
#pragma once
#include "pte_slice5.hpp"

namespace pte {

template <int NLU, int M>
__global__ __launch_bounds__(64) void k_explore_slice6(EngineDev e, SliceParams sp) {
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
    int lm[M + 2];
#pragma unroll
    for (int n = 0; n < M + 2; ++n) lm[n] = (lane == n) ? -1 : 0;

    double BS = 0.0;                                   // lane b: exact fixed-tree sum of block b
    for (int b = 0; b < B; ++b) {
        int64_t i = 64 * (int64_t)b + lane;
        double v = (i < d) ? xrow[i] : 0.0;
        double s = wave_sum_dpp(v * v);
        if (lane == b) BS = s;
    }
    double S = upper_tree_root<NLU>(BS);               // exact root; refreshed at block boundaries
    if (nhp * S == -INFINITY) { if (lane == 0) set_error(e, ERR_SLICE_SUPPORT, (int)c, -1); return; }

    DrawBuf dr;
    dr.init(e.rng[2 * slot], e.rng[2 * slot + 1], lane, s_we, s_ke);
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
            PROF_T(tb0);
            double X = (lane < nl) ? xrow[base + lane] : 0.0;
            double Sest = S;                           // magnitude estimate for the margins inside this block
            PROF_T(tb1); PROF_ADD(7, tb1 - tb0);
            for (int l = 0; l < nl; ++l) {
                PROF_T(t0);
                const double xold = readlane_f64(X, l);
                dr.ensure(2 + M, lane, s_we, s_ke);
                const double E = dr.randexp_ensured(1 + M, lane, s_we, s_ke);
                const double u0 = readlane_f64(dr.unit, dr.p);
                dr.p += 1;
                const double Q = xold * xold - E * inv_nhp;
                const double mg = 1e-12 * (Sest + fabs(Q));
                const double Qlo = Q - mg, Qhi = Q + mg;
                const double L = xold - w * u0;
                const double R = L + w;
                const double thr = 1e-6 * fmax(fabs(L), fabs(R));

                // exact [z < lp(x with x_c = v)] from the fixed tree (sliver / non-finite cases only)
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

                double Lb = L, Rb = R;
                double cand = bitsel(lm[0], L, R);
                double xf = xold;
                bool done = false, first = true;
                int n_base = 0;
                uint64_t ins0 = 0;
                PROF_T(t1); PROF_ADD(0, t1 - t0); PROF_ADD(4, 1);
                while (true) {
                    PROF_ADD(6, 1);
                    const double Lb0 = Lb, Rb0 = Rb;
                    double u[M];
#pragma unroll
                    for (int n = 0; n < M; ++n) u[n] = readlane_f64(dr.unit, dr.p + n);
#pragma unroll
                    for (int n = 0; n < M; ++n) {
                        const double v = Lb + u[n] * (Rb - Lb);
                        cand = bitsel(lm[n + 2], v, cand);
                        const int below = neg_mask(v - xold);
                        Lb = bitsel(below, v, Lb);
                        Rb = bitsel(below, Rb, v);
                    }
                    const double q = cand * cand;
                    const uint64_t ins = ballot64(q < Qlo);
                    const uint64_t outs = ballot64(q > Qhi);
                    const uint64_t live = first ? ((1ull << (M + 2)) - 1ull) : (((1ull << M) - 1ull) << 2);
                    const bool amb = (~(ins | outs) & live) != 0ull;
                    const bool risk = ballot64(!((Rb - Lb) > thr)) != 0ull;
                    if (first) {
                        if (__builtin_expect((ins & 3ull) != 0ull || risk || amb, 0)) {
                            ins0 = (amb || risk) ? 0ull : (ins & 3ull);
#ifdef PTE_DEBUG_V6
                            if (lane == 0 && cl == 1 && b == 0) printf("break l=%d ins=%llx outs=%llx amb=%d risk=%d ins0=%llu\n", l, (unsigned long long)ins, (unsigned long long)outs, (int)amb, (int)risk, (unsigned long long)ins0);
#endif
                            break;
                        }
                        steps_n += 1;
                    } else if (__builtin_expect(risk || amb, 0)) {
                        Lb = Lb0; Rb = Rb0;
                        break;
                    }
                    const uint64_t acc = (ins >> 2) & ((1ull << M) - 1ull);
                    if (acc != 0ull) {
                        const int n = (int)__builtin_ctzll(acc);
                        xf = readlane_f64(cand, n + 2);
                        dr.p += n + 1;
                        steps_sum += n_base + n + 1; steps_n += 1;
                        acc_sum += 1; acc_n += 1;
                        done = true;
                        break;
                    }
                    dr.p += M;
                    n_base += M;
                    first = false;
                    if (__builtin_expect(n_base + M > sp.max_iter, 0)) break;
                    dr.ensure(M, lane, s_we, s_ke);
                }
                if (!done && first && ins0 != 0ull) {
                    // ---- lane-parallel doubling (slice_double, SliceSampler.jl:109-121) + shrinkage with slice_accept
                    constexpr int J = 4;
                    const long long steps_sum0 = steps_sum; const int steps_n0 = steps_n, acc_sum0 = acc_sum, acc_n0 = acc_n;
                    bool bail = false;
                    dr.ensure(J + M, lane, s_we, s_ke);
                    const int p_dbl = dr.p;                  // == logical position after u0 (ensure keeps the stream position)
                    const uint64_t seed_dbl = dr.seed;
                    double Lc = L, Rc = R, cd = 0.0;
                    unsigned sidemask = 0;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const double V = readlane_f64(dr.unit, p_dbl + j);
                        const bool sideL = V <= 0.5;
                        const double W = Rc - Lc;
                        const double Ln = Lc - W, Rn = Rc + W;
                        const double en = sideL ? Ln : Rn;
                        Lc = sideL ? Ln : Lc;
                        Rc = sideL ? Rc : Rn;
                        cd = (lane == j) ? en : cd;
                        sidemask |= (sideL ? 1u : 0u) << j;
                    }
                    const double qd = cd * cd;
                    const uint64_t insD = ballot64(qd < Qlo), outD = ballot64(qd > Qhi);
                    if ((~(insD | outD) & ((1ull << J) - 1ull)) != 0ull) bail = true;
                    int jstar = 0;
                    {
                        bool inL = (ins0 & 1ull) != 0ull, inR = (ins0 & 2ull) != 0ull;
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            if (jstar == 0) {
                                const bool inj = ((insD >> j) & 1ull) != 0ull;
                                if ((sidemask >> j) & 1u) inL = inj; else inR = inj;
                                if (!(inL || inR)) jstar = j + 1;
                            }
                        }
                    }
                    if (jstar == 0) bail = true;             // more than J doublings: scalar procedure
                    double LL = L, RR = R;
                    if (!bail) {
                        const unsigned below = (1u << jstar) - 1u;
                        const unsigned mL = sidemask & below, mR = ~sidemask & below;
                        if (mL) LL = readlane_f64(cd, 31 - __builtin_clz(mL));
                        if (mR) RR = readlane_f64(cd, 31 - __builtin_clz(mR));
                        dr.p = p_dbl + jstar;
                        steps_sum += jstar; steps_n += 1;     // explorer_n_steps += p - K
                        // shrinkage batches; every lane runs slice_accept (:192-237) for its own proposal
                        const double thr3 = 1e-6 * fmax(fabs(LL), fabs(RR));
                        Lb = LL; Rb = RR; n_base = 0;
                        while (!bail && !done) {
                            dr.ensure(M, lane, s_we, s_ke);
                            double u[M];
#pragma unroll
                            for (int n = 0; n < M; ++n) u[n] = readlane_f64(dr.unit, dr.p + n);
#pragma unroll
                            for (int n = 0; n < M; ++n) {
                                const double v = Lb + u[n] * (Rb - Lb);
                                cand = bitsel(lm[n + 2], v, cand);
                                const int below2 = neg_mask(v - xold);
                                Lb = bitsel(below2, v, Lb);
                                Rb = bitsel(below2, Rb, v);
                            }
                            const double q3 = cand * cand;
                            const bool my = lane >= 2 && lane < M + 2;
                            const bool in3 = my && q3 < Qlo, out3 = q3 > Qhi;
                            bool ambl = my && !(in3 || out3);
                            bool ok = true;
                            if (in3) {
                                double Lhat = LL, Rhat = RR;
                                bool oL = true, oR = true, Rst = false, Lst = false, D = false;   // both end points are outside after doubling
                                while (Rhat - Lhat > w11) {
                                    const double Mid = (Lhat + Rhat) / 2.0;
                                    if ((xold < Mid && cand >= Mid) || (xold >= Mid && cand < Mid)) D = true;
                                    if (cand < Mid) { Rhat = Mid; Rst = true; } else { Lhat = Mid; Lst = true; }
                                    if (D) {
                                        if (Lst) { const double qq = Lhat * Lhat; const bool i2 = qq < Qlo; if (!(i2 || qq > Qhi)) ambl = true; oL = !i2; Lst = false; }
                                        if (Rst) { const double qq = Rhat * Rhat; const bool i2 = qq < Qlo; if (!(i2 || qq > Qhi)) ambl = true; oR = !i2; Rst = false; }
                                        if (oL && oR) { ok = false; break; }
                                    }
                                }
                            }
                            const bool risk3 = ballot64(!((Rb - Lb) > thr3)) != 0ull;
                            if (ballot64(ambl) != 0ull || risk3) { bail = true; break; }
                            const uint64_t insm = (ballot64(in3) >> 2) & ((1ull << M) - 1ull);
                            const uint64_t passm = (ballot64(in3 && ok) >> 2) & ((1ull << M) - 1ull);
                            if (passm != 0ull) {
                                const int n = (int)__builtin_ctzll(passm);
                                xf = readlane_f64(cand, n + 2);
                                dr.p += n + 1;
                                steps_sum += n_base + n + 1; steps_n += 1;
                                acc_n += __builtin_popcountll(insm & ((2ull << n) - 1ull));   // one slice_accept call per proposal inside the slice
                                acc_sum += 1;
                                done = true;
                            } else {
                                acc_n += __builtin_popcountll(insm);
                                dr.p += M;
                                n_base += M;
                                if (n_base + M > sp.max_iter) bail = true;
                            }
                        }
                    }
                    if (bail) {       // restore and let the scalar procedure redo this coordinate from the u0 draw
                        done = false;
                        if (dr.seed != seed_dbl) { dr.seed = seed_dbl; dr.fill(lane, s_we, s_ke); }   // a refill happened in between
                        dr.p = p_dbl;
                        steps_sum = steps_sum0; steps_n = steps_n0; acc_sum = acc_sum0; acc_n = acc_n0;
                        Lb = L; Rb = R; n_base = 0; first = true; xf = xold;
                    }
                }
                PROF_T(t2); PROF_ADD(1, t2 - t1);
                if (__builtin_expect(!done, 0)) {
                    PROF_ADD(5, 1);
                    // ---- scalar procedure of the reference (SliceSampler.jl:97-237) with O(1) predicates
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
                    for (int n = n0; n <= sp.max_iter; ++n) {
                        const double W = Rbar - Lbar;
                        if (__builtin_expect(n > 1 && !(W > thr2), 0)) {
                            if (jl_isapprox(Lbar, Rbar)) { steps_sum += (n - 1); steps_n += 1; fin = true; break; }
                        }
                        const double newpos = Lbar + dr.rand(lane, s_we, s_ke) * W;
                        if (inside(newpos)) {
                            bool ok = true;
                            if (doubled) {
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
                Sest = Sest + fabs(xf * xf - xold * xold);      // magnitude only (never decreases inside a block)
                if (lane == l) X = xf;
                PROF_T(t4); PROF_ADD(3, t4 - t3);
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
