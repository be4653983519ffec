// pte_slice5.hpp -- k_explore_slice5: SliceSampler kernel with tree-free filtered predicates.
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
// Per coordinate the wave therefore runs: 3 draws read from the pre-converted buffer, the
// threshold, and batches of M shrinkage proposals evaluated by M lanes (as slice2).
#pragma once
#include "pte_slice_common.hpp"

namespace pte {

template <int NLU, int M>
__global__ __launch_bounds__(64) void k_explore_slice5(EngineDev e, SliceParams sp) {
    __shared__ double s_we[256];
    __shared__ unsigned long long s_ke[256];
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
    long long prof2 = 0;
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
                PROF_T(t1); PROF_ADD(0, t1 - t0); PROF_ADD(4, 1);
                while (true) {
                    PROF_ADD(6, 1);
                    PROF_T(s0);
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
#ifdef PTE_PROFILE_SECTIONS
                    asm volatile("" :: "v"(cand), "v"(Lb), "v"(Rb));
#endif
                    PROF_T(s1); PROF_ADD(7, s1 - s0);
                    const double q = cand * cand;
                    const uint64_t ins = ballot64(q < Qlo);
                    const uint64_t outs = ballot64(q > Qhi);
                    const uint64_t live = first ? ((1ull << (M + 2)) - 1ull) : (((1ull << M) - 1ull) << 2);
                    const bool amb = (~(ins | outs) & live) != 0ull;
                    const bool risk = ballot64(!((Rb - Lb) > thr)) != 0ull;
#ifdef PTE_PROFILE_SECTIONS
                    asm volatile("" :: "s"(ins), "s"(outs), "s"((int)amb), "s"((int)risk));
                    PROF_T(s2); prof2 += s2 - s1;
#endif
                    if (first) {
                        if (__builtin_expect((ins & 3ull) != 0ull || risk || amb, 0)) break;
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
        e.on_m2[8 * cl + 3] = (double)prof2;   // (overwrites the tail column in this debug build)
#endif
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

}  // namespace pte
