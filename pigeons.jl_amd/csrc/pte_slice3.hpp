// pte_slice3.hpp -- k_explore_slice3: SliceSampler kernel with FILTERED slice-membership predicates.
//
// Bit-identical to k_explore_slice / k_explore_slice2 and to the oracle (tests compare all three),
// but the O(log d) tree path is evaluated only where its VALUE is needed (the accepted point).
// Every other use of the log potential in SliceSampler.jl is a comparison  z < lp(x with x_c = v):
//
//   lp_fl(v) = fl(nhp * S_fl(v)),  S_fl(v) = fl(...fl(fl(v*v) + s_0) + ... + s_{NL-1}),  s_k >= 0, nhp < 0
//
// Standard forward error analysis of a sum of non-negative terms gives
//   S_fl(v) = (v^2 + R)(1 + theta), |theta| <= gamma_{NL+1},  R = sum_k s_k (exact),
// hence  [z < lp_fl(v)]  <=>  (v^2 + R)(1 + eta) < T,  T = z / nhp,  |eta| <= gamma_{NL+2} < 2e-15.
// With R recovered from the cached root as S_cur - x_old^2 (error <= 2e-15 S_cur) the decision is
//   TRUE  if v*v < Q - m,   FALSE if v*v > Q + m,   Q = T - (S_cur - x_old^2),
// for any margin m >= 1e-14 (|T| + S_cur).  The kernel uses m = 1e-11 (|T| + S_cur), i.e. a 1000x
// safety factor; in the remaining sliver (probability ~1e-11 per test, and whenever anything is not
// finite) it evaluates the exact tree path.  Decisions, draws and states are therefore exactly those
// of the full recompute; the cached lp / root of an accepted point always come from the exact path.
#pragma once
#include "pte_slice2.hpp"

namespace pte {

template <int NLU>
__global__ __launch_bounds__(64) void k_explore_slice3(EngineDev e, SliceParams sp) {
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

    auto evalS = [&](double v) -> double {
        double t = v * v;
#pragma unroll
        for (int k = 0; k < NL; ++k) t = t + sib[k];
        return t;
    };
    // [z < lp(x with x_c = v)] : filter, exact path in the sliver (also catches NaN / Inf)
    auto inside = [&](double v) -> bool {
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
            for (int l = 0; l < nl; ++l) {
                const double xold = readlane_f64(X, l);
                // ---- slice_sample_coord! (SliceSampler.jl:89-95)
                const double E = dr.randexp(lane, s_we, s_ke);
                z = lp - E;
                {
                    const double T = z * inv_nhp;
                    const double Q = T - (S - xold * xold);
                    const double m = 1e-11 * (fabs(T) + S);
                    Qlo = Q - m; Qhi = Q + m;
                }
                // slice_double (:97-126)
                double L = xold - w * dr.rand(lane, s_we, s_ke);
                double R = L + w;
                bool in_L = inside(L), in_R = inside(R);
                int K = sp.p;
                while (__builtin_expect(K > 0 && (in_L || in_R), 0)) {
                    const double V = dr.rand(lane, s_we, s_ke);
                    if (V <= 0.5) { L = L - (R - L); in_L = inside(L); }
                    else { R = R + (R - L); in_R = inside(R); }
                    K -= 1;
                }
                steps_sum += (sp.p - K); steps_n += 1;
                const bool doubled = (R - L) > w11;       // slice_accept is a no-op otherwise
                // slice_shrink! (:144-186)
                double Lbar = L, Rbar = R;
                const double thr = 1e-6 * fmax(fabs(L), fabs(R));   // isapprox pre-filter (nested brackets)
                bool fin = false;
                xf = xold;
                for (int n = 1; n <= sp.max_iter; ++n) {
                    const double W = Rbar - Lbar;
                    if (__builtin_expect(n > 1 && !(W > thr), 0)) {
                        if (jl_isapprox(Lbar, Rbar)) {       // keep old point; lp(state) == cached lp
                            steps_sum += (n - 1); steps_n += 1;
                            fin = true;
                            break;
                        }
                    }
                    const double newpos = Lbar + dr.rand(lane, s_we, s_ke) * W;
                    if (inside(newpos)) {
                        bool ok = true;
                        if (__builtin_expect(doubled, 0)) {
                            // slice_accept (:192-237): the tests at the bisection points use the same predicate
                            double Lhat = L, Rhat = R;
                            bool oL = !in_L, oR = !in_R;         // "z >= lp" at the current end points
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
                            S = evalS(newpos);                   // exact root / cached lp of the new state
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
                if (__builtin_expect(!fin, 0)) { err = ERR_SLICE_MAX_ITER; err_coord = (int)(base + l); break; }
                if (__builtin_expect(!isfinite(lp), 0)) { err = ERR_SLICE_INVALID_LP; err_coord = (int)(base + l); break; }
                if (lane == l) X = xf;
                if (l + 1 < nl) {
                    const int l1 = l + 1;
                    const int r = __builtin_ctz((unsigned)l1);
                    double t = xf * xf;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (k < r) { t = t + sib[k]; sib[k] = readlane_f64(U[k], l1 ^ (1 << k)); }
                        else if (k == r) sib[k] = t;
                    }
                }
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
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

}  // namespace pte
