// pte_slice2.hpp -- k_explore_slice2: latency-optimised SliceSampler kernel (gfx950).
//
// Same algorithm, same draws, same results as k_explore_slice (pte_kernels.hpp) -- the two are
// compared bit-for-bit in tests -- but organised around what bounds it: ONE wavefront per replica
// issues roughly one instruction every 4-5 cycles, so time = instruction count on the sequential
// path of the replica.  The 64 lanes are therefore used to take work off that path:
//
//  * draws: the replica's counter-based stream is evaluated 64 draws at a time, already converted
//    to rand() doubles and randexp() ziggurat fast-path values (tables staged in LDS);
//  * candidates: for one coordinate, lane 0 / lane 1 evaluate the slice end points L / R and
//    lanes 2..M+1 the first M shrinkage proposals (which depend only on the draws and on earlier
//    proposals, not on log-density values), all through ONE pass of the log2(P)-add tree path.
//    If no doubling is needed (neither end point inside the slice) the first proposal inside the
//    slice is the reference's accepted point and consumes exactly the same draws.  Otherwise the
//    coordinate falls back to the sequential procedure of the reference (doubling, acceptance
//    check of Neal's doubling scheme, further shrinkage), restarted from the same stream position.
#pragma once
#include "pte_slice_common.hpp"

namespace pte {

template <int NLU, int M>
__global__ __launch_bounds__(64) void k_explore_slice2(EngineDev e, SliceParams sp) {
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
    if (is_ref_chain(e, c)) {
        iid_refresh_recorded<NLU>(e, cl, c, slot, e.sd[c], lane);
        return;
    }
    const double lp_before = lp_before_explore(e, c, slot);
    const int64_t d = e.d;
    double *xrow = e.x + (int64_t)slot * e.ld;
    const int B = (int)((d + 63) >> 6);
    const double nhp = e.nhp[c];
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
#ifdef PTE_DEBUG_COUNTERS
    long long dbg[5] = {0, 0, 0, 0, 0};
#define DBG(i, v) dbg[i] += (v)
#else
#define DBG(i, v)
#endif

    auto evalS = [&](double v) -> double {
        double t = v * v;
#pragma unroll
        for (int k = 0; k < NL; ++k) t = t + sib[k];
        return t;
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
                dr.ensure(2 + M, lane, s_we, s_ke);
                const double E = dr.randexp(lane, s_we, s_ke);
                dr.ensure(1 + M, lane, s_we, s_ke);
                const double z = lp - E;
                const double u0 = readlane_f64(dr.unit, dr.p);
                dr.p += 1;
                const double L = xold - w * u0;
                const double R = L + w;
                // ---- speculative batch: lanes 0,1 <- L,R ; lanes 2..M+1 <- proposals 1..M
                double Lb = L, Rb = R;
                double cand = (lane == 0) ? L : R;
#pragma unroll
                for (int n = 1; n <= M; ++n) {
                    const double u = readlane_f64(dr.unit, dr.p + n - 1);
                    const double v = Lb + u * (Rb - Lb);
                    cand = (lane == n + 1) ? v : cand;
                    const bool below = v < xold;
                    Lb = below ? v : Lb;
                    Rb = below ? Rb : v;
                }
                const double Sc = evalS(cand);
                const double lpc = nhp * Sc;
                const uint64_t inside = ballot64(z < lpc);
                const uint64_t acc = (inside >> 2) & ((1ull << M) - 1ull);
                // isapprox(Lbar, Rbar) can only have fired on one of the nested brackets if the final
                // one is tiny relative to the first: conservative test, exact handling in the fallback
                const double amax = fmax(fabs(L), fabs(R));
                const bool risk = !((Rb - Lb) > 1e-6 * amax);
                bool done = false;
                if ((inside & 3ull) == 0ull && !risk) {
                    steps_n += 1;                        // explorer_n_steps += p - K = 0 (no doubling)
                    if (acc != 0ull) {
                        const int n = (int)__builtin_ctzll(acc) + 1;
                        xf = readlane_f64(cand, n + 1);
                        S = readlane_f64(Sc, n + 1);
                        lp = nhp * S;
                        dr.p += n;
                        steps_sum += n; steps_n += 1;
                        acc_sum += 1; acc_n += 1;        // slice_accept is a no-op when R - L = w < 1.1 w
                        done = true;
                        DBG(0, 1);
                    }
                }
                if (!done) {
                    // ---- sequential procedure of the reference (SliceSampler.jl:97-237)
                    double lp_L = readlane_f64(lpc, 0), lp_R = readlane_f64(lpc, 1);
                    double LL = L, RR = R;
                    double Lbar, Rbar;
                    int n0;
                    if ((inside & 3ull) == 0ull && !risk) {
                        // no doubling, first M proposals all rejected: continue the shrinkage at n = M+1
                        dr.p += M;
                        Lbar = Lb; Rbar = Rb; n0 = M + 1;
                        DBG(1, 1);
                    } else {
                        int K = sp.p;
                        DBG(2, 1);
                        while (K > 0 && (z < lp_L || z < lp_R)) {
                            DBG(4, 1); DBG(3, 1);
                            const double V = dr.rand(lane, s_we, s_ke);
                            if (V <= 0.5) { LL = LL - (RR - LL); lp_L = nhp * evalS(LL); }
                            else { RR = RR + (RR - LL); lp_R = nhp * evalS(RR); }
                            K -= 1;
                        }
                        steps_sum += (sp.p - K); steps_n += 1;
                        Lbar = LL; Rbar = RR; n0 = 1;
                    }
                    bool fin = false;
                    for (int n = n0; n <= sp.max_iter; ++n) {
                        const double newpos = Lbar + dr.rand(lane, s_we, s_ke) * (Rbar - Lbar);
                        const double Snew = evalS(newpos);
                        const double newlp = nhp * Snew;
                        DBG(3, 1);
                        if (z < newlp) {
                            // slice_accept (:192-237)
                            double Lhat = LL, Rhat = RR, aL = lp_L, aR = lp_R;
                            bool Rstale = false, Lstale = false, D = false, ok = true;
                            while (Rhat - Lhat > w11) {
                                const double Mid = (Lhat + Rhat) / 2.0;
                                if ((xold < Mid && newpos >= Mid) || (xold >= Mid && newpos < Mid)) D = true;
                                if (newpos < Mid) { Rhat = Mid; Rstale = true; }
                                else { Lhat = Mid; Lstale = true; }
                                if (D) {
                                    if (Lstale) { aL = nhp * evalS(Lhat); Lstale = false; DBG(3, 1); }
                                    if (Rstale) { aR = nhp * evalS(Rhat); Rstale = false; DBG(3, 1); }
                                    if (z >= aL && z >= aR) { ok = false; break; }
                                }
                            }
                            acc_n += 1;
                            if (ok) {
                                acc_sum += 1;
                                xf = newpos; S = Snew; lp = newlp;
                                steps_sum += n; steps_n += 1;
                                fin = true;
                                break;
                            }
                        }
                        if (newpos < xold) Lbar = newpos; else Rbar = newpos;
                        if (jl_isapprox(Lbar, Rbar)) {
                            xf = xold;                    // lp(state) recomputed == cached value (pure function)
                            steps_sum += n; steps_n += 1;
                            fin = true;
                            break;
                        }
                    }
                    if (!fin) { err = ERR_SLICE_MAX_ITER; err_coord = (int)(base + l); break; }
                }
                if (!isfinite(lp)) { err = ERR_SLICE_INVALID_LP; err_coord = (int)(base + l); break; }
                if (lane == l) X = xf;
                // ---- siblings of coordinate l+1: the subtree just completed on the left (level r),
                //      untouched right subtrees below it from the block-start butterfly
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
            {   // new block sum = level-6 node on the path of the last coordinate
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
#ifdef PTE_DEBUG_COUNTERS
        for (int i = 0; i < 5; ++i) e.on_m2[5 * cl + i] += (double)dbg[i];   // debug builds only: reuses on_m2 (needs d >= 5N)
#endif
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

}  // namespace pte
