// pte_automala.hpp -- k_explore_automala: AutoMALA (reference src/explorers/AutoMALA.jl:84-275,
// src/explorers/hamiltonian_dynamics.jl:48-102, src/explorers/Preconditioner.jl:57-77) on the
// device log-potential families, one wavefront per replica.
//
// Every vector of the algorithm (state, momentum, preconditioner, start state, the two "before"
// copies, gradient) lives in registers: lane l holds elements 64j + l, j < E.  A log-density / norm
// evaluation is E DPP wave reductions (the fixed tree of pte_device.hpp) -- no LDS, no HBM traffic
// between the initial load and the final store of the state.  Log potentials:
//   TGT_MVN    ScaledPrecisionNormalLogPotential, analytic gradient (src/paths/ScaledPrecisionNormalPath.jl:19-34)
//   TGT_FUNNEL InterpolatedAD of {ScaledPrecisionNormal(p0) reference, Neal's funnel}
//              (src/explorers/BufferedAD.jl:89-112; funnel test/supporting/dimensional-analysis.jl:36-48)
#pragma once
#include "pte_automala_params.hpp"

namespace pte {

// tree over the E block sums (uniform values), E a power of two
template <int E>
__device__ __forceinline__ double block_tree(double (&s)[E]) {
#pragma unroll
    for (int w = E; w > 1; w >>= 1)
#pragma unroll
        for (int i = 0; i < w / 2; ++i) s[i] = s[2 * i] + s[2 * i + 1];
    return s[0];
}
template <int E>
__device__ __forceinline__ double tree_sum_regs(const double (&t)[E]) {
    if constexpr (E >= 4) {                            // block sums four at a time, blocks 2i and 2i+1 added in the same pass
        double s[E / 2];
#pragma unroll
        for (int j0 = 0; j0 < E; j0 += 4) {
            double v[4], o[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = t[j0 + j];
            wave_sum_pairs<4>(v, o);
            s[j0 / 2] = o[0]; s[j0 / 2 + 1] = o[1];
        }
        return block_tree<E / 2>(s);
    } else {
#ifndef PTE_WSP_ROWS_ONLY
        if constexpr (E == 2) return wave_sum_pair(t[0], t[1]);
#endif
        double s[E];
#pragma unroll
        for (int j = 0; j < E; ++j) s[j] = t[j];
        wave_sum_dpp_multi<E>(s);                      // their tree levels interleaved
        return block_tree<E>(s);
    }
}
// K reductions at once (K = 2 or 4, E >= 2): the block sums of all of them in lockstep, blocks 2i and 2i+1 of each at a time
template <int E, int K>
__device__ __forceinline__ void tree_sum_regs_multi(const double (&t)[K][E], double (&out)[K]) {
    if constexpr (E >= 2 && (K == 2 || K == 4)) {
        double s[K][E / 2];
#pragma unroll
        for (int j0 = 0; j0 < E; j0 += 2) {
            double v[2 * K], o[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { v[2 * k] = t[k][j0]; v[2 * k + 1] = t[k][j0 + 1]; }
            wave_sum_pairs<2 * K>(v, o);
#pragma unroll
            for (int k = 0; k < K; ++k) s[k][j0 / 2] = o[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = block_tree<E / 2>(s[k]);
    } else {
        double s[K][E];
        constexpr int G = E < 2 ? E : 2;
#pragma unroll
        for (int j0 = 0; j0 < E; j0 += G) {
            double v[K * G];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < G; ++j) v[k * G + j] = t[k][j0 + j];
            wave_sum_dpp_multi<K * G>(v);
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < G; ++j) s[k][j0 + j] = v[k * G + j];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = block_tree<E>(s[k]);
    }
}
template <int E>
__device__ __forceinline__ double sqr_norm_regs(const double (&v)[E]) {
    double t[E];
#pragma unroll
    for (int j = 0; j < E; ++j) t[j] = v[j] * v[j];
    return tree_sum_regs<E>(t);
}

// FULL: d == 64 E, every lane of every block holds an element -- the masks, selects and EXEC-guarded divisions of a ragged last
// block vanish (25 instructions of ~600 per leapfrog at E = 2)
template <int E, int TGT, bool FULL = false>
struct AmTarget {
    int64_t d; int lane;
    double nhp, nprec;          // MVN: -0.5*prec, -prec of this chain
    double beta, omb, ref_nhp, ref_nprec, log3;   // funnel path
    // GaussianReference end of the path (variational leg): per-coordinate constants in registers when `vr`
    bool vr = false;
    static constexpr bool V_IN_REGS = (E < 16);      // d > 512: the replica's own vectors already fill the register file
    double vm[V_IN_REGS ? E : 1], vc0[V_IN_REGS ? E : 1], vi2[V_IN_REGS ? E : 1], vgf[V_IN_REGS ? E : 1];
    const double *pm = nullptr, *pc0 = nullptr, *pi2 = nullptr, *pgf = nullptr;
    __device__ __forceinline__ void load_variational(const EngineDev &e) {
        pm = e.v_mean; pc0 = e.v_c0; pi2 = e.v_i2; pgf = e.v_gf;
        if (V_IN_REGS) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const bool ok = valid(j);
                const int64_t i = 64 * (int64_t)j + lane;
                vm[j % (V_IN_REGS ? E : 1)] = ok ? e.v_mean[i] : 0.0; vc0[j % (V_IN_REGS ? E : 1)] = ok ? e.v_c0[i] : 0.0;
                vi2[j % (V_IN_REGS ? E : 1)] = ok ? e.v_i2[i] : 0.0; vgf[j % (V_IN_REGS ? E : 1)] = ok ? e.v_gf[i] : 0.0;
            }
        }
    }
    __device__ __forceinline__ double VM(int j) const { return V_IN_REGS ? vm[j % (V_IN_REGS ? E : 1)] : (valid(j) ? pm[64 * j + lane] : 0.0); }
    __device__ __forceinline__ double VC0(int j) const { return V_IN_REGS ? vc0[j % (V_IN_REGS ? E : 1)] : (valid(j) ? pc0[64 * j + lane] : 0.0); }
    __device__ __forceinline__ double VI2(int j) const { return V_IN_REGS ? vi2[j % (V_IN_REGS ? E : 1)] : (valid(j) ? pi2[64 * j + lane] : 0.0); }
    __device__ __forceinline__ double VGF(int j) const { return V_IN_REGS ? vgf[j % (V_IN_REGS ? E : 1)] : (valid(j) ? pgf[64 * j + lane] : 0.0); }
    // gaussian_logdensity (GaussianReference.jl:43-49) with the fixed tree in place of the sequential sum
    __device__ __forceinline__ double variational_lp(const double (&x)[E]) const {
        double t[E];
#pragma unroll
        for (int j = 0; j < E; ++j) { const double dx = x[j] - VM(j); t[j] = valid(j) ? (VC0(j) - VI2(j) * (dx * dx)) : 0.0; }
        return tree_sum_regs<E>(t);
    }
    __device__ __forceinline__ double ref_lp(const double (&x)[E], double S) const { if (__builtin_expect(vr, 0)) return variational_lp(x); return ref_nhp * S; }
    __device__ __forceinline__ bool valid(int j) const { return FULL || 64 * (int64_t)j + lane < d; }

    // funnel: log density and (optionally) gradient; S = sum x^2 supplied by the caller
    __device__ __forceinline__ double funnel(const double (&x)[E], double (*g)[E]) const {
        const double y = readlane_f64(x[0], 0);
        const double sigma = exp(y / 2.0);
        const double logsigma = log(sigma);
        const double LOG2PI = 1.8378770664093453;
        double t[E], zi[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            zi[j] = x[j] / sigma;
            t[j] = valid(j) ? (-(zi[j] * zi[j] + LOG2PI) / 2.0 - logsigma) : 0.0;
        }
        const double zv = y / 3.0;
        if (lane == 0) t[0] = -(zv * zv + LOG2PI) / 2.0 - log3;
        const double lp = tree_sum_regs<E>(t);
        if (g) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                (*g)[j] = valid(j) ? -(zi[j] / sigma) : 0.0;
                t[j] = valid(j) ? (zi[j] * zi[j] - 1.0) / 2.0 : 0.0;
            }
            if (lane == 0) t[0] = -(y / 9.0);
            const double gy = tree_sum_regs<E>(t);
            if (lane == 0) (*g)[0] = gy;
        }
        return lp;
    }
    // the same, with S = sum x^2 (and, when q is given, Q = sum q^2: the kinetic energy the caller needs next) taken alongside:
    // the block sums of the two to four reductions are independent and run in lockstep (wave_sum_dpp_multi) instead of one
    // dependent DPP chain after the other
    template <bool GRAD, bool WITH_Q>
    __device__ __forceinline__ double funnel_and_sqr_norm(const double (&x)[E], double (&g)[E], double &S, const double (&q)[E], double &Q) const {
        const double y = readlane_f64(x[0], 0);
        const double sigma = exp(y / 2.0);
        const double logsigma = log(sigma);
        const double LOG2PI = 1.8378770664093453;
        constexpr int K = 2 + (GRAD ? 1 : 0) + (WITH_Q ? 1 : 0), KG = 2, KQ = GRAD ? 3 : 2;
        double t[K][E], out[K];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double zi = x[j] / sigma;
            t[0][j] = x[j] * x[j];
            t[1][j] = valid(j) ? (-(zi * zi + LOG2PI) / 2.0 - logsigma) : 0.0;
            if (GRAD) {
                g[j] = valid(j) ? -(zi / sigma) : 0.0;
                t[KG % K][j] = valid(j) ? (zi * zi - 1.0) / 2.0 : 0.0;
            }
            if (WITH_Q) t[KQ % K][j] = q[j] * q[j];
        }
        const double zv = y / 3.0;
        if (lane == 0) { t[1][0] = -(zv * zv + LOG2PI) / 2.0 - log3; if (GRAD) t[KG % K][0] = -(y / 9.0); }
        tree_sum_regs_multi<E, K>(t, out);
        S = out[0];
        if (GRAD) { if (lane == 0) g[0] = out[KG % K]; }
        if (WITH_Q) Q = out[KQ % K];
        return out[1];
    }
    // log_potentials[chain](x) as a plain callable: InterpolatedLogPotential(x) (src/paths/InterpolatedLogPotential.jl:9-16)
    // WITH its beta == 0 / beta == 1 short-circuits -- what SliceSampler evaluates (the AD form below has none)
    __device__ __forceinline__ double path_lp(const double (&x)[E]) const {
        const double S = sqr_norm_regs<E>(x);
        if (TGT == TGT_MVN) return nhp * S;
        if (beta == 0.0) return ref_lp(x, S);
        if (beta == 1.0) return funnel(x, nullptr);
        const double l1 = ref_lp(x, S);
        const double l2 = funnel(x, nullptr);
        return omb * l1 + beta * l2;
    }
    // LogDensityProblems.logdensity
    __device__ __forceinline__ double logdensity(const double (&x)[E]) const {
        if (TGT == TGT_MVN) return nhp * sqr_norm_regs<E>(x);
        double S, l2, dummy[E], dq;
        if (E <= 4) l2 = funnel_and_sqr_norm<false, false>(x, dummy, S, x, dq);
        else { S = sqr_norm_regs<E>(x); l2 = funnel(x, nullptr); }
        const double l1 = ref_lp(x, S);
        return omb * l1 + beta * l2;
    }
    // LogDensityProblems.logdensity_and_gradient; WITH_Q: also Q = sum q^2 (fixed tree), reduced in lockstep with the sums of the density
    template <bool WITH_Q>
    __device__ __forceinline__ double logdensity_and_gradient_q(const double (&x)[E], double (&g)[E], const double (&q)[E], double &Q) const {
        double S = 0.0;
        if (TGT == TGT_MVN) {
            if (WITH_Q) {
                double t[2][E], out[2];
#pragma unroll
                for (int j = 0; j < E; ++j) { t[0][j] = x[j] * x[j]; t[1][j] = q[j] * q[j]; }
                tree_sum_regs_multi<E, 2>(t, out);
                S = out[0]; Q = out[1];
            } else S = sqr_norm_regs<E>(x);
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] = nprec * x[j];
            return nhp * S;
        }
        double logdens = 0.0;
        double g2[E];
        double l2;
        if (E <= 4) l2 = funnel_and_sqr_norm<true, WITH_Q>(x, g2, S, q, Q);
        else { S = sqr_norm_regs<E>(x); l2 = funnel(x, &g2); if (WITH_Q) Q = sqr_norm_regs<E>(q); }
        const double l1 = ref_lp(x, S);
        logdens += l1 * omb;
        logdens += l2 * beta;
        if (__builtin_expect(vr, 0)) {               // BufferedAD{GaussianReference}: -1/s^2 (x - m)   (the fixed reference falls through)
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] = (VGF(j) * (x[j] - VM(j))) * omb + g2[j] * beta;
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] = (ref_nprec * x[j]) * omb + g2[j] * beta;
        }
        return logdens;
    }
    __device__ __forceinline__ double logdensity_and_gradient(const double (&x)[E], double (&g)[E]) const {
        double dq;
        return logdensity_and_gradient_q<false>(x, g, x, dq);
    }
};

// SLICE = true instantiates the same prologue (reference-chain refresh, state load) and epilogue (swap statistics, recorders)
// around the SliceSampler sweep instead of the Langevin refreshes: a separate kernel, so that neither pays for the other's registers.
#ifndef PTE_AM_PERMUTE
#define PTE_AM_PERMUTE 97                  // (a prime: coprime to every chain count it does not divide)
#endif
// Neighbouring chains do similar work (the slow ones -- most step-size trials -- sit next to the reference) and neighbouring
// workgroups share a CU, whose FP64 pipe the four SIMDs contend for: a stride permutation of workgroup -> chain spreads the slow
// chains over the chip.  Measured at C3, interleaved on one box: 0.231 -> 0.227 ms / scan (the same permutation makes the slice
// kernel 2.5 % SLOWER -- it is not applied there).
__device__ __forceinline__ int64_t am_chain_of_workgroup(int64_t K, int64_t wg) { return (K % PTE_AM_PERMUTE) ? (wg * PTE_AM_PERMUTE) % K : wg; }

// DIRECT (the scan loop with several chains per workgroup, k_scans_automala_wg): `wg` IS the local chain, and the table staging ends in a
// wave-level wait instead of a workgroup barrier -- every wave writes all the (identical) entries itself, so it only has to see its own stores
template <int E, int TGT, bool SLICE = false, bool FULL = false, bool DIRECT = false>
__device__ __forceinline__ void automala_body(EngineDev e, AmParams ap, const int64_t wg) {      // wg: blockIdx.x
    constexpr int NLU = (E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : 4);
    const int lane = lane_id();
    // the ziggurat tables of the momentum draws, staged once: a global gather per block of draws costs a memory round trip each time
    __shared__ double s_wi[256];
    __shared__ unsigned long long s_ki[256];
    __shared__ double s_fi[256];
    if (!SLICE) {
        for (int i = lane; i < 256; i += 64) { s_wi[i] = ZIG_WI[i]; s_ki[i] = ZIG_KI[i]; s_fi[i] = ZIG_FI[i]; }
        if constexpr (DIRECT) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
    const int64_t cl = DIRECT ? wg : am_chain_of_workgroup(e.K, wg);
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    const int64_t d = e.d;
    double *xrow = e.x + (int64_t)slot * e.ld;
    AmTarget<E, TGT, FULL> T;
    T.d = d; T.lane = lane;
    T.nhp = e.nhp[c]; T.nprec = e.nprec[c];
    T.beta = e.beta[c]; T.omb = 1.0 - T.beta;
    T.ref_nhp = -0.5 * ap.ref_prec; T.ref_nprec = -ap.ref_prec; T.log3 = ap.log3;
    const bool v_on = (TGT == TGT_FUNNEL) && e.v_use != nullptr;      // a GaussianReference is active on this engine
    if (v_on) { T.load_variational(e); T.vr = e.v_use[c] != 0; }

    double x[E];
    if (is_ref_chain(e, c)) {
        if (e.compose_phase == 2) return;
        const double lp0 = lp_before_explore(e, c, slot);
        double S0;
        if (v_on && T.vr) {
            // sample_iid!(::GaussianReference) (GaussianReference.jl:33-40): x_i = randn * sd_i + mean_i, in draw order
            SeqRng r0{e.rng[2 * slot], e.rng[2 * slot + 1]};
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int nl = (int)max((int64_t)0, min((int64_t)64, d - 64 * (int64_t)j));
                x[j] = 0.0;
                if (nl > 0) {
                    const double z = wave_randn_block(r0, lane, nl);
                    x[j] = lane < nl ? z * e.v_std[64 * j + lane] + T.VM(j) : 0.0;
                    if (lane < nl) xrow[64 * j + lane] = x[j];
                }
            }
            S0 = sqr_norm_regs<E>(x);
            if (lane == 0) { e.suff[slot] = S0; e.rng[2 * slot] = r0.seed; }
        } else {
            S0 = iid_refresh<NLU>(e, slot, e.sd[c], lane);            // sample_iid! at the reference (pigeons.jl:104-105)
            __threadfence_block();
#pragma unroll
            for (int j = 0; j < E; ++j) x[j] = T.valid(j) ? xrow[64 * j + lane] : 0.0;
        }
        double l20 = 0.0, l30 = 0.0;
        if (TGT == TGT_FUNNEL) {
            l20 = T.funnel(x, nullptr);
            if (lane == 0) e.suff2[slot] = l20;
            if (v_on) { l30 = T.variational_lp(x); if (lane == 0) e.suff3[slot] = l30; }
        }
        record_after_explore_impl(e, cl, c, slot, lane, lp0, S0, l20, l30);
        return;
    }
    const double lp_before = lp_before_explore(e, c, slot);
#pragma unroll
    for (int j = 0; j < E; ++j) x[j] = T.valid(j) ? xrow[64 * j + lane] : 0.0;

    SeqRng r{e.rng[2 * slot], e.rng[2 * slot + 1]};
    // build_preconditioner! (Preconditioner.jl:57-77)
    double M[E];
#pragma unroll
    for (int j = 0; j < E; ++j) M[j] = 1.0;
    if (ap.target_std != nullptr && ap.precond != 0) {
        double sdv[E];
#pragma unroll
        for (int j = 0; j < E; ++j) sdv[j] = T.valid(j) ? ap.target_std[64 * j + lane] : 1.0;
        if (ap.precond == 1) {
#pragma unroll
            for (int j = 0; j < E; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : 1.0 / sdv[j];
        } else {
            const double u = r.rand();
            if (u <= ap.p0) {
#pragma unroll
                for (int j = 0; j < E; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : 1.0 / sdv[j];
            } else if (u <= ap.p0 + ap.p1) {
                // ones
            } else {
                const double mix = r.rand(), rmix = 1.0 - mix;
#pragma unroll
                for (int j = 0; j < E; ++j) M[j] = sdv[j] == 0.0 ? 1.0 : mix + rmix / sdv[j];
            }
        }
    }

#ifdef PTE_PROFILE_AM                      // debug builds only (tools/prof_automala.py): shader-clock time per section of the refresh loop
    uint64_t am_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t am_tlast = __builtin_amdgcn_s_memtime();
    const uint64_t am_rt0 = __builtin_amdgcn_s_memrealtime();
#define AM_STAMP(k) do { asm volatile("" ::: "memory"); const uint64_t t_ = __builtin_amdgcn_s_memtime(); am_prof[k] += t_ - am_tlast; am_tlast = t_; asm volatile("" ::: "memory"); } while (0)
#else
#define AM_STAMP(k) do { } while (0)
#endif
    double p[E], g[E], xs[E], xb[E], pb[E];
    long long steps_sum = 0; int steps_n = 0;
    double fac_sum = 0.0; int fac_n = 0;
    int rev_sum = 0, rev_n = 0;
    double acc_sum = 0.0; int acc_n = 0;
    int err = 0;

    // The reference re-evaluates the log density / gradient at points where it already has them (the start point of
    // every trial leapfrog of a step-size search; log_joint right after a leapfrog).  They are pure functions of
    // the state, so the kernel evaluates each point once and reuses the bits: identical results, ~2.5x fewer funnel
    // evaluations per auto_step_size call.
    double g0[E];            // conditioned gradient at the current x (valid after grad_at_start, until x moves for good)
    double lp0 = 0.0;        // log density at the current x
    double pp0 = 0.0;        // |p|^2 of the fresh momentum, reduced alongside the sums of the density
    auto grad_at_start = [&]() {
        lp0 = T.template logdensity_and_gradient_q<true>(x, g0, p, pp0);
#pragma unroll
        for (int j = 0; j < E; ++j) g0[j] = g0[j] / M[j];
    };
    auto kinetic = [&]() -> double { return 0.5 * sqr_norm_regs<E>(p); };
    // hamiltonian_dynamics! with n_steps = 1 from a point whose conditioned gradient is g0; logp_out = log density
    // at the new position (== what log_joint would recompute there)
    // ke_out = 0.5 |p|^2 of the momentum it leaves behind (what log_joint would recompute next)
    auto leap_frog = [&](double eps, double &logp_out, double &ke_out) -> bool {
        const double half = eps / 2;
#pragma unroll
        for (int j = 0; j < E; ++j) p[j] = p[j] + half * g0[j];
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = x[j] + eps * (p[j] / M[j]);
        double pp_mid;                                    // |p|^2 after the first half step: independent of the gradient, reduced with it
        const double logp = T.template logdensity_and_gradient_q<true>(x, g, p, pp_mid);
        logp_out = logp;
#pragma unroll
        for (int j = 0; j < E; ++j) g[j] = g[j] / M[j];
        const double ke_mid = 0.5 * pp_mid;
        ke_out = ke_mid;
        const double cur = logp - ke_mid;
        if (__builtin_expect(!isfinite(cur), 0)) return false;          // (rare: laid out behind the loop -- a lone wave refetches after every taken branch)
#pragma unroll
        for (int j = 0; j < E; ++j) p[j] = p[j] + half * g[j];
        const double sq = sqr_norm_regs<E>(p);
        ke_out = 0.5 * sq;
        if (__builtin_expect(!isfinite(sq), 0)) return false;
        return true;
    };
    // auto_step_size (:184-214): returns the exponent; h_before = log_joint at the start point (g0 / lp0 valid there).
    // keep: the forward search.  The reference follows it with leap_frog!(start point, step_size * 2^exponent) -- a leapfrog the search
    // has ALREADY made from the same point with the same step: its last trial when it shrank (exponent = -n) or did not move
    // (exponent = 0), its last but one when it grew (exponent = n - 1).  That trial's outcome (state, momentum, conditioned gradient,
    // log density, kinetic energy, return value) is kept instead of being thrown away and recomputed: the same bits, one gradient
    // evaluation per refresh less.
    double xk[E], pk[E], gk[E], lpk = 0.0, kek = 0.0; bool okk = true;
    auto auto_step_size = [&](double lower, double upper, double h_before, bool keep) -> int {
#pragma unroll
        for (int j = 0; j < E; ++j) { xb[j] = x[j]; pb[j] = p[j]; }
        double eps = ap.step_size;
        // one trial; save = this trial is (so far) the one the proposal would repeat
        double t_lp = 0.0, t_ke = 0.0; bool t_ok = true;
        auto trial = [&](double ee) -> double {
            t_ok = leap_frog(ee, t_lp, t_ke);
            return (t_lp - t_ke) - h_before;
        };
        auto save_trial = [&]() {
#pragma unroll
            for (int j = 0; j < E; ++j) { xk[j] = x[j]; pk[j] = p[j]; gk[j] = g[j]; }
            lpk = t_lp; kek = t_ke; okk = t_ok;
        };
        auto restore = [&]() {
#pragma unroll
            for (int j = 0; j < E; ++j) { x[j] = xb[j]; p[j] = pb[j]; }
        };
        double diff = trial(eps);
        if (keep) save_trial();
        restore();
        int n_steps = 0, exponent = 0;
        if (!isfinite(diff) || diff < lower) {
            for (int n = 1;; ++n) {
                eps /= 2.0;
                diff = trial(eps);
                if (keep) save_trial();                       // shrinking: the last trial is the one
                restore();
                if (eps == 0.0) { err = ERR_AM_STEP; break; }
                if (diff > lower) { n_steps = n; exponent = -n; break; }
            }
        } else if (diff > upper) {
            for (int n = 1;; ++n) {
                eps *= 2.0;
                diff = trial(eps);
                const bool stop = !isfinite(diff) || diff < upper;
                if (keep && !stop) save_trial();              // growing: the last trial BEFORE the one that went too far
                restore();
                if (stop) { n_steps = n; exponent = n - 1; break; }
            }
        }
        steps_sum += 1 + n_steps; steps_n += 1;
        if (e.am_log != nullptr && lane == 0 && fac_n < e.am_log_cap) e.am_log[(e.trace_idx * e.K + cl) * e.am_log_cap + fac_n] = (int16_t)exponent;
        fac_sum += ldexp(1.0, exponent); fac_n += 1;
        return exponent;
    };

    if constexpr (SLICE) {
        // ---- step!(::SliceSampler) on a path without a closed-form single-coordinate update: the reference's procedure as it
        // stands (slice_sample! :43-62, slice_sample_coord! :89-95, slice_double :97-126, slice_shrink! :144-186, slice_accept
        // :192-237), every log potential evaluated in full (E wave reductions of the fixed tree + the funnel's exp / log),
        // the replica's stream consumed through 64 buffered draws.  All control flow is uniform: every value it branches on
        // comes out of a wave reduction.
        WaveDraws dr;
        dr.init(r.seed, r.gamma, lane);
        double lp = T.path_lp(x);                               // cached_log_potential (:32-41)
        if (lp == -INFINITY) { if (lane == 0) set_error(e, ERR_SLICE_SUPPORT, (int)c, -1); return; }
        const double w = ap.slice_w, w11 = 1.1 * ap.slice_w;
        for (int pass = 0; pass < ap.slice_n_passes && !err; ++pass) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int nl = (int)max((int64_t)0, min((int64_t)64, d - 64 * (int64_t)j));
                for (int l = 0; l < nl && !err; ++l) {
                    const double xold = readlane_f64(x[j], l);
                    auto eval = [&](double v) -> double { x[j] = (lane == l) ? v : x[j]; return T.path_lp(x); };   // pointer[] = v; lp(state)
                    double Ex;
                    {
                        const uint64_t raw = dr.next_raw(lane);
                        const uint64_t ri = raw & MASK52;
                        const int idx = (int)(ri & 0xFF);
                        Ex = (double)ri * ZIG_WE[idx];
                        if (!(ri < ZIG_KE[idx])) { SeqRng sq = dr.to_seq(); Ex = randexp_from_raw(sq, raw); dr.from_seq(sq, lane); }
                    }
                    const double z = lp - Ex;
                    double L = xold - w * dr.rand(lane);
                    double R = L + w;
                    int K = ap.slice_p;
                    double lp_L = eval(L), lp_R = eval(R);
                    while (K > 0 && (z < lp_L || z < lp_R)) {
                        const double V = dr.rand(lane);
                        if (V <= 0.5) { L = L - (R - L); lp_L = eval(L); }
                        else { R = R + (R - L); lp_R = eval(R); }
                        K -= 1;
                    }
                    steps_sum += ap.slice_p - K; steps_n += 1;
                    double Lbar = L, Rbar = R;
                    bool done = false;
                    for (int n = 1; n <= ap.slice_max_iter; ++n) {
                        const double newpos = Lbar + dr.rand(lane) * (Rbar - Lbar);
                        const double newlp = eval(newpos);
                        bool take = z < newlp;
                        if (take) {                              // slice_accept
                            double Lhat = L, Rhat = R, aL = lp_L, aR = lp_R;
                            bool Rstale = false, Lstale = false, D = false;
                            while (Rhat - Lhat > w11) {
                                const double Mid = (Lhat + Rhat) / 2.0;
                                if ((xold < Mid && newpos >= Mid) || (xold >= Mid && newpos < Mid)) D = true;
                                if (newpos < Mid) { Rhat = Mid; Rstale = true; } else { Lhat = Mid; Lstale = true; }
                                if (D) {
                                    if (Lstale) { aL = eval(Lhat); Lstale = false; }
                                    if (Rstale) { aR = eval(Rhat); Rstale = false; }
                                    if (z >= aL && z >= aR) { take = false; break; }
                                }
                            }
                            acc_sum += take ? 1.0 : 0.0; acc_n += 1;
                        }
                        if (take) {
                            x[j] = (lane == l) ? newpos : x[j]; lp = newlp;
                            steps_sum += n; steps_n += 1; done = true; break;
                        }
                        if (newpos < xold) Lbar = newpos; else Rbar = newpos;
                        if (jl_isapprox(Lbar, Rbar)) {
                            lp = eval(xold);
                            steps_sum += n; steps_n += 1; done = true; break;
                        }
                    }
                    if (!done) { if (lane == 0) set_error(e, ERR_SLICE_MAX_ITER, (int)c, 64 * j + l); return; }
                    if (!isfinite(lp)) { if (lane == 0) set_error(e, ERR_SLICE_INVALID_LP, (int)c, 64 * j + l); return; }
                }
            }
        }
        r.seed = dr.final_seed();
    } else
    for (int it = 0; it < ap.n_refresh && !err; ++it) {
        AM_STAMP(7);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            xs[j] = x[j];
            const int nl = FULL ? 64 : (int)max((int64_t)0, min((int64_t)64, d - 64 * (int64_t)j));
            p[j] = 0.0;
            if (nl > 0) { const double v = wave_randn_block(r, lane, nl, s_wi, s_ki, s_fi); p[j] = lane < nl ? v : 0.0; }
        }
        AM_STAMP(0);
        // Log density and conditioned gradient at the start point.  The reference evaluates them afresh in every refresh; they are
        // pure functions of x (the preconditioner is fixed for the scan), and from the second refresh on x is either the point the
        // last proposal leapfrog ended at -- evaluated there -- or the last start point -- evaluated then: the bits are carried
        // instead of recomputed (one of ~7 gradient evaluations per refresh), and only |p|^2 of the new momentum is reduced.
        if (it == 0) grad_at_start();
        else pp0 = sqr_norm_regs<E>(p);
        const double lp_s = lp0;
        double g_s[E];
#pragma unroll
        for (int j = 0; j < E; ++j) g_s[j] = g0[j];
        const double init_joint = lp0 - 0.5 * pp0;
        AM_STAMP(1);
        if (!isfinite(init_joint)) { err = ERR_AM_DENSITY; break; }
        if (ap.mala) {                                   // mala! (MALA.jl:79-96)
            double lpn, ken;
            leap_frog(ap.step_size, lpn, ken);
#pragma unroll
            for (int j = 0; j < E; ++j) p[j] = p[j] * -1.0;
            const double ex = exp((lpn - ken) - init_joint);          // |-p|^2 == |p|^2 bit for bit
            const double probability = ex < 1.0 ? ex : (isnan(ex) ? ex : 1.0);
            acc_sum += probability; acc_n += 1;
            if (!(r.rand() < probability)) {
#pragma unroll
                for (int j = 0; j < E; ++j) x[j] = xs[j];             // (lp0, g0 stay those of the start point)
            } else {
                lp0 = lpn;
#pragma unroll
                for (int j = 0; j < E; ++j) g0[j] = g[j];
            }
            steps_sum += 1; steps_n += 1;
            continue;
        }
        const double ua = r.rand(), ub = r.rand();
        const double lower = log(ua < ub ? ua : ub), upper = log(ua < ub ? ub : ua);
        AM_STAMP(2);
        const int proposed = auto_step_size(lower, upper, init_joint, true);
        if (err) break;
        AM_STAMP(3);
        // leap_frog!(..., step_size * 2^proposed) from the start point == the trial the search kept
#pragma unroll
        for (int j = 0; j < E; ++j) { x[j] = xk[j]; p[j] = pk[j]; g[j] = gk[j]; }
        const double lp_moved = lpk, ke_moved = kek;
        const bool moved_ok = okk;
        AM_STAMP(4);
        if (ap.use_mh) {
#pragma unroll
            for (int j = 0; j < E; ++j) p[j] = p[j] * -1.0;
            // log density and conditioned gradient at the proposed point: the leapfrog just computed them
            lp0 = lp_moved;
#pragma unroll
            for (int j = 0; j < E; ++j) g0[j] = g[j];
            const double h_rev = lp0 - (moved_ok ? ke_moved : kinetic());
            const int reversed = auto_step_size(lower, upper, h_rev, false);
            if (err) break;
            AM_STAMP(5);
            const bool passed = reversed == proposed;
            rev_sum += passed ? 1 : 0; rev_n += 1;
            double probability = 0.0;
            if (passed) {
                const double ex = exp(h_rev - init_joint);      // final_joint_log == log_joint at the proposed point
                probability = ex < 1.0 ? ex : (isnan(ex) ? ex : 1.0);
            }
            acc_sum += probability; acc_n += 1;
            if (!(r.rand() < probability)) {
                lp0 = lp_s;
#pragma unroll
                for (int j = 0; j < E; ++j) { x[j] = xs[j]; g0[j] = g_s[j]; }
            }
            AM_STAMP(6);
        } else {                                         // no MH step: the chain stays where the proposal leapfrog ended
            lp0 = lp_moved;
#pragma unroll
            for (int j = 0; j < E; ++j) g0[j] = g[j];
        }
    }
#ifdef PTE_PROFILE_AM
    if (lane == 0) {
        double *o = e.on_m2 + 2 * (d + 1) + 12 * cl;
        for (int k = 0; k < 8; ++k) o[k] = (double)am_prof[k];
        o[8] = (double)(__builtin_amdgcn_s_memrealtime() - am_rt0); o[9] = (double)steps_sum; o[10] = (double)steps_n; o[11] = (double)ap.n_refresh;
    }
#endif
    if (err) { if (lane == 0) set_error(e, err, (int)c, -1); return; }
#pragma unroll
    for (int j = 0; j < E; ++j) if (T.valid(j)) xrow[64 * j + lane] = x[j];
    const double S = sqr_norm_regs<E>(x);
    double l2 = 0.0, l3 = 0.0;
    if (TGT == TGT_FUNNEL) l2 = T.funnel(x, nullptr);
    if (v_on) l3 = T.variational_lp(x);
    if (lane == 0) {
        e.suff[slot] = S;
        if (TGT == TGT_FUNNEL) e.suff2[slot] = l2;
        if (v_on) e.suff3[slot] = l3;
        e.rng[2 * slot] = r.seed;
        e.expl_steps_sum[cl] += (double)steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += acc_sum;             e.expl_acc_n[cl] += acc_n;
        e.am_fac_sum[cl] += fac_sum;               e.am_fac_n[cl] += fac_n;
        e.am_rev_sum[cl] += (double)rev_sum;       e.am_rev_n[cl] += rev_n;
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, l2, l3);
}

#ifndef PTE_AM_E16_ONE_WAVE
#define PTE_AM_E16_ONE_WAVE 1
#endif
template <int E, int TGT, bool SLICE = false, bool FULL = false>
__global__ __launch_bounds__(64)
#if PTE_AM_E16_ONE_WAVE
__attribute__((amdgpu_waves_per_eu(1, (E >= 16 && !SLICE) ? 1 : 8)))      // E = 16 (d > 512): one wave per SIMD may use the whole unified register file -- spills go to AGPRs, not to scratch
#endif
void k_explore_automala(EngineDev e, AmParams ap) {
    automala_body<E, TGT, SLICE, FULL>(e, ap, blockIdx.x);
}

// (the body as a CALLED function in the scan loop: inlined, the loop's long-lived values -- the engine's ~70 pointers, the hand-shake words --
// push 32 spill reloads into every step-size search loop; called, the body keeps the register allocation of the per-scan kernel)
template <int E, int TGT, bool FULL>
__device__ __attribute__((noinline)) void automala_body_called(const EngineDev &e, const AmParams &ap, const int64_t wg) {
    automala_body<E, TGT, false, FULL>(e, ap, wg);
}

// One launch per pte_run_scans (pte_kernels.hpp "ScanLoop"; round 5): AutoMALA / MALA refreshes, then the pairwise swap hand-shake, for all
// the scans of the call.  `scan != 1` (AutoMALA.jl:87,96-102: no MH step in the first scan of a round) is decided per scan inside.
template <int E, int TGT, bool FULL>
__global__ __launch_bounds__(64) void k_scans_automala(EngineDev e, AmParams ap, ScanLoop sl) {
    const int lane = lane_id();
    const int64_t cl = am_chain_of_workgroup(e.K, blockIdx.x);
    int go = 1;
    if (lane == 0) go = scan_loop_gate(sl) ? 1 : 0;        // every workgroup of the launch is resident, or nobody starts (pte_kernels.hpp)
    if (!__builtin_amdgcn_readfirstlane(go)) return;
    for (int64_t i = 0; i < sl.n_scans; ++i) {
#ifndef PTE_TEST_NO_E_COPY
        e.trace_idx = sl.scan_idx0 + i;
#endif
        if (!ap.mala) ap.use_mh = (sl.first_scan + i != 1) ? 1 : 0;
#ifdef PTE_AM_SCANS_INLINE
        automala_body<E, TGT, false, FULL>(e, ap, blockIdx.x);
#else
        automala_body_called<E, TGT, FULL>(e, ap, blockIdx.x);
#endif
        __syncthreads();                                   // every lane's stores of the explore step happen before lane 0's release
        int slot = 0;
        if (lane == 0) slot = swap_handshake(e, sl, i, cl, e.slot_of_chain[cl]);
        slot = __builtin_amdgcn_readfirstlane(slot);
        __syncthreads();                                   // ... and lane 0's acquire before every lane's loads of the next one
        if (slot < 0) return;
    }
}

// The same loop with PTE_SCAN_WG consecutive chains per workgroup, one per wave (pte_kernels.hpp, ScanWg): three of four pairs shake hands
// through LDS.  Workgroup b holds the chain GROUP the XCD-aware dealing gives it (scan_loop_group); the stride permutation of the per-scan
// kernel does not apply (it would tear the pairs apart).
#ifndef PTE_SCAN_WG
#define PTE_SCAN_WG 4
#endif
#ifndef PTE_AM_WG_PERMUTE
#define PTE_AM_WG_PERMUTE 0          // measured at C3: 0.2000 against 0.1979-0.1986 ms per scan -- the XCD-aware dealing of consecutive groups wins
#endif
template <int E, int TGT, bool FULL>
__device__ __attribute__((noinline)) void automala_body_called_direct(const EngineDev &e, const AmParams &ap, const int64_t cl) {
    automala_body<E, TGT, false, FULL, true>(e, ap, cl);
}
template <int E, int TGT, bool FULL>
__global__ __launch_bounds__(64 * PTE_SCAN_WG) void k_scans_automala_wg(EngineDev e, AmParams ap, ScanLoop sl) {
    constexpr int NW = PTE_SCAN_WG;
    __shared__ ScanWg<NW> wg;
    __shared__ int wg_go;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < NW) wg.flag[threadIdx.x] = sl.epoch0;          // "has published every epoch up to the last call's"
    if (threadIdx.x == 0) wg_go = scan_loop_gate(sl) ? 1 : 0;        // every workgroup of the launch is resident, or nobody starts (pte_kernels.hpp)
    __syncthreads();
    if (!wg_go) return;
#if PTE_AM_WG_PERMUTE       // the per-scan kernel's stride permutation, over the GROUPS: neighbouring groups do similar work (the slow chains sit next to the reference)
    const int64_t G = (e.K + NW - 1) / NW;
    const int64_t cl = am_chain_of_workgroup(G, blockIdx.x) * NW + w;
#else
    const int64_t cl = scan_loop_group((e.K + NW - 1) / NW) * NW + w;
#endif
    if (cl >= e.K) return;
    for (int64_t i = 0; i < sl.n_scans; ++i) {
        e.trace_idx = sl.scan_idx0 + i;
        if (!ap.mala) ap.use_mh = (sl.first_scan + i != 1) ? 1 : 0;
        automala_body_called_direct<E, TGT, FULL>(e, ap, cl);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // every lane's stores of the explore step are out before lane 0 publishes
        __builtin_amdgcn_wave_barrier();
        int slot = 0;
        if (lane == 0) slot = swap_handshake<NW>(e, sl, i, cl, e.slot_of_chain[cl], &wg);
        slot = __builtin_amdgcn_readfirstlane(slot);
        asm volatile("" ::: "memory");                                      // (lane 0's acquire precedes the other lanes' loads in program order: one wave)
        __builtin_amdgcn_wave_barrier();
        if (slot < 0) return;
    }
}

// Swap statistics of every slot recomputed from the stored states (pte_set_state on an interpolated path):
// suff = sum x^2 with the fixed tree, suff2 = the funnel's log density.
template <int E>
__global__ __launch_bounds__(64) void k_refresh_funnel_stats(EngineDev e, double log3) {
    const int lane = lane_id();
    const int64_t slot = blockIdx.x;
    if (slot >= e.K) return;
    AmTarget<E, TGT_FUNNEL> T;
    T.d = e.d; T.lane = lane; T.log3 = log3;
    const double *xrow = e.x + slot * e.ld;
    double x[E];
#pragma unroll
    for (int j = 0; j < E; ++j) x[j] = T.valid(j) ? xrow[64 * j + lane] : 0.0;
    const double S = sqr_norm_regs<E>(x);
    const double l2 = T.funnel(x, nullptr);
    if (lane == 0) { e.suff[slot] = S; e.suff2[slot] = l2; }
    if (e.v_use != nullptr) {
        T.load_variational(e);
        const double l3 = T.variational_lp(x);
        if (lane == 0) e.suff3[slot] = l3;
    }
}

}  // namespace pte
