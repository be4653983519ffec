// pte_kernels.hpp -- the HIP kernels of the explore-then-swap scan loop (gfx950).
//
// Layout in HBM (struct-of-arrays over replica *slots*; a replica's state never moves inside a
// GPU, only its chain label does -- "swap betas, not states", reference src/swap/swap.jl:123-125):
//   x[slot][ld]           f64  replica state, one contiguous row per slot (coalesced 512 B per wave)
//                              Ising: the row is the lattice BIT-PACKED (site s = i*L + j -> bit s & 31 of 32-bit word s >> 5,
//                              8 KiB at L = 256; examples/ising.jl:18-22 keeps a BitMatrix too), ld = ceil(ceil(d/32) / 2)
//   rng[slot][2]          u64  SplittableRandom (seed, gamma) of the replica
//   chain_of_slot[slot]   i32  Replica.chain          slot_of_chain[chain] i32 (inverse permutation)
//   suff[slot]            f64  sufficient statistic of the state for the swap: sum_i x_i^2
//   nhp[chain], sd[chain] f64  -0.5*precision(beta_chain), sqrt(precision(beta_chain))
//   per-pair / per-chain / per-slot recorder accumulators (see EngineDev)
//
// One wavefront per replica; mapping chain -> wave so that wave c works at beta_c.
#pragma once
#include "pte_device.hpp"
#include "pte_normals.hpp"

namespace pte {

enum { ERR_NONE = 0, ERR_NAN_RATIO = 1, ERR_SLICE_SUPPORT = 2, ERR_SLICE_INVALID_LP = 3, ERR_SLICE_MAX_ITER = 4 };

struct EngineDev {
    int64_t N, d, ld;          // N: GLOBAL number of chains; ld: 8-byte words per state row
    int64_t sw;                // 8-byte words of a replica's state in a boundary message: d (f64 coordinates); Ising: ld (bit-packed spins)
    int64_t K, c0;             // this engine owns chains [c0, c0+K) and K replica slots
    int64_t *replica_id;       // [slot] global replica index (travels with the replica across shards)
    double *stat;              // [2K] SwapStat (log_ratio, uniform) of every local chain (two-phase swap)
    double *nbr_stat;          // [4]  SwapStats received from the neighbour shards: low (lr,u), high (lr,u)
    int32_t *bflag;            // [2]  boundary swap accepted: low, high
    int32_t *slot_of_chain_alt;// [K]  ping-pong target of the two-phase swap
    int32_t *ip_replica;       // [scan][slot] replica id of the slot at that scan
    double *x;
    uint64_t *rng;
    int32_t *chain_of_slot;
    int32_t *slot_of_chain;
    double *suff;
    const double *nhp;
    const double *sd;
    const double *nprec;       // [chain] -precision(beta_chain)            (AutoMALA gradient)
    const double *beta;        // [chain] Schedule.grids                     (interpolated paths)
    double *suff2;             // [slot]  second swap statistic: target log density (interpolated paths)
    double ref_nhp;            // -0.5 * precision of the normal reference of an interpolated path
    double ising_beta;         // beta of the target IsingLogPotential
    double *am_fac_sum; int64_t *am_fac_n;     // [chain] am_factors          (AutoMALA.jl:277)
    double *am_rev_sum; int64_t *am_rev_n;     // [chain] reversibility_rate  (AutoMALA.jl:294)
    // recorders (reset every round)
    double *swap_sum;  int64_t *swap_n;                    // [N-1] swap_acceptance_pr
    double *lsr_up;    double *lsr_dn;   int64_t *lsr_n;   // [N-1] log_sum_ratio (c,c+1) / (c+1,c)
    double *swap_log;                                      // null, or [max_scans][K][2] {log ratio of the lower chain's replica, of the upper's} per scan and pair, at the lower chain (PTE_RECORD_REFERENCE_REDUCTION)
    int16_t *am_log; int am_log_cap;                       // null, or [max_scans][K][am_log_cap] exponents of the step-size searches of a scan, in call order (32639 = unused): PTE_RECORD_REFERENCE_REDUCTION
    int64_t *rt_state; int64_t *rt_restarts; int64_t *rt_trips;   // [slot] round_trip
    double *expl_acc_sum; int64_t *expl_acc_n;             // [chain] explorer_acceptance_pr
    double *expl_steps_sum; int64_t *expl_steps_n;         // [chain] explorer_n_steps
    double *on_mean; double *on_m2; int64_t *on_n;         // [d+1],[d+1],[1] target-chain online stats of [state; log density]
    double *eac_log;                                       // null, or [max_scans][K][2] {log density before, after} the explore step per scan and local chain (PTE_RECORD_REFERENCE_REDUCTION with energy_ac1): k_log_energy
    double *eac; int64_t *eac_n;                           // [5K],[K] energy_ac1: Welford (mean before, mean after, C_bb, C_ba, C_aa) per local chain
    double *traces; int64_t trace_idx;                     // [max_scans][d+1] target-chain [state; log density]; row of the current scan
    // StabilizedPT (two legs, src/tempering/StabilizedPT.jl): second reference chain (-1: one leg), the two target chains
    // explore! asks for (VariationalDEO.jl:20-21) and the two swap! / round trips ask for (OddEven.jl:47-48)
    int64_t ref2, tgt_a, tgt_b, rt_tgt_a, rt_tgt_b;
    // GaussianReference of the variational leg (src/variational/GaussianReference.jl), null until activated:
    // v_use[c] != 0 <=> chain c's path starts at it; per coordinate mean, std, c0 = -0.5 log(2 pi s^2),
    // i2 = 1/(2 s^2), gf = -1/s^2 (computed on the host: same libm as the oracle); suff3[slot] = its log density
    const int32_t *v_use; const double *v_mean, *v_std, *v_c0, *v_i2, *v_gf; double *suff3;
    int compose_phase; double *lp_stash;                   // Compose(first, second): 0 single explorer, 1 first, 2 second kernel of the scan; [K] lp before the first
    int32_t *index_process;                                // [scan][slot]
    int32_t *error;                                        // [4] code, chain, coordinate, spare
    double *mw_gk;                                         // null, or [K][1024]: k_explore_langevin_mw on the funnel path keeps the kept trial's conditioned gradient here (each lane reads back what it wrote)
    unsigned int *pace;                                    // [1] refreshes begun by the workgroups of the running k_explore_langevin_mw launch (zeroed before it): wave priority by pace
    uint32_t record_flags;
    int32_t target;
    double test_swapper_pr;
};

struct SliceParams { double w; int p; int n_passes; int max_iter; };

__device__ __forceinline__ bool is_ref_chain(const EngineDev &e, int64_t c) { return (c == 0 && e.N > 1) || c == e.ref2; }
__device__ __forceinline__ bool is_tgt_chain(const EngineDev &e, int64_t c) { return c == e.tgt_a || c == e.tgt_b; }

__device__ __forceinline__ void set_error(const EngineDev &e, int code, int chain, int coord) {
    if (atomicCAS(&e.error[0], 0, code) == 0) { e.error[1] = chain; e.error[2] = coord; }
}

// ---------------------------------------------------------------------------------------------
// k_init: create_replicas (reference src/replicas/replicas.jl:87-98, src/utils/misc.jl:21-31,
// src/targets/toy_mvn_target.jl:10-11).  Replica i gets the i-th sequential split of
// SplittableRandom(seed); the split is counter based, so every wave derives its own stream.
// ---------------------------------------------------------------------------------------------
template <int NLU>
__global__ __launch_bounds__(64 * NRM_WPB) NRM_ATTR void k_init(EngineDev e, uint64_t master_seed, double init_sd) {
    __shared__ NormalsLds L;
    const int lane = lane_id();
    normals_lds_init(L, lane);
    const int64_t il = (int64_t)blockIdx.x * NRM_WPB + (NRM_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0);   // local slot: one wave per replica (wave-uniform)
    if (il >= e.K) return;
    const int64_t i = e.c0 + il;              // global replica index == initial chain
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    SeqRng r;
    r.seed = mix64(master_seed + (uint64_t)(2 * i + 1) * G);
    r.gamma = mix_gamma(master_seed + (uint64_t)(2 * i + 2) * G);
    if (e.d > 0) {
        const double S = upper_tree_root<NLU>(normals_row(L, r, e.x + il * e.ld, e.d, init_sd, lane));
        if (lane == 0) e.suff[il] = S;
    }
    if (lane == 0) {
        e.rng[2 * il] = r.seed; e.rng[2 * il + 1] = r.gamma;
        e.chain_of_slot[il] = (int32_t)i; e.slot_of_chain[il] = (int32_t)il;
        e.replica_id[il] = i;
    }
}

// i.i.d. refresh of one replica at precision sd^2 (sample_iid! / ToyExplorer.step!,
// reference src/targets/toy_mvn_target.jl:15-21, src/explorers/ToyExplorer.jl:7-12) fused with
// the evaluation of its swap statistic.
template <int NLU>
__device__ __forceinline__ double iid_refresh(const EngineDev &e, int slot, double sd, int lane,
                                              const double *wi = ZIG_WI, const unsigned long long *ki = ZIG_KI) {
    SeqRng r{e.rng[2 * slot], e.rng[2 * slot + 1]};
    double *xrow = e.x + (int64_t)slot * e.ld;
    const int B = (int)((e.d + 63) >> 6);
    double BS = 0.0;
    for (int b = 0; b < B; ++b) {
        int nl = (int)min((int64_t)64, e.d - 64 * (int64_t)b);
        double v = wave_randn_block(r, lane, nl, wi, ki) / sd;
        if (lane < nl) xrow[64 * b + lane] = v; else v = 0.0;
        double s = wave_sum_dpp(v * v);                     // same tree as wave_tree_sum64, DPP instead of LDS permutes
        if (lane == b) BS = s;
    }
    double S = upper_tree_root<NLU>(BS);
    if (lane == 0) { e.suff[slot] = S; e.rng[2 * slot] = r.seed; }
    return S;
}

// coordinate i of the state of `slot` as the reference's sample sees it (Ising: spin 0.0 / 1.0 out of the bit-packed row)
__device__ __forceinline__ double state_value(const EngineDev &e, int slot, int64_t i) {
    const double *xrow = e.x + (int64_t)slot * e.ld;
    if (e.target == 3) return (double)((reinterpret_cast<const uint32_t *>(xrow)[i >> 5] >> (i & 31)) & 1u);
    return xrow[i];
}

// Target-chain online statistics (reference src/pt/pigeons.jl:110-115,
// src/recorders/OnlineStateRecorder.jl:87-96): per-coordinate Welford mean / M2 of the recorded sample
// extract_sample(state::Array, lp) = [state; lp(state)] (src/pt/state.jl:79), so d + 1 entries.
// `which`: 0 / 1 = first / second target chain (two legs: one accumulator set each, merged on the host)
__device__ __forceinline__ void record_online(const EngineDev &e, int slot, int lane, double lp, int which) {
    double *mean = e.on_mean + which * (e.d + 1), *m2 = e.on_m2 + which * (e.d + 1);
    int64_t n = e.on_n[which] + 1;
    for (int64_t i = lane; i <= e.d; i += 64) {
        double v = (i < e.d) ? state_value(e, slot, i) : lp, mu = mean[i];
        double mu2 = mu + (v - mu) / (double)n;
        m2[i] += (v - mu) * (v - mu2);
        mean[i] = mu2;
    }
    if (lane == 0) e.on_n[which] = n;
}

// log_potentials[chain](state) from the swap statistics: S = sum x^2 (Ising: sum_pair_products), l2 = the
// target log density of an interpolated path.  Same expressions as swap_log_ratio below.
__device__ __forceinline__ bool chain_uses_variational(const EngineDev &e, int64_t c) { return e.v_use != nullptr && e.v_use[c] != 0; }
__device__ __forceinline__ double chain_lp(const EngineDev &e, int64_t c, double S, double l2, double l3 = 0.0) {
    if (e.target == 2 || e.target == 3) {
        const double ref = (e.target == 2) ? (chain_uses_variational(e, c) ? l3 : e.ref_nhp * S) : 0.0 * S;
        const double tgt = (e.target == 2) ? l2 : e.ising_beta * S;
        const double b = e.beta[c];
        return b == 0.0 ? ref : (b == 1.0 ? tgt : (1.0 - b) * ref + b * tgt);
    }
    return e.nhp[c] * S;
}
// explore!'s bookkeeping around the explorer (reference src/pt/pigeons.jl:101-143):
//   before = eval_if_ac_requested  -> lp_before_explore at kernel entry (from the statistics of the last scan)
//   process_ac! / :online / :traces -> record_after_explore at kernel exit, all lanes, S and l2 in registers
// Compose(first, second) (src/explorers/Compose.jl:16-19) runs two explorer kernels per scan: the first
// stashes `before`, only the second records; the reference chain is refreshed (and recorded) by the first.
__device__ __forceinline__ double lp_before_explore(const EngineDev &e, int64_t c, int slot) {
    if (!(e.record_flags & 16u)) return 0.0;
    if (e.compose_phase == 2) return e.lp_stash[c - e.c0];
    return chain_lp(e, c, e.suff[slot], e.suff2[slot], e.v_use ? e.suff3[slot] : 0.0);
}
// PTE_RECORD_REFERENCE_REDUCTION with energy_ac1: the pair (log density before, after explore!) of every chain and scan, logged by a launch of its own
// before and after the scan's explorer kernel(s) -- the expressions of lp_before_explore / record_after_explore_impl on the statistics in memory, so the
// same bits -- and replayed by pte_reduce with OnlineStats' CovMatrix arithmetic.  (Not written by the explorer kernels themselves: one more pointer and
// store in their epilogues moved the register allocation of k_explore_toy -- 12 -> 28 B of scratch, +1.2 % -- for a diagnostic nobody adapts on.)
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_log_energy(EngineDev e, int which) {
    const int64_t cl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    e.eac_log[(e.trace_idx * e.K + cl) * 2 + which] = chain_lp(e, c, e.suff[slot], e.suff2[slot], e.v_use ? e.suff3[slot] : 0.0);
}
#endif

__device__ __forceinline__ void record_after_explore_impl(const EngineDev &e, int64_t cl, int64_t c, int slot, int lane,
                                                          double lp_before, double S, double l2, double l3 = 0.0) {
    const unsigned f = e.record_flags;
    if (!(f & (4u | 8u | 16u))) return;
    const double lp = chain_lp(e, c, S, l2, l3);
    if ((f & 16u) && lane == 0) {                         // energy_ac1: (chain, SVector(before, after)) -> CovMatrix(2)
        double *o = e.eac + 5 * cl;
        const int64_t n = e.eac_n[cl] + 1;
        const double db = lp_before - o[0], da = lp - o[1];
        const double mb = o[0] + db / (double)n, ma = o[1] + da / (double)n;
        o[2] += db * (lp_before - mb); o[3] += db * (lp - ma); o[4] += da * (lp - ma);
        o[0] = mb; o[1] = ma; e.eac_n[cl] = n;
    }
    const bool tgt = is_tgt_chain(e, c);
    const int which = (c == e.tgt_b && e.tgt_b != e.tgt_a) ? 1 : 0;
    if (tgt && (f & 4u)) {
        __threadfence_block();
        record_online(e, slot, lane, lp, which);
    }
    if ((f & 8u) && (tgt || (f & 32u))) {                  // traces[(chain, scan)] = [state; lp]; 32: inputs.extended_traces
        __threadfence_block();
        const int64_t ntgt = (e.tgt_b != e.tgt_a) ? 2 : 1;
        double *row = e.traces + ((f & 32u) ? (e.trace_idx * e.K + cl) : (e.trace_idx * ntgt + which)) * (e.d + 1);
        for (int64_t i = lane; i < e.d; i += 64) row[i] = state_value(e, slot, i);
        if (lane == 0) row[e.d] = lp;
    }
}
__device__ __forceinline__ void record_after_explore(const EngineDev &e, int64_t cl, int64_t c, int slot, int lane,
                                                     double lp_before, double S, double l2, double l3 = 0.0) {
    if (e.compose_phase == 1) { if ((e.record_flags & 16u) && lane == 0) e.lp_stash[cl] = lp_before; return; }
    record_after_explore_impl(e, cl, c, slot, lane, lp_before, S, l2, l3);
}

// the same at a chain of the MVN path, with explore!'s recorders around it
template <int NLU>
__device__ __forceinline__ void iid_refresh_recorded(const EngineDev &e, int64_t cl, int64_t c, int slot, double sd, int lane,
                                                     const double *wi = ZIG_WI, const unsigned long long *ki = ZIG_KI) {
    if (e.compose_phase == 2) return;                     // the first explorer's kernel already did
    const double lp0 = lp_before_explore(e, c, slot);
    const double S = iid_refresh<NLU>(e, slot, sd, lane, wi, ki);
    record_after_explore_impl(e, cl, c, slot, lane, lp0, S, 0.0);
}

// ---------------------------------------------------------------------------------------------
// k_explore_toy: explore! with ToyExplorer -- every chain is refreshed i.i.d. at its own precision.
// HBM-write bound: 8d bytes stored per replica.
// ---------------------------------------------------------------------------------------------
template <int NLU>
__global__ __launch_bounds__(64 * NRM_WPB) NRM_ATTR void k_explore_toy(EngineDev e) {
    __shared__ NormalsLds L;                              // ziggurat tables, the chunk's values by stream position, event list, block sums
    const int lane = lane_id();
    normals_lds_init(L, lane);
    const int64_t cl = (int64_t)blockIdx.x * NRM_WPB + (NRM_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0);   // wave-uniform
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    if (e.compose_phase == 2) return;                     // the first explorer's kernel already did
    const double lp0 = lp_before_explore(e, c, slot);
    SeqRng r{e.rng[2 * slot], e.rng[2 * slot + 1]};
    const double S = upper_tree_root<NLU>(normals_row(L, r, e.x + (int64_t)slot * e.ld, e.d, e.sd[c], lane));
    if (lane == 0) { e.suff[slot] = S; e.rng[2 * slot] = r.seed; }
    record_after_explore_impl(e, cl, c, slot, lane, lp0, S, 0.0);
}

// ---------------------------------------------------------------------------------------------
// k_explore_slice: explore! with SliceSampler (reference src/explorers/SliceSampler.jl:24-237) on
// the scaled-precision MVN path.  Chain 0 is refreshed i.i.d. (src/pt/pigeons.jl:104-105).
//
// The reference re-evaluates the full O(d) log potential after every single-coordinate change.
// Here sum(abs2, x) is a FIXED binary tree; changing leaf c only changes the log2(P) nodes on the
// path leaf->root, so lp(x with x_c = v) = nhp * (((v^2 + s_0) + s_1) + ... ) where s_k is the
// sibling subtree at level k: bit-identical to the full recompute, O(log d) instead of O(d).
// The wave keeps the siblings in registers:
//   levels 0..5  (inside the current 64-coordinate block, lane l <-> coordinate 64b+l):
//       right siblings come from the butterfly U[k] taken at block start (unchanged this pass),
//       left siblings are the chain intermediates of the coordinate that completed that subtree;
//   levels 6..   (whole blocks): from the butterfly over the block sums BS (lane b <-> block b).
// The 64 lanes also pre-evaluate the next 64 raw draws of the replica's counter-based stream.
// ---------------------------------------------------------------------------------------------
template <int NLU>
struct SliceCoord {
    static constexpr int NL = 6 + NLU;
    double sib[NL];
    double nhp, z, w;
    int lane;
    __device__ __forceinline__ double evalS(double v) const {
        double t = v * v;
#pragma unroll
        for (int k = 0; k < NL; ++k) t = t + sib[k];
        return t;
    }
    __device__ __forceinline__ double evalS(double v, double (&t)[NL + 1]) const {
        t[0] = v * v;
#pragma unroll
        for (int k = 0; k < NL; ++k) t[k + 1] = t[k] + sib[k];
        return t[NL];
    }
    // slice_accept, SliceSampler.jl:192-237.  Returns accept; `evals` of lp at the bisected endpoints.
    __device__ __forceinline__ bool accept(double old_position, double new_position, double L, double R,
                                           double lp_L, double lp_R, double &acc_sum, int64_t &acc_n) const {
        double Lhat = L, Rhat = R;
        bool Rstale = false, Lstale = false, D = false;
        while (Rhat - Lhat > 1.1 * w) {
            double M = (Lhat + Rhat) / 2.0;
            if ((old_position < M && new_position >= M) || (old_position >= M && new_position < M)) D = true;
            if (new_position < M) { Rhat = M; Rstale = true; }
            else { Lhat = M; Lstale = true; }
            if (D) {
                if (Lstale) { lp_L = nhp * evalS(Lhat); Lstale = false; }
                if (Rstale) { lp_R = nhp * evalS(Rhat); Rstale = false; }
                if (z >= lp_L && z >= lp_R) { acc_n += 1; return false; }
            }
        }
        acc_sum += 1.0; acc_n += 1;
        return true;
    }
};

__device__ __forceinline__ bool jl_isapprox(double x, double y) {
    if (x == y) return true;
    if (!isfinite(x) || !isfinite(y)) return false;
    double ax = fabs(x), ay = fabs(y);
    return fabs(x - y) <= 1.4901161193847656e-8 * (ax > ay ? ax : ay);
}

template <int NLU>
__global__ __launch_bounds__(64) void k_explore_slice(EngineDev e, SliceParams sp) {
    constexpr int NL = 6 + NLU;
    const int lane = lane_id();
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
    SliceCoord<NLU> sc;
    sc.nhp = e.nhp[c]; sc.w = sp.w; sc.lane = lane;

    // cached_log_potential (SliceSampler.jl:32-41): full evaluation once per step
    double BS = 0.0;
    for (int b = 0; b < B; ++b) {
        int64_t i = 64 * (int64_t)b + lane;
        double v = (i < d) ? xrow[i] : 0.0;
        double s = wave_tree_sum64(v * v);
        if (lane == b) BS = s;
    }
    double S = upper_tree_root<NLU>(BS);
    double lp = sc.nhp * S;
    if (lp == -INFINITY) { if (lane == 0) set_error(e, ERR_SLICE_SUPPORT, (int)c, -1); return; }

    WaveDraws dr;
    dr.init(e.rng[2 * slot], e.rng[2 * slot + 1], lane);
    double steps_sum = 0.0, acc_sum = 0.0;
    int64_t steps_n = 0, acc_n = 0;
    bool failed = false;

    for (int pass = 0; pass < sp.n_passes && !failed; ++pass) {
        for (int b = 0; b < B && !failed; ++b) {
            const int64_t base = 64 * (int64_t)b;
            const int nl = (int)min((int64_t)64, d - base);
            double X = (lane < nl) ? xrow[base + lane] : 0.0;
            double U[7];
            butterfly6(X * X, U);
            {   // siblings above the block: butterfly over the block sums
                double V = BS;
#pragma unroll
                for (int q = 0; q < NLU; ++q) {
                    sc.sib[6 + q] = readlane_f64(V, b ^ (1 << q));
                    V = V + shfl_xor_f64(V, 1 << q);
                }
            }
            double tsave[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int l = 0; l < nl; ++l) {
                // ---- siblings inside the block for coordinate l
                if (l == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) sc.sib[k] = readlane_f64(U[k], 1 << k);
                } else {
                    const int r = __builtin_ctz((unsigned)l);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (k < r) sc.sib[k] = readlane_f64(U[k], l ^ (1 << k));
                        else if (k == r) sc.sib[k] = tsave[k];
                    }
                }
                const double xold = readlane_f64(X, l);
                // ---- slice_sample_coord! (SliceSampler.jl:89-95)
                double E;
                {
                    uint64_t raw = dr.next_raw(lane);
                    uint64_t ri = raw & MASK52;
                    int idx = (int)(ri & 0xFF);
                    E = (double)ri * ZIG_WE[idx];
                    if (!(ri < ZIG_KE[idx])) {
                        SeqRng s = dr.to_seq();
                        E = randexp_from_raw(s, raw);
                        dr.from_seq(s, lane);
                    }
                }
                sc.z = lp - E;
                // slice_double (:97-126) with initialize_slice_endpoints (:129-133)
                double L = xold - sp.w * dr.rand(lane);
                double R = L + sp.w;
                int K = sp.p;
                double lp_L = sc.nhp * sc.evalS(L);
                double lp_R = sc.nhp * sc.evalS(R);
                while (K > 0 && (sc.z < lp_L || sc.z < lp_R)) {
                    double V = dr.rand(lane);
                    if (V <= 0.5) { L = L - (R - L); lp_L = sc.nhp * sc.evalS(L); }
                    else { R = R + (R - L); lp_R = sc.nhp * sc.evalS(R); }
                    K -= 1;
                }
                steps_sum += (double)(sp.p - K); steps_n += 1;
                // slice_shrink! (:144-186)
                double Lbar = L, Rbar = R;
                double t[NL + 1];
                double xf = xold;
                bool done = false;
                for (int n = 1; n <= sp.max_iter; ++n) {
                    double newpos = Lbar + dr.rand(lane) * (Rbar - Lbar);
                    double Snew = sc.evalS(newpos, t);
                    double newlp = sc.nhp * Snew;
                    bool consider = sc.z < newlp;
                    if (consider && sc.accept(xold, newpos, L, R, lp_L, lp_R, acc_sum, acc_n)) {
                        xf = newpos; S = Snew; lp = newlp;
                        steps_sum += (double)n; steps_n += 1;
                        done = true;
                        break;
                    }
                    if (newpos < xold) Lbar = newpos; else Rbar = newpos;
                    if (jl_isapprox(Lbar, Rbar)) {
                        S = sc.evalS(xold, t); lp = sc.nhp * S;
                        steps_sum += (double)n; steps_n += 1;
                        done = true;
                        break;
                    }
                }
                if (!done) { if (lane == 0) set_error(e, ERR_SLICE_MAX_ITER, (int)c, (int)(base + l)); failed = true; break; }
                if (!isfinite(lp)) { if (lane == 0) set_error(e, ERR_SLICE_INVALID_LP, (int)c, (int)(base + l)); failed = true; break; }
                if (lane == l) X = xf;
#pragma unroll
                for (int k = 0; k < 7; ++k) tsave[k] = t[k];
            }
            if (lane < nl) xrow[base + lane] = X;
            if (lane == b) BS = tsave[6];
        }
    }
    if (lane == 0) {
        e.suff[slot] = S;
        e.rng[2 * slot] = dr.final_seed();
        e.expl_steps_sum[cl] += steps_sum; e.expl_steps_n[cl] += steps_n;
        e.expl_acc_sum[cl] += acc_sum;     e.expl_acc_n[cl] += acc_n;
    }
    record_after_explore(e, cl, c, slot, lane, lp_before, S, 0.0);
}

// ---------------------------------------------------------------------------------------------
// k_swap: communicate! (reference src/pt/pigeons.jl:64-69, src/swap/swap.jl:6-39,106-126,
// src/swap/pair_swapper.jl:42-88, src/swap/OddEven.jl:23-31, src/swap/DEO.jl:12).
// Thread t handles chain (t + off) mod N with off = 1 on the even graph, so that the two chains
// of every DEO pair sit in adjacent lanes (t, t^1) of one wavefront: the pair's SwapStats are
// exchanged with __shfl_xor and both lanes take the same decision.
// ---------------------------------------------------------------------------------------------
// log_unnormalized_ratio(log_potentials, partner, mine, state) (src/log_potentials/log_potentials.jl:43-51)
// from the swap statistics the explorer left behind.  target 0: ScaledPrecisionNormalPath;
// target 2: InterpolatedLogPotential with its beta == 0 / 1 short-circuits (InterpolatedLogPotential.jl:9-16).
__device__ __forceinline__ double swap_log_ratio(const EngineDev &e, int slot, int64_t c, int64_t pc) {
    const double S = e.suff[slot];
    if (e.target == 2) {
        const double tgt = e.suff2[slot];
        const double refn = chain_uses_variational(e, pc) ? e.suff3[slot] : e.ref_nhp * S;   // each chain's own reference end
        const double refd = chain_uses_variational(e, c) ? e.suff3[slot] : e.ref_nhp * S;
        const double bn = e.beta[pc], bd = e.beta[c];
        const double num = bn == 0.0 ? refn : (bn == 1.0 ? tgt : (1.0 - bn) * refn + bn * tgt);
        const double den = bd == 0.0 ? refd : (bd == 1.0 ? tgt : (1.0 - bd) * refd + bd * tgt);
        return num - den;
    }
    if (e.target == 3) {     // Ising: S holds sum_pair_products; ref = 0.0 * spp, target = beta_target * spp (examples/ising.jl:74)
        const double ref = 0.0 * S, tgt = e.ising_beta * S;
        const double bn = e.beta[pc], bd = e.beta[c];
        const double num = bn == 0.0 ? ref : (bn == 1.0 ? tgt : (1.0 - bn) * ref + bn * tgt);
        const double den = bd == 0.0 ? ref : (bd == 1.0 ? tgt : (1.0 - bd) * ref + bd * tgt);
        return num - den;
    }
    return e.nhp[pc] * S - e.nhp[c] * S;
}

__device__ __forceinline__ double dev_logaddexp(double x, double y) {
    double delta = (x == y) ? 0.0 : fabs(x - y);
    double m = (x > y) ? x : y;
    double nd = -delta;
    double t = (nd <= -37.0) ? exp(nd) : log1p(exp(nd));
    return m + t;
}

#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_swap(EngineDev e, int even, int64_t scan_idx) {
    const int64_t N = e.N;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t < N;
    const int off = even ? 1 : 0;
    const int64_t c = valid ? (t + off) % N : 0;
    // partner_chain (OddEven.jl:23-31), 0-based
    int64_t pc = c;
    if (valid) {
        const bool chain_even = ((c + 1) % 2 == 0);
        int64_t proposed = (c + 1) + ((chain_even == (even != 0)) ? 1 : -1);
        pc = (proposed == 0) ? 0 : (proposed == N + 1 ? N - 1 : proposed - 1);
    }
    double lr = 0.0, u = 0.0;
    int slot = 0;
    if (valid) {
        slot = e.slot_of_chain[c];
        if (e.target != 1) {
            lr = swap_log_ratio(e, slot, c, pc);
            if (isnan(lr)) set_error(e, ERR_NAN_RATIO, (int)c, -1);
        }
        uint64_t seed = e.rng[2 * slot] + e.rng[2 * slot + 1];    // one rand(replica.rng) per replica
        e.rng[2 * slot] = seed;
        u = u52_to_unit(mix64(seed));
        if (e.record_flags & 2u) {
            e.index_process[scan_idx * N + slot] = (int32_t)c;
            e.ip_replica[scan_idx * N + slot] = (int32_t)e.replica_id[slot];
        }
        if (e.record_flags & 1u) {     // RoundTripRecorder.jl:43-54
            const bool is_ref = is_ref_chain(e, c), is_tgt = (c == e.rt_tgt_a || c == e.rt_tgt_b);
            int64_t st = e.rt_state[slot];
            if (st == 0 && is_ref) e.rt_state[slot] = 1;
            else if (st == 1 && is_tgt) { e.rt_state[slot] = 2; e.rt_restarts[slot] += 1; }
            else if (st == 2 && is_ref) { e.rt_state[slot] = 1; e.rt_trips[slot] += 1; }
        }
    }
    const double lr_p = __shfl_xor(lr, 1, 64);
    const double u_p = __shfl_xor(u, 1, 64);
    if (valid && pc != c) {
        const bool lower = c < pc;
        const double uu = lower ? u : u_p;
        bool do_swap;
        if (e.target == 1) {
            do_swap = uu < e.test_swapper_pr;                  // TestSwapper, pair_swapper.jl:135-138
        } else {
            const double ex = exp(lr + lr_p);
            const double alpha = ex < 1.0 ? ex : 1.0;         // swap_acceptance_probability :88
            do_swap = uu < alpha;                               // swap_decision :81-85
            if (lower) {                                        // record_swap_stats! :59-66
                e.swap_sum[c] += alpha; e.swap_n[c] += 1;
                e.lsr_up[c] = dev_logaddexp(e.lsr_up[c], lr);
                e.lsr_dn[c] = dev_logaddexp(e.lsr_dn[c], lr_p);
                e.lsr_n[c] += 1;
                if (e.swap_log) { double *w = e.swap_log + (scan_idx * N + c) * 2; w[0] = lr; w[1] = lr_p; }
            }
        }
        if (do_swap) { e.chain_of_slot[slot] = (int32_t)pc; e.slot_of_chain[pc] = slot; }
    }
}
#endif


// ---------------------------------------------------------------------------------------------
// Two-phase swap for chain-sharded engines (rank g owns chains [c0, c0+K)): the pair (c0-1, c0) /
// (c0+K-1, c0+K) has its two replicas on different GPUs, so the SwapStats of the boundary chains
// are exchanged between the phases (16 B each way), and -- iff the swap is accepted -- the two
// replicas' payloads {state, sum x^2, rng, replica id, round-trip state} trade places.
// Same per-replica arithmetic and the same single rand() per replica as k_swap.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t deo_partner(int64_t N, int even, int64_t c) {
    const bool chain_even = ((c + 1) % 2 == 0);
    int64_t proposed = (c + 1) + ((chain_even == (even != 0)) ? 1 : -1);
    return (proposed == 0) ? 0 : (proposed == N + 1 ? N - 1 : proposed - 1);
}

// ---------------------------------------------------------------------------------------------
// One launch per pte_run_scans (round 5): the scan loop `while next_scan!(pt)` (reference src/pt/pigeons.jl:46-55) inside ONE kernel
// for shapes whose workgroups are all resident -- k_scans_* (pte_slice8.hpp).  Workgroup c holds chain c for the whole call; after its
// explore step it takes part in the DEO swap of ITS pair only: communicate! needs nothing of the other N - 2 chains (swap.jl:6-26,
// pair_swapper.jl:42-88: a pair's decision is a function of the two SwapStats), so there is no grid-wide barrier -- each wave publishes
// {log ratio, uniform, slot} with a release store of its epoch and waits for its PARTNER'S epoch.  A launch is then no longer as long as
// the slowest of N waves per scan: waves wait for neighbours only and fluctuations average out along the ladder (tools/sim_pairsync.py
// replays recorded per-wave durations: x1.06 at the metric shape, where a device-scope barrier would give x1.014).
//   * same arithmetic, same single rand(replica.rng), same recorder updates as k_swap: bit-identical (tests compare the two paths);
//   * everything a replica owns (state row, rng, sum x^2, round-trip state) is written BEFORE the release store; the new holder reads it
//     after its acquire -- agent scope, i.e. across the XCDs' L2s;
//   * every chain publishes every scan (also the chain a graph leaves idle), so "partner has published epoch - 1" -- the condition under
//     which the publish buffer of this parity, read by the same partner two scans ago, may be overwritten -- always becomes true;
//   * every wait has a time-out (3 s on the 100 MHz clock): a workgroup that is not resident after all ends the call with an error
//     instead of hanging the GPU.
// ---------------------------------------------------------------------------------------------
enum { ERR_HANDSHAKE_TIMEOUT = 9 };
struct ScanLoop {
    int64_t first_scan, n_scans;       // scan numbers first_scan .. first_scan + n_scans - 1 (DEO parity = iseven(scan); AutoMALA's scan == 1 rule)
    int64_t scan_idx0;                 // scans of this round already run: row of index_process / traces the first scan writes
    unsigned long long epoch0;         // hand-shake epochs of this call: epoch0 + 1 .. epoch0 + n_scans (monotone over the engine's life; re-based by pte_set_state only)
    unsigned long long *flag;          // [K] the last epoch chain c has published
    double *pub;                       // [K][2][4] {log ratio, uniform, slot, -} of chain c, double-buffered by the epoch's parity
    // the residency gate (round 6, scan_loop_gate below): gate[0] = workgroups of scan-loop launches that have arrived so far (monotone),
    // gate[1] = the decision word, (launch number << 1) | aborted
    unsigned long long *gate;
    unsigned long long gate_seq;       // this launch's number, 1, 2, ... (monotone over the engine's life)
    unsigned long long gate_target;    // gate[0] once every workgroup of this launch has arrived
    int test_fault;                    // 0; test build only (PTE_KERNEL_TEST_*): 1 = the wave of chain 7 dies before its third publish, 2 = workgroup 3 arrives 80 ms late
};

// ---------------------------------------------------------------------------------------------
// The residency gate: forward progress of the scan loop, ENFORCED (round 6; the reference's loop, src/pt/pigeons.jl:46-55, cannot hang).
// The hand-shakes below spin, so every workgroup of the launch must be on the device at once.  Rounds 5 inferred that from an occupancy query,
// which knows nothing of another engine, stream or process holding compute units.  Now no workgroup touches anything before ALL of them have
// ARRIVED: each adds one to gate[0]; the one that completes the count proposes "go", a workgroup that has waited PTE_GATE_TICKS (50 ms on the
// 100 MHz clock) proposes "abort"; whichever compare-and-swap on the decision word lands first decides for the whole launch, late arrivals
// included -- so all workgroups agree.  go: every workgroup holds its wave slots until the kernel ends (a resident wave is not descheduled in
// favour of another queue's work), the spins cannot starve.  abort: every workgroup returns at once, NOTHING has been written (states, streams,
// recorders, epochs as before the call) and the host runs the same scans as explore + swap launches (pte.hip, run_scans_fused).
// One relaxed atomic add + a few polls per workgroup and launch: 2-4 us at 1024 workgroups (profiles/r06_gate_cost.txt).
// ---------------------------------------------------------------------------------------------
#ifndef PTE_GATE_TICKS
#define PTE_GATE_TICKS 5000000ull
#endif
__device__ __forceinline__ bool scan_loop_gate_decide(unsigned long long *dec, unsigned long long seq, unsigned long long proposal) {
    unsigned long long cur = __hip_atomic_load(dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((cur >> 1) < seq) {      // nobody has decided this launch yet
        if (__hip_atomic_compare_exchange_strong(dec, &cur, proposal, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { cur = proposal; break; }
    }
    return cur == (seq << 1);
}
// called by ONE lane per workgroup; true = go
__device__ __forceinline__ bool scan_loop_gate(const ScanLoop &sl) {
    unsigned long long *cnt = sl.gate, *dec = sl.gate + 1;
    const unsigned long long go = sl.gate_seq << 1;
#ifdef PTE_TEST_KERNELS
    if (sl.test_fault == 2 && blockIdx.x == 3) {           // a workgroup the dispatcher could not place for 80 ms
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 8000000ull) __builtin_amdgcn_s_sleep(64);
    }
#endif
    const unsigned long long before = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1ull == sl.gate_target) return scan_loop_gate_decide(dec, sl.gate_seq, go);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long cur = __hip_atomic_load(dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur >> 1) >= sl.gate_seq) return cur == go;
        if (__builtin_amdgcn_s_memrealtime() - t0 > PTE_GATE_TICKS) return scan_loop_gate_decide(dec, sl.gate_seq, go | 1ull);
        __builtin_amdgcn_s_sleep(4);
    }
}

// Which chain a workgroup of the scan loop holds.  Observed (not promised by HIP): consecutive workgroups are dealt round-robin over the 8
// XCDs, workgroup b runs on XCD b mod 8, and each XCD has its own L2.  Dealing the chains out the other way round -- workgroup b holds chain
// (b mod 8) * K/8 + b / 8 -- puts K/8 consecutive chains behind one L2, so a replica handed to the neighbouring chain is usually read through
// the L2 it was written in (the guide: same-XCD placement is a speed bonus, never correctness -- the hand-shake below is the agent-scope one
// for every pair).
__device__ __forceinline__ int64_t scan_loop_chain(int64_t K) {
    const int64_t b = blockIdx.x, g = b & 7, j = b >> 3, q = K >> 3, r = K & 7;
    return g * q + (g < r ? g : r) + j;
}

// ... and which GROUP of consecutive chains, when a workgroup holds several (G groups): the same dealing over the groups
__device__ __forceinline__ int64_t scan_loop_group(int64_t G) {
    const int64_t b = blockIdx.x, g = b & 7, j = b >> 3, q = G >> 3, r = G & 7;
    return g * q + (g < r ? g : r) + j;
}

// Wait for *p >= want.  false: give up -- 3 s have passed, or ANOTHER chain has already failed (the engine's error word is polled every
// 16th spin, round 6: a broken launch ends when its first wave gives up, not after every blocked wave's own 3 s).
__device__ __forceinline__ bool hs_wait(const unsigned long long *p, unsigned long long want, const int32_t *err) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 1;; ++spin) {
#ifndef PTE_HS_MEASURE_NO_SLEEP
        __builtin_amdgcn_s_sleep(8);
#endif
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((spin & 15u) == 0u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) return false;
        }
    }
}

// lane 0 of the wave that holds chain c (world_size == 1: local index == chain) after the explore step of scan number sl.first_scan + i:
// swap_stat + the replica's recorders + hand-shake + decision.  Returns the slot chain c holds afterwards, -1 after a time-out.
// NW > 1 (round 5, late): a workgroup of the scan loop holds NW CONSECUTIVE chains, one per wave, so NW - 1 of every NW pairs have both
// waves on ONE compute unit: they shake hands through the workgroup's LDS (`ScanWg`) under workgroup-scope fences -- the waves of a workgroup
// share their CU's vector L1, nothing has to be written back or invalidated -- and only the pairs that straddle two workgroups pay the
// agent-scope hand-off below.  Every chain keeps BOTH of its flags (LDS and global) current every scan: the partner of the next scan looks
// at the one it shares with this chain.
template <int NW> struct ScanWg { unsigned long long flag[NW]; double pub[NW][2][4]; };
__device__ __forceinline__ bool hs_wait_lds(unsigned long long *p, unsigned long long want, const int32_t *err) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return true;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 1;; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return true;
        if ((spin & 63u) == 0u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) return false;
        }
    }
}
template <int NW = 1>
__device__ __forceinline__ int swap_handshake(const EngineDev &e, const ScanLoop &sl, int64_t i, int64_t c, int slot, ScanWg<NW> *wg = nullptr) {
    const int64_t N = e.N;
    const int even = ((sl.first_scan + i) % 2 == 0) ? 1 : 0;              // create_swap_graph(::DEO), DEO.jl:12
    const int64_t scan_idx = sl.scan_idx0 + i;
    const unsigned long long epoch = sl.epoch0 + 1ull + (unsigned long long)i;
    const int64_t pc = deo_partner(N, even, c);
    if (__hip_atomic_load(e.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;   // a chain has failed: the call is lost, leave (round 6)
#ifdef PTE_TEST_KERNELS
    if (sl.test_fault == 1 && c == 7 && i == 2) return -1;                // test build: this wave dies silently -- its partner must time out, everybody else must leave early
#endif
    double lr = 0.0;
    if (e.target != 1) {
        lr = swap_log_ratio(e, slot, c, pc);
        if (isnan(lr)) set_error(e, ERR_NAN_RATIO, (int)c, -1);
    }
    uint64_t seed = e.rng[2 * slot] + e.rng[2 * slot + 1];               // one rand(replica.rng) per replica
    e.rng[2 * slot] = seed;
    const double u = u52_to_unit(mix64(seed));
    if (e.record_flags & 2u) {
        e.index_process[scan_idx * N + slot] = (int32_t)c;
        e.ip_replica[scan_idx * N + slot] = (int32_t)e.replica_id[slot];
    }
    if (e.record_flags & 1u) {     // RoundTripRecorder.jl:43-54
        const bool is_ref = is_ref_chain(e, c), is_tgt = (c == e.rt_tgt_a || c == e.rt_tgt_b);
        int64_t st = e.rt_state[slot];
        if (st == 0 && is_ref) e.rt_state[slot] = 1;
        else if (st == 1 && is_tgt) { e.rt_state[slot] = 2; e.rt_restarts[slot] += 1; }
        else if (st == 2 && is_ref) { e.rt_state[slot] = 1; e.rt_trips[slot] += 1; }
    }
    if (pc == c) {                                                          // idle on this graph: publish the epoch only (nobody reads this replica before the next scan's hand-shake)
        __hip_atomic_store(&sl.flag[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (NW > 1) __hip_atomic_store(&wg->flag[c % NW], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return slot;
    }
    double lr_p, u_p; int slot_p;
    bool near = false;
    if constexpr (NW > 1) near = (pc / NW == c / NW);
    if (near) {
        if constexpr (NW > 1) {
            const int w = (int)(c % NW), pw = (int)(pc % NW), par = (int)(epoch & 1ull);
            if (!hs_wait_lds(&wg->flag[pw], epoch - 1ull, e.error)) { set_error(e, ERR_HANDSHAKE_TIMEOUT, (int)c, -1); return -1; }
            __hip_atomic_store(&wg->pub[w][par][0], lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&wg->pub[w][par][1], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&wg->pub[w][par][2], (double)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // this wave's row / statistics stores have reached the L2 (through the L1 both waves share)
            __hip_atomic_store(&wg->flag[w], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&sl.flag[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for the partner of the next scan, if it lives in another workgroup)
            if (!hs_wait_lds(&wg->flag[pw], epoch, e.error)) { set_error(e, ERR_HANDSHAKE_TIMEOUT, (int)c, -1); return -1; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            lr_p = __hip_atomic_load(&wg->pub[pw][par][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            u_p = __hip_atomic_load(&wg->pub[pw][par][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            slot_p = (int)__hip_atomic_load(&wg->pub[pw][par][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
    double *mine = sl.pub + ((c * 2 + (int64_t)(epoch & 1ull)) * 4);
    if (!hs_wait(&sl.flag[pc], epoch - 1ull, e.error)) { set_error(e, ERR_HANDSHAKE_TIMEOUT, (int)c, -1); return -1; }
    // {log ratio, uniform, slot}: device-coherent (sc1) stores / loads.  Then the hand-off proper, the form MI355X_MICROARCH.md
    // ("Workgroup dispatch, XCD placement & inter-workgroup visibility") prescribes for plain payload stores: producer = agent-scope release
    // (buffer_wbl2 sc1: the XCD's dirty L2 lines -- the state row this wave has just written -- reach memory) + an explicit s_waitcnt vmcnt(0)
    // the compiler cannot drop + relaxed agent flag store; consumer = relaxed polls, ONE agent-scope acquire (buffer_inv sc1: this CU's vector
    // L1), then plain loads.  Workgroup scope / buffer_inv sc0 is NOT an acquire for another CU's data, same XCD or not: a build that took
    // that shortcut for same-XCD pairs read stale sum x^2 words at once (round 5).
    __hip_atomic_store(&mine[0], lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine[1], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine[2], (double)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PTE_HS_MEASURE_NO_FENCE     // measurement builds only (not coherent: wrong results): what the L2 write-back / L1 invalidate cost
    __hip_atomic_store(&sl.flag[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!hs_wait(&sl.flag[pc], epoch, e.error)) { set_error(e, ERR_HANDSHAKE_TIMEOUT, (int)c, -1); return -1; }
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&sl.flag[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (NW > 1) __hip_atomic_store(&wg->flag[c % NW], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (for the partner of the next scan: a wave of this workgroup)
    if (!hs_wait(&sl.flag[pc], epoch, e.error)) { set_error(e, ERR_HANDSHAKE_TIMEOUT, (int)c, -1); return -1; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const double *theirs = sl.pub + ((pc * 2 + (int64_t)(epoch & 1ull)) * 4);
    lr_p = __hip_atomic_load(&theirs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); u_p = __hip_atomic_load(&theirs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot_p = (int)__hip_atomic_load(&theirs[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool lower = c < pc;
    const double uu = lower ? u : u_p;
    bool do_swap;
    if (e.target == 1) {
        do_swap = uu < e.test_swapper_pr;                  // TestSwapper, pair_swapper.jl:135-138
    } else {
        const double ex = exp(lr + lr_p);
        const double alpha = ex < 1.0 ? ex : 1.0;         // swap_acceptance_probability :88
        do_swap = uu < alpha;                               // swap_decision :81-85
#ifndef PTE_HS_MEASURE_NO_RECORD
        if (lower) {                                        // record_swap_stats! :59-66
            e.swap_sum[c] += alpha; e.swap_n[c] += 1;
            e.lsr_up[c] = dev_logaddexp(e.lsr_up[c], lr);
            e.lsr_dn[c] = dev_logaddexp(e.lsr_dn[c], lr_p);
            e.lsr_n[c] += 1;
            if (e.swap_log) { double *w = e.swap_log + (scan_idx * N + c) * 2; w[0] = lr; w[1] = lr_p; }
        }
#endif
    }
    if (do_swap) { e.chain_of_slot[slot_p] = (int32_t)c; e.slot_of_chain[c] = slot_p; return slot_p; }
    return slot;
}

#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_swap_stats(EngineDev e, int even, int64_t scan_idx) {
    const int64_t cl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cl >= e.K) return;
    const int64_t N = e.N, c = e.c0 + cl;
    const int64_t pc = deo_partner(N, even, c);
    const int slot = e.slot_of_chain[cl];
    double lr = 0.0;
    if (e.target != 1) {
        lr = swap_log_ratio(e, slot, c, pc);
        if (isnan(lr)) set_error(e, ERR_NAN_RATIO, (int)c, -1);
    }
    uint64_t seed = e.rng[2 * slot] + e.rng[2 * slot + 1];
    e.rng[2 * slot] = seed;
    e.stat[2 * cl] = lr;
    e.stat[2 * cl + 1] = u52_to_unit(mix64(seed));
    if (e.record_flags & 2u) {
        e.index_process[scan_idx * e.K + slot] = (int32_t)c;
        e.ip_replica[scan_idx * e.K + slot] = (int32_t)e.replica_id[slot];
    }
    if (e.record_flags & 1u) {
        const bool is_ref = is_ref_chain(e, c), is_tgt = (c == e.rt_tgt_a || c == e.rt_tgt_b);
        int64_t st = e.rt_state[slot];
        if (st == 0 && is_ref) e.rt_state[slot] = 1;
        else if (st == 1 && is_tgt) { e.rt_state[slot] = 2; e.rt_restarts[slot] += 1; }
        else if (st == 2 && is_ref) { e.rt_state[slot] = 1; e.rt_trips[slot] += 1; }
    }
}
#endif

#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_swap_decide(EngineDev e, int even, int64_t scan_idx) {
    const int64_t cl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cl >= e.K) return;
    const int64_t N = e.N, c = e.c0 + cl;
    const int64_t pc = deo_partner(N, even, c);
    const int slot = e.slot_of_chain[cl];
    int64_t new_cl = cl;
    if (pc != c) {
        const bool local = (pc >= e.c0 && pc < e.c0 + e.K);
        const int side = pc < c ? 0 : 1;
        const double lr = e.stat[2 * cl], u = e.stat[2 * cl + 1];
        const double lr_p = local ? e.stat[2 * (pc - e.c0)] : e.nbr_stat[2 * side];
        const double u_p = local ? e.stat[2 * (pc - e.c0) + 1] : e.nbr_stat[2 * side + 1];
        const bool lower = c < pc;
        const double uu = lower ? u : u_p;
        bool do_swap;
        if (e.target == 1) {
            do_swap = uu < e.test_swapper_pr;
        } else {
            const double ex = exp(lr + lr_p);
            const double alpha = ex < 1.0 ? ex : 1.0;
            do_swap = uu < alpha;
            if (lower) {
                e.swap_sum[cl] += alpha; e.swap_n[cl] += 1;
                e.lsr_up[cl] = dev_logaddexp(e.lsr_up[cl], lr);
                e.lsr_dn[cl] = dev_logaddexp(e.lsr_dn[cl], lr_p);
                e.lsr_n[cl] += 1;
                if (e.swap_log) { double *w = e.swap_log + (scan_idx * e.K + cl) * 2; w[0] = lr; w[1] = lr_p; }
            }
        }
        if (do_swap) {
            if (local) { e.chain_of_slot[slot] = (int32_t)pc; new_cl = pc - e.c0; }
            else e.bflag[side] = 1;        // the slot keeps its chain; its contents are traded
        }
    }
    e.slot_of_chain_alt[new_cl] = slot;
}
#endif

// payload layout (8-byte words, sw = e.sw state words: d coordinates, or the bit-packed Ising row):
// [0..sw) state, sw: sum x^2 / sum_pair_products, sw+1,sw+2: rng, sw+3: replica id, sw+4: round-trip state, sw+5: suff2
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_boundary_export(EngineDev e, int side, double *buf) {
    const int slot = e.slot_of_chain[side == 0 ? 0 : e.K - 1];
    const double *xrow = e.x + (int64_t)slot * e.ld;
    for (int64_t i = threadIdx.x; i < e.sw; i += blockDim.x) buf[i] = xrow[i];
    if (threadIdx.x == 0) {
        buf[e.sw] = e.suff[slot];
        unsigned long long *w = reinterpret_cast<unsigned long long *>(buf + e.sw + 1);
        w[0] = e.rng[2 * slot]; w[1] = e.rng[2 * slot + 1];
        w[2] = (unsigned long long)e.replica_id[slot];
        w[3] = (unsigned long long)e.rt_state[slot];
        buf[e.sw + 5] = e.suff2[slot];
    }
}
#endif
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_boundary_import(EngineDev e, int side, const double *buf) {
    const int slot = e.slot_of_chain[side == 0 ? 0 : e.K - 1];
    double *xrow = e.x + (int64_t)slot * e.ld;
    for (int64_t i = threadIdx.x; i < e.sw; i += blockDim.x) xrow[i] = buf[i];
    if (threadIdx.x == 0) {
        e.suff[slot] = buf[e.sw];
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(buf + e.sw + 1);
        e.rng[2 * slot] = w[0]; e.rng[2 * slot + 1] = w[1];
        e.replica_id[slot] = (int64_t)w[2];
        e.rt_state[slot] = (int64_t)w[3];
        e.suff2[slot] = buf[e.sw + 5];
    }
}
#endif

// ---- device-resident boundary exchange (no host round trip per scan) -----------------------------
// message layout (8-byte words): [0] log_ratio, [1] uniform of the boundary chain, [2 .. sw+8) payload.
// The payload travels speculatively with the SwapStat; the receiver applies it iff the swap is accepted
// (both sides take the same decision from the same two stats).
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_boundary_pack(EngineDev e, int active0, int active1, double *msg0, double *msg1) {
    const int side = blockIdx.x;
    if (!(side == 0 ? active0 : active1)) return;
    double *msg = side == 0 ? msg0 : msg1;
    const int64_t cl = side == 0 ? 0 : e.K - 1;
    const int slot = e.slot_of_chain[cl];
    const double *xrow = e.x + (int64_t)slot * e.ld;
    double *buf = msg + 2;
    for (int64_t i = threadIdx.x; i < e.sw; i += blockDim.x) buf[i] = xrow[i];
    if (threadIdx.x == 0) {
        msg[0] = e.stat[2 * cl]; msg[1] = e.stat[2 * cl + 1];
        buf[e.sw] = e.suff[slot];
        unsigned long long *w = reinterpret_cast<unsigned long long *>(buf + e.sw + 1);
        w[0] = e.rng[2 * slot]; w[1] = e.rng[2 * slot + 1];
        w[2] = (unsigned long long)e.replica_id[slot];
        w[3] = (unsigned long long)e.rt_state[slot];
        buf[e.sw + 5] = e.suff2[slot];
    }
}
#endif
#ifndef PTE_TU_LANGEVIN
__global__ void k_boundary_stats_in(EngineDev e, int active0, int active1, const double *msg0, const double *msg1) {
    if (threadIdx.x == 0) {
        e.nbr_stat[0] = active0 ? msg0[0] : 0.0; e.nbr_stat[1] = active0 ? msg0[1] : 0.0;
        e.nbr_stat[2] = active1 ? msg1[0] : 0.0; e.nbr_stat[3] = active1 ? msg1[1] : 0.0;
        e.bflag[0] = 0; e.bflag[1] = 0;
    }
}
#endif
// runs after k_swap_decide (slot maps already flipped on the host side: e.slot_of_chain is the new map,
// in which a boundary slot keeps its chain)
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(256) void k_boundary_apply(EngineDev e, const double *msg0, const double *msg1, int64_t *n_applied) {
    const int side = blockIdx.x;
    if (!e.bflag[side]) return;
    const double *buf = (side == 0 ? msg0 : msg1) + 2;
    const int slot = e.slot_of_chain[side == 0 ? 0 : e.K - 1];
    double *xrow = e.x + (int64_t)slot * e.ld;
    for (int64_t i = threadIdx.x; i < e.sw; i += blockDim.x) xrow[i] = buf[i];
    if (threadIdx.x == 0) {
        e.suff[slot] = buf[e.sw];
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(buf + e.sw + 1);
        e.rng[2 * slot] = w[0]; e.rng[2 * slot + 1] = w[1];
        e.replica_id[slot] = (int64_t)w[2];
        e.rt_state[slot] = (int64_t)w[3];
        e.suff2[slot] = buf[e.sw + 5];
        n_applied[side] += 1;
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// test kernels
// ---------------------------------------------------------------------------------------------
#ifndef PTE_TU_LANGEVIN
__global__ __launch_bounds__(64) void k_test_rng(uint64_t *sg, int kind, int64_t n, double *out) {
    const int lane = lane_id();
    if (kind == 0 || kind == 3) {
        uint64_t seed = sg[0], gamma = sg[1];
        const unsigned bb = rng_bool_bit();
        for (int64_t i = lane; i < n; i += 64) {
            const uint64_t raw = mix64(seed + (uint64_t)(i + 1) * gamma);
            out[i] = kind == 0 ? u52_to_unit(raw) : (double)((raw >> bb) & 1ull);
        }
        __syncthreads();
        if (lane == 0) sg[0] = seed + (uint64_t)n * gamma;
    } else if (kind == 1) {
        SeqRng r{sg[0], sg[1]};
        for (int64_t i0 = 0; i0 < n; i0 += 64) {
            int nl = (int)min((int64_t)64, n - i0);
            double v = wave_randn_block(r, lane, nl);
            if (lane < nl) out[i0 + lane] = v;
        }
        if (lane == 0) sg[0] = r.seed;
    } else {
        WaveDraws dr;
        dr.init(sg[0], sg[1], lane);
        for (int64_t i = 0; i < n; ++i) {
            uint64_t raw = dr.next_raw(lane);
            uint64_t ri = raw & MASK52;
            int idx = (int)(ri & 0xFF);
            double E = (double)ri * ZIG_WE[idx];
            if (!(ri < ZIG_KE[idx])) {
                SeqRng s = dr.to_seq();
                E = randexp_from_raw(s, raw);
                dr.from_seq(s, lane);
            }
            if (lane == 0) out[i] = E;
        }
        if (lane == 0) sg[0] = dr.final_seed();
    }
}
#endif

#ifndef PTE_TU_LANGEVIN
// pte_test_quotient: the quotient procedure of the Langevin-family kernels (pte_device.hpp: markstein_quotient and its guards) element by element:
// out[i] = what the kernels compute for a[i] / b[i], took_division[i] = 1 where the guards sent it to the division itself
__global__ __launch_bounds__(256) void k_test_quotient(const double *a, const double *b, int64_t n, double *out, int32_t *took_division) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double rinv = 1.0 / b[i];
    double q;
    const double m = markstein_quotient(a[i], b[i], rinv, q);
    const bool fast = markstein_divisor_ok(b[i]) && quotient_in_range(q);
    out[i] = fast ? m : a[i] / b[i];
    took_division[i] = fast ? 0 : 1;
}

__global__ __launch_bounds__(64) void k_test_sqr_norm(const double *x, int64_t rows, int64_t d, int nlu, double *out) {
    const int lane = lane_id();
    const int64_t r = blockIdx.x;
    if (r >= rows) return;
    const double *xrow = x + r * d;
    const int B = (int)((d + 63) >> 6);
    double acc = 0.0;   // lane b of chunk g holds block sum (64g + b); supports d up to 64*64*... via chunks of 64 blocks
    // d <= 4096 here (one register of block sums)
    for (int b = 0; b < B; ++b) {
        int64_t i = 64 * (int64_t)b + lane;
        double v = (i < d) ? xrow[i] : 0.0;
        double s = wave_tree_sum64(v * v);
        if (lane == b) acc = s;
    }
    double S = upper_tree_root_dyn(acc, nlu);
    if (lane == 0) out[r] = S;
}
#endif

}  // namespace pte
