// pte_ising.hpp -- k_explore_ising: single-site Metropolis sweeps of the 2-D Ising model
// (reference examples/ising.jl:96-116) and the Bernoulli(1/2) refresh of the reference chain
// (ising.jl:49-58), one wavefront per replica.
//
// The reference explorer is a SEQUENTIAL raster sweep whose RNG consumption is data dependent
// (a uniform is drawn only when the flip lowers the density), so a replica's L^2 * n_steps site
// updates form one dependency chain: the sweep is uniform integer work (spins staged in LDS, one
// byte per site), the 64 lanes pre-evaluate the next 64 draws of the counter-based stream and
// parallelise the load / store / refresh / energy recomputation.  Acceptance uses a filtered
// predicate: accept_ratio = exp(lp(spp') - lp(spp)) differs from exp(-|delta| beta beta_I) only by
// rounding of the two interpolated log potentials (<= 1e-10 relative), so `rand > accept_ratio` is
// decided against the per-chain constants with a 1e-9 guard band and evaluated exactly in the band.
#pragma once
#include "pte_slice2.hpp"

namespace pte {

struct IsingParams { int L; int n_steps; double beta_target; };

// InterpolatedLogPotential between IsingLogPotential(0.0, L) and IsingLogPotential(beta_target, L)
// (examples/ising.jl:74-77, src/paths/InterpolatedLogPotential.jl:9-16) as a function of sum_pair_products
__device__ __forceinline__ double ising_lp(double beta, double beta_target, double spp) {
    const double ref = 0.0 * spp, tgt = beta_target * spp;
    return beta == 0.0 ? ref : (beta == 1.0 ? tgt : (1.0 - beta) * ref + beta * tgt);
}

__device__ __forceinline__ int ising_site(const unsigned char *sp, int s) {       // uniform read of one spin: +1 / -1
    return __builtin_amdgcn_readfirstlane((int)sp[s]) ? 1 : -1;
}

__device__ inline long long ising_recompute(const unsigned char *sp, int L, int lane) {   // ising.jl:27-35
    long long acc = 0;
    const int d = L * L;
    for (int s = lane; s < d; s += 64) {
        const int i = s / L, j = s - i * L;
        const int up = ((i == 0 ? L : i) - 1) * L + j, dn = (i == L - 1 ? 0 : i + 1) * L + j;
        const int lf = i * L + (j == 0 ? L : j) - 1, rt = i * L + (j == L - 1 ? 0 : j + 1);
        const int sg = sp[s] ? 1 : -1;
        acc += sg * ((sp[up] ? 1 : -1) + (sp[dn] ? 1 : -1) + (sp[lf] ? 1 : -1) + (sp[rt] ? 1 : -1));
    }
    for (int k = 1; k < 64; k <<= 1) acc += __shfl_xor(acc, k, 64);
    return acc / 2;
}

__global__ __launch_bounds__(64) void k_explore_ising(EngineDev e, IsingParams ip) {
    extern __shared__ unsigned char spins[];
    const int lane = lane_id();
    const int64_t cl = blockIdx.x;
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    const int L = ip.L, d = L * L;
    double *xrow = e.x + (int64_t)slot * e.ld;
    uint64_t seed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];

    if (c == 0 && e.N > 1) {
        // iid_bernoulli!: site s (row-major, i outer / j inner) <- rand(rng, Bool) = low bit of draw s+1
        for (int s = lane; s < d; s += 64) spins[s] = (unsigned char)(mix64(seed + (uint64_t)(s + 1) * gamma) & 1ull);
        seed += (uint64_t)d * gamma;
        __syncthreads();
        const long long spp = ising_recompute(spins, L, lane);
        for (int s = lane; s < d; s += 64) xrow[s] = spins[s] ? 1.0 : 0.0;
        if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed; }
        return;
    }
    for (int s = lane; s < d; s += 64) spins[s] = xrow[s] != 0.0 ? 1 : 0;
    __syncthreads();
    long long spp = (long long)e.suff[slot];
    const double beta = e.beta[c], bt = ip.beta_target;
    const double bb = beta * bt;
    // |delta| = 4 or 8: guard-banded thresholds for `rand > accept_ratio`
    const double r4 = exp(-4.0 * bb), r8 = exp(-8.0 * bb);
    const double r4lo = r4 * (1.0 - 1e-9), r4hi = r4 * (1.0 + 1e-9), r8lo = r8 * (1.0 - 1e-9), r8hi = r8 * (1.0 + 1e-9);
    const bool filter_ok = bb > 1e-6;       // below: the two log potentials may round equal -> always decide exactly

    // 64 buffered uniforms of the replica's stream
    double unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma));
    int p = 0;

    for (int k = 0; k < ip.n_steps; ++k) {
        int s = 0;
        for (int i = 0; i < L; ++i) {
            const int rowu = ((i == 0 ? L : i) - 1) * L, rowd = (i == L - 1 ? 0 : i + 1) * L, row = i * L;
            for (int j = 0; j < L; ++j, ++s) {
                const int sg = ising_site(spins, s);
                const int nb = ising_site(spins, rowu + j) + ising_site(spins, rowd + j) +
                               ising_site(spins, row + (j == 0 ? L : j) - 1) + ising_site(spins, row + (j == L - 1 ? 0 : j + 1));
                const int delta = -2 * sg * nb;            // sum_pair_products after - before (flip!, ising.jl:38-46)
                bool accept = true;
                if (delta < 0) {
                    bool need_draw = true, decided = false;
                    double ratio = 0.0;
                    if (__builtin_expect(!filter_ok, 0)) {
                        ratio = exp(ising_lp(beta, bt, (double)(spp + delta)) - ising_lp(beta, bt, (double)spp));
                        need_draw = ratio < 1;
                        decided = true;
                    }
                    if (need_draw) {
                        if (p == 64) { seed += 64ull * gamma; unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma)); p = 0; }
                        const double u = readlane_f64(unit, p);
                        p += 1;
                        if (!decided) {
                            const double lo = delta == -4 ? r4lo : r8lo, hi = delta == -4 ? r4hi : r8hi;
                            if (u > hi) accept = false;
                            else if (u < lo) accept = true;
                            else {
                                ratio = exp(ising_lp(beta, bt, (double)(spp + delta)) - ising_lp(beta, bt, (double)spp));
                                accept = !(ratio < 1 && u > ratio);
                            }
                        } else {
                            accept = !(u > ratio);
                        }
                    }
                }
                if (accept) {
                    if (lane == 0) spins[s] = sg > 0 ? 0 : 1;
                    spp += delta;
                }
            }
        }
    }
    __syncthreads();
    for (int s = lane; s < d; s += 64) xrow[s] = spins[s] ? 1.0 : 0.0;
    if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed + (uint64_t)p * gamma; }
}

}  // namespace pte
