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

#ifndef PTE_ISING_FILTER_MIN
#define PTE_ISING_FILTER_MIN 1e-13          // beta * beta_target above which the thresholds decide (see k_explore_ising)
#endif
namespace pte {

struct IsingParams { int L; int n_steps; double beta_target; };

#ifndef PTE_ISING_STORE
#define PTE_ISING_STORE 2       // how k_explore_ising_spec writes a swept word back: 0 = lane 0 if it changed (a compare, two scalar ANDs, an EXEC save / restore:
                                // 4.34 ms per scan at the C5 shard shape), 1 = lane 0 always (4.32), 2 = every lane the same word to the same address (4.18)
#endif

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
    const int L = ip.L, d = L * L, NW = (d + 31) >> 5;
    unsigned *wrow = reinterpret_cast<unsigned *>(e.x + (int64_t)slot * e.ld);     // the lattice bit-packed in HBM: site s -> bit s & 31 of word s >> 5
    auto store_lattice = [&]() {                                                    // LDS bytes -> HBM bits (call after a barrier)
        for (int wd = lane; wd < NW; wd += 64) {
            unsigned v = 0;
            for (int t = 0; t < 32 && 32 * wd + t < d; ++t) v |= (unsigned)(spins[32 * wd + t] & 1u) << t;
            wrow[wd] = v;
        }
    };
    uint64_t seed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];
    const double lp_before = lp_before_explore(e, c, slot);

    if (is_ref_chain(e, c)) {
        // iid_bernoulli!: site s (row-major, i outer / j inner) <- rand(rng, Bool) = low bit of draw s+1
        const unsigned bb = rng_bool_bit();              // include/pte_rng_policy.h (default 0: `% Bool`)
        for (int s = lane; s < d; s += 64) spins[s] = (unsigned char)((mix64(seed + (uint64_t)(s + 1) * gamma) >> bb) & 1ull);
        seed += (uint64_t)d * gamma;
        __syncthreads();
        const long long spp = ising_recompute(spins, L, lane);
        store_lattice();
        if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed; }
        record_after_explore(e, cl, c, slot, lane, lp_before, (double)spp, 0.0);
        return;
    }
    for (int s = lane; s < d; s += 64) spins[s] = (unsigned char)((wrow[s >> 5] >> (s & 31)) & 1u);
    __syncthreads();
    long long spp = (long long)e.suff[slot];
    const double beta = e.beta[c], bt = ip.beta_target;
    const double bb = beta * bt;
    // |delta| = 4 or 8: guard-banded thresholds for `rand > accept_ratio`
    const double r4 = exp(-4.0 * bb), r8 = exp(-8.0 * bb);
    const double r4lo = r4 * (1.0 - 1e-9), r4hi = r4 * (1.0 + 1e-9), r8lo = r8 * (1.0 - 1e-9), r8hi = r8 * (1.0 + 1e-9);
    // When may a proposal with delta < 0 be decided by comparing the uniform with the guard-banded thresholds?  The reference evaluates
    // exp(lp(S + delta) - lp(S)) with lp(S) = fl(beta * fl(bt * S)): the two roundings put a relative error of <= 2^-51 S / |delta| on the
    // exponent (1.5e-11 at 256 x 256, 2.4e-10 at 1024 x 1024: inside the 1e-9 band whatever beta is), and what the filter ALSO assumes --
    // that a uniform is drawn at all, i.e. that this ratio is < 1 in floating point -- holds while 4 beta bt is well above 2^-53.  Below the
    // limit every decision takes the exact arithmetic (one full recount of the lattice per decision: ~1 us).  (Rounds 3-5 had 1e-6 here,
    // and the second chain of a ladder adapted on a handful of scans does get there: 2.6e-7 after round 2 of C5 -- 207 ms per scan for that
    // round instead of 3.8, tools/diag_regimes.py.)
    const bool filter_ok = bb > PTE_ISING_FILTER_MIN;

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
    store_lattice();
    if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed + (uint64_t)p * gamma; }
    record_after_explore(e, cl, c, slot, lane, lp_before, (double)spp, 0.0);
}

}  // namespace pte

namespace pte {

// ---------------------------------------------------------------------------------------------
// k_explore_ising_bits: same sweep for base_length % 32 == 0 with the lattice bit-packed in LDS
// (L*L/8 bytes) and the current / upper / lower words of the row held in scalar registers: the
// per-site work is ~25 scalar integer instructions; the uniform is compared against the guard-banded
// thresholds in the integer domain (bit patterns of positive doubles are ordered like the doubles).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_word(const unsigned *w, int i) { return (unsigned)__builtin_amdgcn_readfirstlane((int)w[i]); }

#ifdef PTE_TEST_KERNELS   // scalar bit-packed sweep: dominated by k_explore_ising_spec, kept in libpte_test.so for A/B parity
__global__ __launch_bounds__(64) void k_explore_ising_bits(EngineDev e, IsingParams ip) {
    extern __shared__ unsigned words[];
    const int lane = lane_id();
    const int64_t cl = blockIdx.x;
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    const int L = ip.L, d = L * L, W = L >> 5, NW = d >> 5;
    unsigned *wrow = reinterpret_cast<unsigned *>(e.x + (int64_t)slot * e.ld);     // bit-packed lattice in HBM, same word layout as the LDS copy
    uint64_t seed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];
    const double lp_before = lp_before_explore(e, c, slot);

    if (is_ref_chain(e, c)) {
        for (int wd = lane; wd < NW; wd += 64) {
            unsigned v = 0;
            const unsigned bb = rng_bool_bit();
            for (int t = 0; t < 32; ++t) v |= (unsigned)((mix64(seed + (uint64_t)(32 * wd + t + 1) * gamma) >> bb) & 1ull) << t;
            words[wd] = v;
        }
        seed += (uint64_t)d * gamma;
    } else {
        for (int wd = lane; wd < NW; wd += 64) words[wd] = wrow[wd];
    }
    __syncthreads();
    long long spp;
    if (is_ref_chain(e, c)) {
        // recompute_sum_pair_products: every bond once = sum over sites of (right + down neighbour products)
        long long acc = 0;
        for (int wd = lane; wd < NW; wd += 64) {
            const int i = wd / W, wj = wd - i * W;
            const unsigned cur = words[wd], dn = words[(i == L - 1 ? 0 : i + 1) * W + wj];
            const unsigned nxt = words[i * W + (wj == W - 1 ? 0 : wj + 1)];
            const unsigned right = (cur >> 1) | (nxt << 31);
            acc += 64 - 2 * ((int)__popc(cur ^ right) + (int)__popc(cur ^ dn));     // +1 per equal pair, -1 per unequal pair
        }
        for (int k = 1; k < 64; k <<= 1) acc += __shfl_xor(acc, k, 64);
        spp = acc;
    } else {
        spp = (long long)e.suff[slot];
        const double beta = e.beta[c], bt = ip.beta_target;
        const double bb = beta * bt;
        const double r4 = exp(-4.0 * bb), r8 = exp(-8.0 * bb);
        // guard-banded thresholds as integer bit patterns held in scalar registers
        auto hi32 = [](double v) { return (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v)); };
        auto lo32 = [](double v) { return (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v)); };
        const double r4l = r4 * (1.0 - 1e-9), r4h = r4 * (1.0 + 1e-9), r8l = r8 * (1.0 - 1e-9), r8h = r8 * (1.0 + 1e-9);
        const unsigned r4lo_h = hi32(r4l), r4hi_h = hi32(r4h), r8lo_h = hi32(r8l), r8hi_h = hi32(r8h);
        const unsigned long long r4lo = ((unsigned long long)r4lo_h << 32) | lo32(r4l), r4hi = ((unsigned long long)r4hi_h << 32) | lo32(r4h);
        const unsigned long long r8lo = ((unsigned long long)r8lo_h << 32) | lo32(r8l), r8hi = ((unsigned long long)r8hi_h << 32) | lo32(r8h);
        const bool filter_ok = bb > PTE_ISING_FILTER_MIN;
        double unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma));
        int p = 0;
        for (int k = 0; k < ip.n_steps; ++k) {
            for (int i = 0; i < L; ++i) {
                const int rowu = ((i == 0 ? L : i) - 1) * W, rowd = (i == L - 1 ? 0 : i + 1) * W, row = i * W;
                unsigned leftbit = lds_word(words, row + W - 1) >> 31;       // left neighbour of (i, 0): (i, L-1), not yet updated
                unsigned first_updated = 0;
                for (int wj = 0; wj < W; ++wj) {
                    unsigned cur = lds_word(words, row + wj);
                    const unsigned up = lds_word(words, rowu + wj), dn = lds_word(words, rowd + wj);
                    // right neighbour of bit 31: bit 0 of the next word (old value), or of word 0 of this row (updated) at the row end
                    const unsigned rightbit = (wj == W - 1) ? (first_updated & 1u) : (lds_word(words, row + wj + 1) & 1u);
                    const unsigned cur0 = cur;
                    for (int t = 0; t < 32; ++t) {
                        // branch-light scalar code: selects instead of jumps, one rare branch for the guard band
                        const unsigned sgb = (cur >> t) & 1u;
                        const unsigned lf = t == 0 ? leftbit : ((cur >> (t - 1)) & 1u);
                        // at the end of a one-word row the right neighbour of bit 31 is bit 0 of this very word (already updated)
                        const unsigned rt = t == 31 ? (W == 1 ? (cur & 1u) : rightbit) : ((cur >> (t + 1)) & 1u);
                        const int nb = 2 * (int)(((up >> t) & 1u) + ((dn >> t) & 1u) + lf + rt) - 4;
                        const int delta = (1 - 2 * (int)sgb) * 2 * nb;
                        const int need = (delta < 0) ? 1 : 0;
                        if (__builtin_expect(p == 64, 0)) { seed += 64ull * gamma; unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma)); p = 0; }
                        const unsigned uhi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(unit), p);   // read speculatively, consumed iff `need`
                        const unsigned hi_h = delta == -4 ? r4hi_h : r8hi_h, lo_h = delta == -4 ? r4lo_h : r8lo_h;
                        int rej = need & (uhi > hi_h ? 1 : 0);
                        const int sure_acc = uhi < lo_h ? 1 : 0;
                        if (__builtin_expect((need & (1 - rej) & (1 - sure_acc)) | (need & (filter_ok ? 0 : 1)), 0)) {
                            // guard band (or a chain where the filter is not valid): exact arithmetic of the reference
                            const unsigned ulo = (unsigned)__builtin_amdgcn_readlane(__double2loint(unit), p);
                            const unsigned long long ub = ((unsigned long long)uhi << 32) | ulo;
                            const unsigned long long lo = delta == -4 ? r4lo : r8lo, hi = delta == -4 ? r4hi : r8hi;
                            if (filter_ok && ub > hi) rej = 1;
                            else if (filter_ok && ub < lo) rej = 0;
                            else {
                                const double ratio = exp(ising_lp(beta, bt, (double)(spp + delta)) - ising_lp(beta, bt, (double)spp));
                                if (ratio < 1) rej = (__longlong_as_double((long long)ub) > ratio) ? 1 : 0;
                                else { rej = 0; p -= 1; }        // accept_ratio >= 1: the reference draws nothing
                            }
                        }
                        p += need;
                        const int acc = 1 - rej;
                        cur ^= (unsigned)acc << t;
                        spp += acc * delta;
                    }
                    if (cur != cur0 && lane == 0) words[row + wj] = cur;
                    if (wj == 0) first_updated = cur;
                    leftbit = cur >> 31;
                }
            }
        }
        seed += (uint64_t)p * gamma;
    }
    __syncthreads();
    for (int wd = lane; wd < NW; wd += 64) wrow[wd] = words[wd];
    if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed; }
    record_after_explore(e, cl, c, slot, lane, lp_before, (double)spp, 0.0);
}

#endif  // PTE_TEST_KERNELS

// ---------------------------------------------------------------------------------------------
// k_explore_ising_spec: the bit-packed sweep with the 64 lanes as hypotheses, the way k_explore_slice7/8
// break the slice sampler's chain.  The outcome of site t depends on the sweep so far only through
// (b, c): the NEW value b of its left neighbour and the number c of uniforms consumed since the chunk
// started (which picks the uniform it would read).  A 16-site chunk is cut into four quads; quad k can start
// in 2 (4k + 1) states, 56 hypotheses in all: each lane walks the four sites of its quad under its (c, b)
// in one vector pass (neighbour counts, deltas, the filtered accept decisions against the integer thresholds),
// and a scalar chase of four steps per chunk picks the true quads and carries the state on.  Guard-band decisions (and chains
// whose filter is not valid) are taken by the exact arithmetic of the reference, with sum_pair_products
// recomputed on demand; the final sum_pair_products is recomputed from the lattice by popcounts.
// ---------------------------------------------------------------------------------------------
// ONE_WORD: base_length == 32 (a row is one word: bit 31's right neighbour is bit 0 of the same word, swept in the same iteration) -- its
// own instantiation, so that the wider lattices do not test for it twice per word.  Dynamic LDS: L * L / 8 bytes + 8 (the read-ahead of the
// word to the right runs two words past a row's end).
template <bool ONE_WORD>
__global__ __launch_bounds__(64) void k_explore_ising_spec(EngineDev e, IsingParams ip) {
    extern __shared__ unsigned words[];
    const int lane = lane_id();
    const int64_t cl = blockIdx.x;
    if (cl >= e.K) return;
    const int64_t c = e.c0 + cl;
    const int slot = e.slot_of_chain[cl];
    const int L = ip.L, d = L * L, W = ONE_WORD ? 1 : (L >> 5), NW = d >> 5;
    unsigned *wrow = reinterpret_cast<unsigned *>(e.x + (int64_t)slot * e.ld);     // bit-packed lattice in HBM, same word layout as the LDS copy
    uint64_t seed = e.rng[2 * slot];
    const uint64_t gamma = e.rng[2 * slot + 1];
    const double lp_before = lp_before_explore(e, c, slot);
    const bool refresh = is_ref_chain(e, c);
#ifdef PTE_PROFILE_WAVES
    const uint64_t wave_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PTE_PROFILE_ISING_SECTIONS
    unsigned long long prof_pass = 0, prof_chase = 0, prof_loop = 0;
#endif

    for (int wd = lane; wd < NW; wd += 64) {
        unsigned v = 0;
        if (refresh) { const unsigned bb = rng_bool_bit(); for (int t = 0; t < 32; ++t) v |= (unsigned)((mix64(seed + (uint64_t)(32 * wd + t + 1) * gamma) >> bb) & 1ull) << t; }
        else         { v = wrow[wd]; }
        words[wd] = v;
    }
    if (refresh) seed += (uint64_t)d * gamma;
    __syncthreads();
    // recompute_sum_pair_products from the LDS lattice: every bond once (right + down neighbour products)
    auto recompute = [&]() -> long long {
        long long acc = 0;
        for (int wd = lane; wd < NW; wd += 64) {
            const int i = wd / W, wj = wd - i * W;
            const unsigned cur = words[wd], dn = words[(i == L - 1 ? 0 : i + 1) * W + wj];
            const unsigned nxt = words[i * W + (wj == W - 1 ? 0 : wj + 1)];
            const unsigned right = (cur >> 1) | (nxt << 31);
            acc += 64 - 2 * ((int)__popc(cur ^ right) + (int)__popc(cur ^ dn));
        }
        for (int k = 1; k < 64; k <<= 1) acc += __shfl_xor(acc, k, 64);
        return acc;
    };
    if (!refresh) {
        const double beta = e.beta[c], bt = ip.beta_target;
        const double bb = beta * bt;
        const double r4 = exp(-4.0 * bb), r8 = exp(-8.0 * bb);
        auto hi32 = [](double v) { return (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v)); };
        auto lo32 = [](double v) { return (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v)); };
        const double r4l = r4 * (1.0 - 1e-9), r4h = r4 * (1.0 + 1e-9), r8l = r8 * (1.0 - 1e-9), r8h = r8 * (1.0 + 1e-9);
        const unsigned r4lo_h = hi32(r4l), r4hi_h = hi32(r4h), r8lo_h = hi32(r8l), r8hi_h = hi32(r8h);
        const unsigned long long r4lo = ((unsigned long long)r4lo_h << 32) | lo32(r4l), r4hi = ((unsigned long long)r4hi_h << 32) | lo32(r4h);
        const unsigned long long r8lo = ((unsigned long long)r8lo_h << 32) | lo32(r8l), r8hi = ((unsigned long long)r8hi_h << 32) | lo32(r8h);
        const bool filter_ok = bb > PTE_ISING_FILTER_MIN;
        // this lane's hypothesis (lk, lc, lb): quad lk of a 16-site chunk (sites 4 lk .. 4 lk + 3), lc uniforms consumed
        // since the chunk started, left neighbour of the quad's first site now lb; 2 (4 lk + 1) hypotheses per quad = 56 lanes
        const int lk = (lane >= 2) + (lane >= 12) + (lane >= 30);
        const int lbase = lk == 0 ? 0 : lk == 1 ? 2 : lk == 2 ? 12 : 30;
        const int lidx = lane - lbase;
        const int lc = lidx >> 1;
        const unsigned lb = (unsigned)(lidx & 1);
        const int lnext = (lk == 0 ? 2 : lk == 1 ? 12 : lk == 2 ? 30 : 0) + 2 * lc;      // lane of the next quad's hypothesis (c, spin) = lnext + 2 dc + spin
        const int lacc_sh = 7 + 4 * lk;
        double unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma));
        // The 64 buffered uniforms enter the vector pass only through four comparisons of their high words with the guard-banded
        // thresholds: taken once per refill for the whole buffer (four ballots, bit i = uniform i), a hypothesis that has consumed
        // lc uniforms reads bit (p + lc + its own count) of the mask its delta selects -- no LDS copy of the buffer, no load on the
        // chain p -> pass -> chase -> p.   R = certainly rejected (u above the band), A = inside the band (or no valid filter).
        unsigned long long mR4, mA4, mR8, mA8;
        auto classify = [&]() {
            const unsigned uh = (unsigned)__double2hiint(unit);
            mR4 = ballot64(uh > r4hi_h); mR8 = ballot64(uh > r8hi_h);
            mA4 = filter_ok ? ballot64(!(uh > r4hi_h) && !(uh < r4lo_h)) : ~0ull;
            mA8 = filter_ok ? ballot64(!(uh > r8hi_h) && !(uh < r8lo_h)) : ~0ull;
        };
        classify();
        int p = 0;
        // What a chunk's sites need from their surroundings (see the vector pass below) does not depend on the sweep of the chunk BEFORE it:
        // that one flips its own 16 bits only.  So the boolean functions of a chunk are evaluated before the chase of the previous one, in
        // whose wait states they can issue (a chained hop leaves ~20 cycles in which a lone wave issues nothing otherwise).
        struct ChunkStatics { unsigned NN, II, SN; };
        auto chunk_statics = [&](unsigned upw, unsigned dnw, unsigned curw, unsigned cur_r, int T0) -> ChunkStatics {
            const int t0 = T0 + 4 * lk;
            const unsigned U = (upw >> t0) & 15u, D = (dnw >> t0) & 15u, R = (cur_r >> t0) & 15u, S = (curw >> t0) & 15u;
            const unsigned b0 = U ^ D ^ R, b1 = (U & D) | (R & (U ^ D));
            const unsigned N0 = (S & b1 & b0) | (~S & ~b1),       I0 = (S & b1 & b0) | (~S & ~b1 & b0);
            const unsigned N1 = (S & b1) | (~S & ~b1 & ~b0),      I1 = (S & b1 & ~b0) | (~S & ~b1 & ~b0);
            return ChunkStatics{(N0 & 15u) | ((N1 & 15u) << 4), (I0 & 15u) | ((I1 & 15u) << 4), ~S};      // bit j + 4 left
        };
#ifdef PTE_PROFILE_ISING_SECTIONS          // debug builds only (with -DPTE_PROFILE_WAVES): shader-clock cycles of the vector pass / the chase, summed over the chunks
        const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
        for (int k = 0; k < ip.n_steps; ++k) {
            for (int i = 0; i < L; ++i) {
                const int rowu = ((i == 0 ? L : i) - 1) * W, rowd = (i == L - 1 ? 0 : i + 1) * W, row = i * W;
                unsigned b = lds_word(words, row + W - 1) >> 31;             // left neighbour of (i, 0): (i, L-1), not yet updated
                unsigned first_updated = 0;
                // The words of a row are read one iteration AHEAD (the sweep of word wj writes words[row + wj] only; the rows above and
                // below and the words to its right keep their values while it runs): the LDS round trip of the next word's three reads
                // (~120 cycles of a lone wave, 8 % of a word's time) runs under this word's two passes, and the word to the right --
                // read for its bit 0 -- IS the next word to sweep.
                unsigned cur = lds_word(words, row), up = lds_word(words, rowu), dn = lds_word(words, rowd);
                unsigned nxt = ONE_WORD ? 0u : lds_word(words, row + 1);
                ChunkStatics st0 = chunk_statics(up, dn, cur, cur >> 1, 0), st1 = st0;      // (chunk 0 never looks at bit 31's right neighbour)
                for (int wj = 0; wj < W; ++wj) {
                    // (every lane reads the same address: a broadcast.  Unconditional: behind a row's last word these are words of the next
                    // row or of the two words of padding behind the lattice, and nobody uses them -- a branch around three loads costs more)
                    const unsigned pf_up = words[rowu + wj + 1], pf_dn = words[rowd + wj + 1], pf_nx = words[row + wj + 2];
                    const unsigned rightbit = (wj == W - 1) ? (first_updated & 1u) : (nxt & 1u);
#if PTE_ISING_STORE == 0
                    const unsigned cur0 = cur;
#endif
#pragma unroll
                    for (int T0 = 0; T0 < 32; T0 += 16) {
#ifdef PTE_PROFILE_ISING_SECTIONS
                        const unsigned long long pa = __builtin_readcyclecounter();
#endif
                        if (__builtin_expect(p + 16 > 64, 0)) {           // (one chunk in ~5: laid out behind the loop, so that the common path falls through -- a lone wave refetches after a taken branch)
                            seed += (uint64_t)p * gamma; unit = u52_to_unit(mix64(seed + (uint64_t)(lane + 1) * gamma)); p = 0;
                            classify();
                        }
                        const unsigned rt31 = ONE_WORD ? (cur & 1u) : rightbit;
                        // ---- vector pass: every (quad, consumed, left) hypothesis of the chunk walks its four sites
                        // (its uniforms are the next <= 4 of the buffer from position p + lc: bits p + lc .. of the masks)
                        const int sh = p + lc;                                   // <= 48 + 12
                        unsigned wR4 = (unsigned)(mR4 >> sh), wA4 = (unsigned)(mA4 >> sh), wR8 = (unsigned)(mR8 >> sh), wA8 = (unsigned)(mA8 >> sh);
                        asm volatile("" : "+v"(wR4), "+v"(wA4), "+v"(wR8), "+v"(wA8));   // (keep the four 64-bit shifts here: hipcc sinks them below the per-site selects, 8 per pass)
                        // What a site needs from its surroundings does not depend on the walk except through its NEW left neighbour: for
                        // the quad's four sites at once (bit j = site j), from the nibbles of the word above, below, to the right (old
                        // values) and of the spins themselves -- cnt = neighbours that are 1 = (U + D + R) + left, delta = (1 - 2 s) 2 (2 cnt - 4):
                        //   a draw is needed iff  s ? cnt > 2 : cnt < 2,   delta == -4 iff  s ? cnt == 3 : cnt == 1
                        // as boolean functions of (b1 b0 = U + D + R, s), once for left = 0 and once for left = 1; the walk then only
                        // picks bits: 12 instead of 19 instructions per site.
                        if (T0 == 16 && ONE_WORD) st1 = chunk_statics(up, dn, cur, (cur >> 1) | (rt31 << 31), 16);   // (a one-word row: bit 31's right neighbour is bit 0, just swept)
                        const ChunkStatics st = T0 == 0 ? st0 : st1;
                        const unsigned NN = st.NN, II = st.II, SN = st.SN;
                        const unsigned WR = (wR8 & 15u) | ((wR4 & 15u) << 4), WA = (wA8 & 15u) | ((wA4 & 15u) << 4);   // bit dc + 4 [delta == -4]
                        int dc = 0;
                        unsigned left = lb, ambu = 0, rejn = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned shj = (left << 2) + (unsigned)j;
                            const unsigned need = (NN >> shj) & 1u, is4 = (II >> shj) & 1u;
                            const unsigned idx = (is4 << 2) + (unsigned)dc;
                            const unsigned rej = need & (WR >> idx);             // (bit 0; the bits above are dropped where it is used)
                            ambu |= need & (WA >> idx);
                            left = ((SN >> j) ^ rej) & 1u;                       // the site's new spin: flipped unless rejected
                            rejn |= (rej & 1u) << j;
                            dc += (int)need;
                        }
                        const int accbits = (int)(rejn ^ 15u);
                        ambu &= 1u;
                        // The word a hypothesis hands to the chase: bits 0-5 = the LANE of the hypothesis that continues it in the next quad
                        // (quad 3: the state 2 c + spin the chunk ends in), bit 6 = a guard-band decision somewhere in the quad, bits 7-22 =
                        // its accepts already at the quad's place in the chunk.  A hop is then ONE v_readlane whose lane select is the word
                        // read before (the hardware takes bits 0-5), and the chunk's flips are the OR of the four words: round 4 priced a hop
                        // with a shift and an add between the reads at 34-42 cycles against 27.5 chained (tools/ubench/round_cost.hip).
                        const int pk = (lnext + 2 * dc + (int)left) | ((int)ambu << 6) | (accbits << lacc_sh);
                        // ---- chase over the four quads: state s2 = 2 c + b
                        int s2 = (int)b;
#ifdef PTE_PROFILE_ISING_SECTIONS
                        asm volatile("" :: "v"(pk));
                        const unsigned long long pb = __builtin_readcyclecounter();
#endif
                        // all four quads at once when none of them met a guard-band decision (the common case): no branches.
                        // The statics of the chunk AFTER this one are evaluated in three pieces of four instructions BETWEEN the hops: the
                        // empty asm statements tie each piece's inputs to the hop before it and its results to the hop after it (pure
                        // data flow: hipcc would otherwise schedule all of it above the first hop and fill the gaps with s_nop).
                        const unsigned n_up = T0 == 0 ? up : pf_up, n_dn = T0 == 0 ? dn : pf_dn, n_cw = T0 == 0 ? cur : nxt;
                        const unsigned n_cr = T0 == 0 ? ((cur >> 1) | (rightbit << 31)) : (nxt >> 1);
                        int nt0 = (T0 == 0 ? 16 : 0) + 4 * lk;
                        int q0 = __builtin_amdgcn_readlane(pk, s2);
                        asm volatile("" : "+s"(q0), "+v"(nt0));
                        unsigned sU = n_up >> nt0, sD = n_dn >> nt0, sR = n_cr >> nt0, sS = n_cw >> nt0;
                        asm volatile("" : "+s"(q0), "+v"(sU), "+v"(sD), "+v"(sR), "+v"(sS));
                        int q1 = __builtin_amdgcn_readlane(pk, q0);
                        asm volatile("" : "+s"(q1), "+v"(sU), "+v"(sD), "+v"(sR), "+v"(sS));
                        unsigned sb0 = sU ^ sD ^ sR, sb1 = (sU & sD) | (sR & (sU ^ sD));
                        unsigned sN0 = (sS & sb1 & sb0) | (~sS & ~sb1), sN1 = (sS & sb1) | (~sS & ~sb1 & ~sb0);
                        asm volatile("" : "+s"(q1), "+v"(sb0), "+v"(sb1), "+v"(sN0), "+v"(sN1));
                        int q2 = __builtin_amdgcn_readlane(pk, q1);
                        asm volatile("" : "+s"(q2), "+v"(sb0), "+v"(sb1), "+v"(sN0), "+v"(sN1));
                        unsigned sI0 = (sS & sb1 & sb0) | (~sS & ~sb1 & sb0), sI1 = (sS & sb1 & ~sb0) | (~sS & ~sb1 & ~sb0);
                        unsigned sNN = (sN0 & 15u) | ((sN1 & 15u) << 4);
                        asm volatile("" : "+s"(q2), "+v"(sI0), "+v"(sI1), "+v"(sNN));
                        const int q3 = __builtin_amdgcn_readlane(pk, q2);
                        {
                            const ChunkStatics nst{sNN, (sI0 & 15u) | ((sI1 & 15u) << 4), ~sS};
                            if (T0 == 0) st1 = nst; else st0 = nst;          // (T0 == 16: the next word, read ahead; zeros behind the row's last word)
                        }
                        const int qa = q0 | q1 | q2 | q3;
                        // (the result of the common case first, ONE branch around the rest: with an if / else hipcc keeps a "took the fast
                        // side" flag in a scalar pair and tests it again behind the join)
                        unsigned cur_fast = cur ^ ((((unsigned)qa >> 7) & 0xFFFFu) << T0);
                        int s2_fast = q3 & 63;
                        asm volatile("" : "+s"(cur_fast), "+s"(s2_fast));      // (evaluated HERE: hipcc sinks them into an else side otherwise)
                        if (__builtin_expect((qa & 64) != 0, 0)) {
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq) {
                                const int qbase = kq == 0 ? 0 : kq == 1 ? 2 : kq == 2 ? 12 : 30, nbase = kq == 0 ? 2 : kq == 1 ? 12 : kq == 2 ? 30 : 0;
                                const int q = __builtin_amdgcn_readlane(pk, qbase + s2);
                                if (__builtin_expect(q & 64, 0)) {
                                    // a guard-band decision (or a chain whose filter is not valid) inside this quad: its four sites by
                                    // the scalar procedure with the exact arithmetic of the reference where needed
                                    int cc = s2 >> 1;
                                    unsigned bb_ = (unsigned)(s2 & 1);
                                    for (int j = 0; j < 4; ++j) {
                                        const int tt = T0 + 4 * kq + j;
                                        const unsigned sgs = (cur >> tt) & 1u;
                                        const unsigned rts = tt == 31 ? rt31 : ((cur >> (tt + 1)) & 1u);
                                        const int nbs = 2 * (int)(((up >> tt) & 1u) + ((dn >> tt) & 1u) + bb_ + rts) - 4;
                                        const int dl = (1 - 2 * (int)sgs) * 2 * nbs;
                                        int rj = 0, nd = 0;
                                        if (dl < 0) {
                                            nd = 1;
                                            const unsigned uh = (unsigned)__builtin_amdgcn_readlane(__double2hiint(unit), p + cc);
                                            const unsigned ul = (unsigned)__builtin_amdgcn_readlane(__double2loint(unit), p + cc);
                                            const unsigned long long ub = ((unsigned long long)uh << 32) | ul;
                                            const unsigned long long lo = dl == -4 ? r4lo : r8lo, hi = dl == -4 ? r4hi : r8hi;
                                            if (filter_ok && ub > hi) rj = 1;
                                            else if (filter_ok && ub < lo) rj = 0;
                                            else {
                                                if (lane == 0) words[row + wj] = cur;
                                                __syncthreads();
                                                const long long spp = recompute();
                                                const double ratio = exp(ising_lp(beta, bt, (double)(spp + dl)) - ising_lp(beta, bt, (double)spp));
                                                if (ratio < 1) rj = (__longlong_as_double((long long)ub) > ratio) ? 1 : 0;
                                                else { rj = 0; nd = 0; }          // accept_ratio >= 1: the reference draws nothing
                                            }
                                        }
                                        cur ^= (unsigned)(rj ? 0 : 1) << tt;
                                        bb_ = (cur >> tt) & 1u;
                                        cc += nd;
                                    }
                                    // (every lane computed the same values; tell the compiler, so that the chunk loop stays scalar)
                                    s2 = __builtin_amdgcn_readfirstlane(2 * cc + (int)bb_);
                                    cur = (unsigned)__builtin_amdgcn_readfirstlane((int)cur);
                                } else {
                                    cur ^= (((unsigned)q >> 7) & 0xFFFFu) << T0;      // (the accepts sit at the quad's place; the other quads' bits are 0)
                                    s2 = (q & 63) - nbase;
                                }
                            }
                        } else {
                            cur = cur_fast; s2 = s2_fast;
                        }
#ifdef PTE_PROFILE_ISING_SECTIONS
                        asm volatile("" :: "s"(s2), "s"(cur));
                        { const unsigned long long pc = __builtin_readcyclecounter(); prof_pass += pb - pa; prof_chase += pc - pb; }
#endif
                        p += s2 >> 1;
                        b = (unsigned)(s2 & 1);
                    }
#if PTE_ISING_STORE == 0
                    if (cur != cur0 && lane == 0) words[row + wj] = cur;
#elif PTE_ISING_STORE == 1
                    if (lane == 0) words[row + wj] = cur;
#else
                    words[row + wj] = cur;                                      // (every lane the same word to the same address)
#endif
                    if (wj == 0) first_updated = cur;
                    // (behind the row's last word these are zeros nobody reads: the row loop reloads)
                    cur = nxt;
                    up = (unsigned)__builtin_amdgcn_readfirstlane((int)pf_up); dn = (unsigned)__builtin_amdgcn_readfirstlane((int)pf_dn);
                    nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)pf_nx);
                }
            }
        }
        seed += (uint64_t)p * gamma;
#ifdef PTE_PROFILE_ISING_SECTIONS
        prof_loop = __builtin_readcyclecounter() - prof_t0;
#endif
    }
    __syncthreads();
    const long long spp = recompute();
    for (int wd = lane; wd < NW; wd += 64) wrow[wd] = words[wd];
    if (lane == 0) { e.suff[slot] = (double)spp; e.rng[2 * slot] = seed; }
    record_after_explore(e, cl, c, slot, lane, lp_before, (double)spp, 0.0);
#ifdef PTE_PROFILE_WAVES                   // debug builds only: per-wave start / end on the 100 MHz clock, placement
    if (lane == 0) {
        double *o = e.on_m2 + 2 * (e.d + 1) + 4 * cl;
        o[0] = (double)wave_t0; o[1] = (double)__builtin_amdgcn_s_memrealtime();
        o[2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4); o[3] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);
#ifdef PTE_PROFILE_ISING_SECTIONS
        o[0] = (double)prof_loop; o[1] = (double)prof_pass; o[2] = (double)prof_chase;
#endif
    }
#endif
}

}  // namespace pte
