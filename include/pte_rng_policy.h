/* pte_rng_policy.h -- the ONE place that names the conventions of Julia's Random stdlib which no fixture from a live
 * Julia pins yet (tests/golden/reference_pigeons.json, produced by tools/gen_golden.jl, decides them).  Both the product
 * library (pte_set_rng_policy) and the CPU oracle (po_set_rng_policy) read this header and nothing else about them.
 *
 *  PTE_RNG_TAIL_LOG1P   the ziggurat tails of randn / randexp (Random/src/normal.jl `randn_unlikely`, `randexp_unlikely`;
 *                       call sites in the reference: src/explorers/SliceSampler.jl:91, src/targets/toy_mvn_target.jl:11,20,
 *                       src/explorers/AutoMALA.jl:125) draw  -log(rand(rng))  in the Julia releases the reference's CI runs
 *                       (1.10 / 1.11, .github/workflows/CI.yml:25-27) -- policy bit clear, the default -- and
 *                       -log1p(-rand(rng))  in later sources (rand may return 0.0, log(0.0) = -Inf) -- bit set.
 *  PTE_RNG_BOOL_BIT(k)  rand(rng, Bool) (examples/ising.jl:53) is bit k of one UInt64 draw; k = 0 is `rand(rng, UInt64) % Bool`
 *                       (SplittableRandoms.jl 0.1 samples every bits type as `next % T`), the default; 63 would be the sign bit.
 *
 * Bits 0..7: flags; bits 8..13: the Bool bit index. */
#ifndef PTE_RNG_POLICY_H
#define PTE_RNG_POLICY_H

#define PTE_RNG_TAIL_LOG1P        (1u << 0)
#define PTE_RNG_BOOL_BIT(k)       (((unsigned)(k) & 63u) << 8)
#define PTE_RNG_POLICY_BOOL_BIT(policy)  (((policy) >> 8) & 63u)
#define PTE_RNG_POLICY_DEFAULT    0u
#define PTE_RNG_POLICY_VALID_MASK (PTE_RNG_TAIL_LOG1P | (63u << 8))

#endif /* PTE_RNG_POLICY_H */
