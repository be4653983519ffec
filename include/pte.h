/* pte.h -- C ABI of the MI355X-native parallel-tempering engine ("pte").
 *
 * Drop-in boundary for the explore-then-swap hot path of Pigeons.jl v0.4.10.
 * The reference has NO C ABI on this path (it is Julia multiple dispatch); the
 * entry points below are what a Julia `ccall` glue (INTEGRATION.md,
 * pigeons.jl_amd/julia/PigeonsMI355X.jl) binds at the three dispatch hooks that
 * bracket the hot path, plus construction / adaptation / checkpoint hooks:
 *
 *   create_replicas(inputs, shared, source)      src/replicas/replicas.jl:65-98   -> pte_create
 *   explore!(pt, explorer, ::Val)                src/pt/pigeons.jl:82-132         -> pte_explore
 *   swap!(pair_swapper, replicas, swap_graph)    src/swap/swap.jl:6-39,106-126    -> pte_swap
 *   `while next_scan!(pt)` loop                  src/pt/pigeons.jl:49-52          -> pte_run_scans (fused)
 *   reduce_recorders!(pt, replicas)              src/recorders/recorders.jl:88-120 -> pte_reduce + pte_get_*
 *   adapt_tempering / discretize                 src/tempering/NonReversiblePT.jl:46-66 -> pte_set_schedule
 *   adapt_explorer(::AutoMALA, ...)              src/explorers/AutoMALA.jl:70-79   -> pte_set_explorer_adaptation
 *   checkpoint (Replica fields)                  src/pt/checkpoint.jl:110-145     -> pte_get_state / pte_set_state
 *
 * Conventions (precedent: the reference's only FFI, ext/PigeonsBridgeStanExt/interface.jl:118-183):
 *   - every call returns int: 0 = ok, != 0 = error; message via pte_last_error().
 *   - plain pointers and sizes; caller allocates all output arrays (host memory).
 *   - the engine owns all device memory; handles are freed by pte_destroy.
 *   - all calls on one handle come from one host thread; calls are synchronous at
 *     return (results visible to the host), asynchronous internally (HIP streams).
 *   - indices are 0-based: chain 0 = reference (beta = 0), chain N-1 = target;
 *     replica r = reference `replica_index` r+1.
 *   - floating point is IEEE binary64 throughout ("f64").
 *   - there is NO CPU fallback: on a machine without a HIP device pte_create fails.
 */
#ifndef PTE_H
#define PTE_H

#include <stdint.h>
#include "pte_rng_policy.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTE_ABI_VERSION 2

/* Device log-potential families (closed set; arbitrary Julia closures cannot run
 * on the GPU -- unsupported combinations make pte_create fail, and the caller
 * keeps the reference CPU path). */
enum {
    PTE_TARGET_MVN_SCALED_PRECISION = 0, /* toy_mvn_target: src/paths/ScaledPrecisionNormalPath.jl:5-48   */
    PTE_TARGET_TEST_SWAPPER         = 1, /* TestSwapper:    src/swap/pair_swapper.jl:100-149               */
    PTE_TARGET_FUNNEL               = 2, /* InterpolatingPath(normal ref, Neal's funnel)                   */
    PTE_TARGET_ISING                = 3  /* InterpolatingPath(Ising(0), Ising(beta)): examples/ising.jl        */
};
enum {
    PTE_EXPLORER_NONE     = 0,           /* `nothing` (TestSwapper)                                        */
    PTE_EXPLORER_TOY      = 1,           /* ToyExplorer:  src/explorers/ToyExplorer.jl:5-14                */
    PTE_EXPLORER_SLICE    = 2,           /* SliceSampler: src/explorers/SliceSampler.jl:8-237 -- the Float64-coordinate methods, on
                                            the scaled-precision MVN path (closed-form single-coordinate update, the fast kernels)
                                            and on the interpolated funnel path (the log potential evaluated in full per proposal,
                                            as slice_sample! does for any log_potential :105-118; dim <= 1024).  Not on the
                                            device: its Bool / Integer coordinate variants (:65-86,136-142,189) -- no device target
                                            has integer coordinates (Ising has its own explorer) -- and a GaussianReference under it;
                                            pte_create / pte_set_variational_reference refuse those, the caller keeps the CPU path */
    PTE_EXPLORER_AUTOMALA = 3,           /* AutoMALA:     src/explorers/AutoMALA.jl:29-294                 */
    PTE_EXPLORER_ISING_METROPOLIS = 4,   /* IsingMetropolis: examples/ising.jl:91-116 (n_steps in slice_n_passes) */
    PTE_EXPLORER_MALA     = 5            /* MALA:         src/explorers/MALA.jl:19-105 (am_* fields; step size fixed) */
};
enum {                                   /* Inputs.record (src/pt/Inputs.jl:57-62)                         */
    PTE_RECORD_ROUND_TRIP    = 1u << 0,  /* round_trip     src/recorders/RoundTripRecorder.jl              */
    PTE_RECORD_INDEX_PROCESS = 1u << 1,  /* index_process  src/recorders/recorder.jl:81                    */
    PTE_RECORD_ONLINE        = 1u << 2,  /* online / _transformed_online (target chain mean, variance)     */
    PTE_RECORD_TRACES        = 1u << 3,  /* traces: [state; log density] of the target chain per scan (src/recorders/recorder.jl:27,39-43; src/pt/pigeons.jl:116-125) */
    PTE_RECORD_TRACES_EXTENDED = 1u << 5, /* with PTE_RECORD_TRACES: inputs.extended_traces, every chain is traced (src/pt/pigeons.jl:116) */
    PTE_RECORD_ENERGY_AC1    = 1u << 4,  /* energy_ac1: per-chain covariance of the log density before / after explore! (recorder.jl:113; pigeons.jl:134-143) */
    PTE_RECORD_REFERENCE_REDUCTION = 1u << 6 /* swap_acceptance_pr / log_sum_ratio reduced the way the reference reduces them instead of by chain-keyed sums on the
                                              * device: every replica's own Mean (mu += (x - mu) / n, src/recorders/recorders.jl:88-130 over OnlineStats) and LogSum
                                              * (src/recorders/LogSum.jl:1-24) fitted in scan order, then merged over the binary tree on the replica index
                                              * (all_reduce_deterministically, src/mpi_utils/Entangler.jl:188-251).  The device logs the two log ratios of every active pair and
                                              * scan ([max_scans_per_round][n_chains][2] doubles); pte_reduce replays the fits and merges on the host, so the adapted schedule
                                              * is the reference's to the last bit instead of to 1e-11.  Needs PTE_RECORD_INDEX_PROCESS (who held the lower chain).  A
                                              * chain-shard replays the pairs whose lower chain it owns (log and index rows are local; the tree runs over the global
                                              * replica index).  With PTE_RECORD_TRACES the online statistics are rebuilt the same way, and AutoMALA's am_factors (the
                                              * exponent of every step-size search is logged: the step size adapts on their mean) always, with its reversibility_rate (the same log).
                                              * With PTE_RECORD_ENERGY_AC1 the pair of log densities around every explore step is logged too (by a launch of its own before and
                                              * after the explorer kernels: such an engine runs the launch-per-scan loop) and energy_ac1's correlation replayed.  Off by default: the values agree to ~1e-12 either way, and a round of 1024 x 1024 chain-scans logs 16 MB. */
};

enum {                                   /* pte_config.debug_kernel: which kernel generation explores (0 = the default)   */
    PTE_KERNEL_DEFAULT          = 0,
    PTE_KERNEL_SLICE_SEQUENTIAL = 1,     /* SliceSampler: the plain sequential kernel (exact fallback of the default one)  */
    /* 2, 5, 7 (and 8 = the default named explicitly): earlier SliceSampler generations, test build libpte_test.so only    */
    PTE_KERNEL_ISING_BITS       = 101,   /* IsingMetropolis: scalar bit-packed sweep, test build only                      */
    PTE_KERNEL_ISING_BYTES      = 102,   /* IsingMetropolis: scalar byte-lattice sweep (the kernel of base_length % 32 != 0) */
    /* a FLAG, or-ed to any of the above: pte_run_scans launches explore and swap per scan (the loop of rounds 1-4) even where the whole
     * call could run as ONE kernel with pairwise swap hand-shakes (pte_scan_loop_name); for A/B runs and the parity tests of the two forms */
    PTE_KERNEL_TWO_LAUNCHES     = 0x1000,
    /* a FLAG: where the one-kernel scan loop has a form with several consecutive chains per workgroup (their pairs shake hands through
     * LDS; pte_scan_loop_name ends in "_wg"), use the form with one chain per workgroup instead; bit-identical, for A/B runs and tests */
    PTE_KERNEL_SCAN_LOOP_ONE_CHAIN = 0x2000,
    PTE_KERNEL_FLAG_BITS        = 0x3000,
    /* FAULT INJECTION, test build libpte_test.so only (pte_create of the product library refuses them): what the one-kernel scan loop does when
     * its forward-progress assumptions break (tests/test_gpu_scan_loop_progress.py).
     *   DEAD_CHAIN      the wave of chain 7 leaves the loop silently before its third swap of every call: its partner's hand-shake times out after
     *                   3 s, every other wave sees the error word and leaves, pte_run_scans fails, the engine is poisoned (pte_scan_loop_stats)
     *   LATE_WORKGROUP  workgroup 3 of every scan-loop launch reaches the residency gate 80 ms late (as if the device had no room for it): the
     *                   launch aborts with nothing written and the call runs as explore + swap launches -- same results, no error */
    PTE_KERNEL_TEST_DEAD_CHAIN     = 0x4000,
    PTE_KERNEL_TEST_LATE_WORKGROUP = 0x8000,
    /* A/B reference, test build only: AutoMALA / MALA at 512 < dim <= 1024 on the ONE-wave kernel with sixteen blocks per lane (rounds 1-5; it
     * spills 250-300 VGPRs) instead of the four-waves-per-replica kernel k_explore_langevin_mw (round 6) -- bit-identical, tests/test_gpu_langevin_mw.py */
    PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE = 0x10000,
    PTE_KERNEL_TEST_BITS        = 0x1C000
};

/* Mirrors the fields of `Inputs` (src/pt/Inputs.jl:9-102) and of the explorer
 * structs that the hot path reads. */
typedef struct pte_config {
    uint32_t struct_size;        /* = sizeof(pte_config); checked by pte_create                            */
    uint32_t abi_version;        /* = PTE_ABI_VERSION                                                      */
    int32_t  device;             /* HIP device ordinal                                                     */
    int32_t  target;             /* PTE_TARGET_*                                                           */
    int32_t  explorer;           /* PTE_EXPLORER_*                                                         */
    uint32_t record_flags;       /* PTE_RECORD_*                                                           */
    int64_t  n_chains;           /* Inputs.n_chains (global N)                                             */
    int64_t  dim;                /* state dimension d                                                      */
    uint64_t seed;               /* Inputs.seed                                                            */
    int64_t  max_scans_per_round;/* capacity of the index-process buffer, 2^n_rounds                       */
    double   target_params[4];   /* MVN: {precision0, precision1}; TestSwapper: {accept pr};
                                    FUNNEL: {reference precision}; ISING: {beta}, dim = base_length^2; across the ABI
                                    (pte_get_state / pte_set_state / traces / online) a state is 0/1 spins as f64,
                                    row-major matrix[i,j] -> state[i*L + j]; in HBM and in boundary messages the
                                    lattice is bit-packed (8 KiB at base_length 256)                               */
    /* SliceSampler fields (SliceSampler.jl:8-20) */
    double   slice_w;
    int32_t  slice_p;
    int32_t  slice_n_passes;
    int32_t  slice_max_iter;
    /* AutoMALA fields (AutoMALA.jl:29-68) */
    int32_t  am_base_n_refresh;
    double   am_exponent_n_refresh;
    double   am_step_size;
    double   am_p0, am_p1;       /* MixDiagonalPreconditioner(p0, p1), Preconditioner.jl:43-52             */
    int32_t  am_preconditioner;  /* 0 identity, 1 diagonal, 2 mix-diagonal                                 */
    /* chain sharding: this engine owns chains [rank*N/world, (rank+1)*N/world)                           */
    int32_t  rank;
    int32_t  world_size;
    int32_t  explorer2;          /* Compose(explorer, explorer2), src/explorers/Compose.jl:5-19; PTE_EXPLORER_NONE = single explorer */
    /* StabilizedPT with inputs.variational == nothing (src/tempering/StabilizedPT.jl:37-51, src/swap/VariationalDEO.jl):
     * n_chains fixed-leg chains + n_chains_variational variational-leg chains, N = their sum (Inputs.jl:128); global
     * chain order = variational leg reference -> target, then the fixed leg target -> reference; references at both
     * ends, the two targets in the middle.  pte_set_schedule takes the N per-chain betas in that order
     * (concatenate_log_potentials, StabilizedPT.jl:67-69).  0 = one leg (NonReversiblePT).  Single engine only. */
    int64_t  n_chains_variational;
    /* Debug / bisecting: PTE_KERNEL_*.  The kernel is chosen by this field only -- the library never reads the
     * environment -- and pte_create fails on a value this build does not contain.  pte_kernel_name reports the choice. */
    int32_t  debug_kernel;
    int32_t  reserved0;
} pte_config;

typedef struct pte_engine pte_engine;

/* Fill `cfg` with the reference defaults (Inputs.jl:14-20, SliceSampler.jl:8-20, AutoMALA.jl:29-68). */
int pte_default_config(pte_config *cfg);

/* create_replicas: split RNG streams, initial states, chain = replica index, equally spaced schedule. */
int pte_create(const pte_config *cfg, pte_engine **out);
int pte_destroy(pte_engine *h);
const char *pte_last_error(const pte_engine *h);   /* h may be NULL: error of the last failed pte_create */

/* Schedule.grids (src/schedules/Schedule.jl) -> log_potentials along the ladder (discretize). */
int pte_set_schedule(pte_engine *h, const double *betas, int64_t n_chains);
int pte_get_schedule(const pte_engine *h, double *betas);

/* adapt_explorer(::AutoMALA): new step size and estimated_target_std_deviations (NULL = nothing). */
int pte_set_explorer_adaptation(pte_engine *h, double step_size, const double *target_std, int64_t dim);

/* One explore! over all local replicas for scan index `scan` (1-based within the round,
 * src/pt/Iterators.jl:37-47; AutoMALA skips the MH step when scan == 1). */
int pte_explore(pte_engine *h, int64_t scan);
/* One communicate!: DEO graph parity = iseven(scan) (src/swap/DEO.jl:12). */
int pte_swap(pte_engine *h, int64_t scan);
/* The fused scan loop: for s = first_scan .. first_scan+n_scans-1: explore!(s); communicate!(s).
 * On a chain-sharded engine (world_size > 1) this needs pte_comm_init and is collective over the ranks. */
int pte_run_scans(pte_engine *h, int64_t first_scan, int64_t n_scans);

/* reduce_recorders!: snapshot the round's accumulators to the host and reset them on the
 * device (incl. the round-trip state machines, RoundTripRecorder.jl:30-34). */
int pte_reduce(pte_engine *h);

/* Reduced recorders of the last pte_reduce. Array lengths in brackets (world_size == 1; a sharded
 * engine returns its local slice: pairs keyed by a local lower chain, chains [c0, c0+K)). */
int pte_get_swap_acceptance(const pte_engine *h, double *mean /*N-1*/, int64_t *n /*N-1*/);
int pte_get_log_sum_ratio(const pte_engine *h, double *up /*N-1*/, int64_t *up_n, double *dn /*N-1*/, int64_t *dn_n);
int pte_get_round_trip(const pte_engine *h, int64_t *n_tempered_restarts, int64_t *n_round_trips);
int pte_get_index_process(const pte_engine *h, int64_t *out /*N * n_scans, [replica][scan]*/, int64_t *n_scans);
int pte_get_explorer_stats(const pte_engine *h, double *acceptance_mean /*N*/, int64_t *acceptance_n /*N*/,
                           double *n_steps_sum /*N*/, int64_t *n_steps_n /*N*/);
int pte_get_automala_stats(const pte_engine *h, double *factor_mean /*N*/, int64_t *factor_n /*N*/,
                           double *reversibility_mean /*N*/, int64_t *reversibility_n /*N*/);
int pte_get_online(const pte_engine *h, double *mean /*d*/, double *variance /*d*/, int64_t *n);   /* two legs: both target chains, merged */
/* the (d+1)-th entry of the `online` sample extract_sample(state::Array, lp) = [state; lp(state)] (src/pt/state.jl:79) */
int pte_get_online_log_density(const pte_engine *h, double *mean, double *variance);
/* energy_ac1s(pt) (src/recorders/recorder.jl:156-173) for the local chains: cor[K] (NaN where n < 2), n[K],
 * moments[5K] = running (mean before, mean after, C_bb, C_ba, C_aa); NULL pointers are skipped. */
int pte_get_energy_ac1(const pte_engine *h, double *cor, int64_t *n, double *moments);
/* traces of the last round, out[scan][d+1] (two legs: out[scan][2][d+1], the variational leg's target first);
 * *n_scans = 0 on shards that do not own the target chain.
 * With PTE_RECORD_TRACES_EXTENDED: out[scan][K][d+1], the K local chains in chain order, on every shard. */
int pte_get_traces(const pte_engine *h, double *out, int64_t *n_scans);

/* Replica fields in replica order (src/replicas/Replica.jl:5-30): state [N*d], chain [N],
 * rng [2N] = (seed, gamma) of each SplittableRandom.  NULL pointers are skipped. */
/* GaussianReference (src/variational/GaussianReference.jl:4-74) taking over the reference end of the interpolated
 * path on the chains with uses[c] != 0 (update_reference! + update_path_variational, src/variational/variational.jl:28-41):
 * log density -0.5 log(2 pi s^2) - (x - m)^2 / (2 s^2) per coordinate, gradient -(x - m) / s^2, sample_iid! = randn * s + m.
 * Funnel target, single engine.  NULL mean / std deactivates. */
int pte_set_variational_reference(pte_engine *h, const double *mean /*d*/, const double *std_dev /*d*/, int64_t dim,
                                  const int32_t *uses /*N*/);
int pte_get_state(const pte_engine *h, double *state, int64_t *chain, uint64_t *rng);
int pte_set_state(pte_engine *h, const double *state, const int64_t *chain, const uint64_t *rng);

/* ---- chain-sharded engines (one engine per GPU; rank g owns chains [g*N/G, (g+1)*N/G)) -----------
 * The reference's distributed swap! (src/swap/swap.jl:79-102) exchanges 16-byte SwapStats with the
 * partner replica's rank and moves chain labels.  Here the shard boundary is between CHAINS, so only
 * the G-1 boundary pairs cross GPUs: their SwapStats are exchanged between the two phases below and,
 * iff the swap is accepted, the two replicas' payloads {state, sum x^2, rng, replica id, round-trip
 * state} trade places.  With these two-phase calls the HOST moves the bytes (any transport; the CPU tests use
 * gloo, loopback tests plain copies); the transport the library itself provides is pte_comm_* below.  side: 0 = pair (c0-1, c0), 1 = pair (c0+K-1, c0+K).
 * With world_size == 1 these calls are valid too (no boundary is ever active) and equal pte_swap. */
int pte_shard_info(const pte_engine *h, int64_t *first_chain, int64_t *n_local_chains, int64_t *n_local_pairs);
/* phase 1: swap_stat of every local chain (one rand per replica), index_process / round_trip records.
 * stats_out[4] = (log_ratio, uniform) of the lowest and of the highest local chain;
 * active_out[2] = whether the boundary pair on that side is a pair of this scan's DEO graph. */
int pte_swap_begin(pte_engine *h, int64_t scan, double *stats_out, int32_t *active_out);
/* phase 2: nbr_stats[4] = SwapStats received from the lower / upper neighbour (ignored where inactive);
 * decisions, recorders, chain relabelling of local pairs; accepted_out[2] = boundary swap accepted. */
int pte_swap_finish(pte_engine *h, int64_t scan, const double *nbr_stats, int32_t *accepted_out);
int64_t pte_boundary_payload_bytes(const pte_engine *h);            /* 8 * (sw + 6), sw = d (f64 coordinates) or ceil(ceil(d/32)/2) (bit-packed Ising lattice) */
int pte_boundary_export(pte_engine *h, int side, void *dst, int dst_is_device);
int pte_boundary_import(pte_engine *h, int side, const void *src, int src_is_device);
/* Device-resident, stream-ordered variant of the two phases (no host round trip per scan): the boundary
 * chain's SwapStat and -- speculatively -- its replica's payload are packed into one message of
 * pte_shard_message_bytes() = 8 (sw + 8) bytes per active side; the caller moves send -> neighbour's recv
 * with stream-ordered transfers (RCCL send/recv enqueued on pte_get_stream(), replacing the reference's
 * MPI transmits, src/mpi_utils/Entangler.jl:118-180) and the receiving engine decides and applies the
 * payload on the device iff the swap is accepted.  All four buffers are caller-owned device memory.
 *   pte_shard_scan_begin : explore!(scan) + phase 1 + pack; active_out[2] as for pte_swap_begin (host-computable)
 *   pte_shard_scan_finish: phase 2 from the received messages + conditional import
 *   pte_shard_sync       : synchronise, raise device errors, boundary_swaps_out[2] = applied swaps per side */
void *pte_get_stream(const pte_engine *h);                          /* hipStream_t of the engine */
int64_t pte_shard_message_bytes(const pte_engine *h);
int pte_shard_set_buffers(pte_engine *h, void *send_lo, void *recv_lo, void *send_hi, void *recv_hi);
int pte_shard_scan_begin(pte_engine *h, int64_t scan, int32_t *active_out);
int pte_shard_scan_finish(pte_engine *h, int64_t scan);
int pte_shard_sync(pte_engine *h, int64_t *boundary_swaps_out);
/* ---- transport behind the ABI -------------------------------------------------------------------
 * The reference keeps its communication inside the package (MPI: src/mpi_utils/Entangler.jl:118-180
 * `transmit!`, :188-251 `all_reduce_deterministically`; distributed swap! src/swap/swap.jl:79-102).  Here the
 * boundary exchange of a chain-sharded engine is RCCL point-to-point (ncclSend / ncclRecv in one group per scan,
 * over xGMI inside a node) enqueued on the engine's own HIP stream between the pack and the decide kernels:
 * no host synchronisation inside the scan loop, no collective on the data path.
 *   one process per GPU : rank 0 calls pte_comm_unique_id, the host hands the 128 bytes to every rank by any
 *                         means it has (MPI.bcast, a socket, a file), every rank calls pte_comm_init; from then
 *                         on pte_run_scans works on the sharded engine exactly as on a single one.
 *   one process, G GPUs : pte_group_run_scans drives G engines (rank g = engines[g]) from one host thread; the
 *                         messages move by stream-ordered device-to-device copies (peer copies over xGMI).
 * Message buffers are engine-owned unless the caller installed its own with pte_shard_set_buffers.
 * RCCL is mapped at run time (dlopen; $PTE_RCCL_LIB overrides the search), libpte.so does not link it. */
#define PTE_COMM_ID_BYTES 128
/* Which library the transport's entry points come from (the file, via dladdr) and what its ncclGetVersion reports -- for the
 * log line of a multi-GPU run.  $PTE_RCCL_LIB can name another library (tests: a stand-in that accepts two ranks on one
 * device); it is honoured only after pte_comm_allow_library_override(1), and a set variable WITHOUT that opt-in makes every
 * pte_comm_* call fail (a stale variable must not silently re-route the boundary traffic). */
int pte_comm_allow_library_override(int32_t allow);
int pte_comm_library(char *path_out, int64_t capacity, int32_t *version_out);
int pte_comm_unique_id(uint8_t *id_out /*PTE_COMM_ID_BYTES*/);
int pte_comm_init(pte_engine *h, const uint8_t *id /*PTE_COMM_ID_BYTES*/);   /* ncclCommInitRank(cfg.world_size, id, cfg.rank); collective */
int pte_comm_destroy(pte_engine *h);
/* kind: 0 none, 1 RCCL; n_ranks_seen: sum over the communicator of 1 (measured, not configured);
 * boundary_swaps[2]: boundary swaps applied on the device since pte_create, per side. NULLs are skipped. */
int pte_comm_info(pte_engine *h, int32_t *kind, int32_t *n_ranks_seen, int64_t *boundary_swaps);
/* Small host-side collectives over the same communicator, so that a host without MPI can bracket a round:
 * barrier; element-wise MAX / SUM of n doubles (op: 0 max, 1 sum); all-gather of `bytes` bytes per rank into
 * recv[world * bytes] in rank order (the per-round recorder slices -- recorders are keyed by chain / pair, so
 * concatenation in rank order is deterministic).  With world_size == 1 they are local no-ops / copies. */
int pte_comm_barrier(pte_engine *h);
int pte_comm_allreduce(pte_engine *h, double *inout, int64_t n, int32_t op);
int pte_comm_allgather(pte_engine *h, const void *send, int64_t bytes, void *recv);
/* G engines of ONE process, engines[g] = rank g of a world of G (same N, d, explorer): the fused scan loop. */
int pte_group_run_scans(pte_engine *const *engines, int32_t n_engines, int64_t first_scan, int64_t n_scans);

/* Name of the kernel that explores on this engine (e.g. "k_explore_slice8"), for profiles and bench.py. */
const char *pte_kernel_name(const pte_engine *h);

/* The form pte_run_scans takes on this engine -- the reference's `while next_scan!(pt)` loop, src/pt/pigeons.jl:46-55: "" = two launches
 * per scan (explore, swap); otherwise the ONE kernel that runs all the scans of a call (e.g. "k_scans_slice8": workgroup c holds chain c,
 * the DEO swap is a hand-shake between the two waves of a pair, no launch boundary and no grid-wide barrier per scan).  Chosen when one GPU
 * holds the whole ladder, every workgroup is resident at once, the explorer has such a kernel and pte_config.debug_kernel does not carry
 * PTE_KERNEL_TWO_LAUNCHES; results are bit-identical either way.  A name ending in "_wg" (AutoMALA / MALA) is the form with several
 * consecutive chains per workgroup, whose inner pairs shake hands through LDS (PTE_KERNEL_SCAN_LOOP_ONE_CHAIN selects the other form);
 * "k_scans_langevin_mw" (AutoMALA / MALA on the scaled-precision MVN path, 512 < dim <= 1024) gives a chain a 256-thread workgroup.
 * pte_timing_get(kernel = 4) times these launches. */
const char *pte_scan_loop_name(const pte_engine *h);
int pte_scan_loop_info(const pte_engine *h, int64_t *resident_limit, int64_t *timed_launches, int64_t *timed_scans);
/* Forward progress of that one kernel (round 6).  Its hand-shakes spin, so all its workgroups must be on the device together.  That is ENFORCED
 * inside the launch: no workgroup touches anything before every workgroup has arrived at a counter; if one is still missing after 50 ms (another
 * engine, stream or process holds compute units) all of them return with nothing written and pte_run_scans runs the same scans as explore + swap
 * launches -- same results, no error, the next 1, 2, 4, ... 256 calls skip the attempt.  So pte_run_scans cannot hang, as the reference's loop
 * cannot (src/pt/pigeons.jl:46-55).  What is left: a wave that dies or stays descheduled for 3 s AFTER all have arrived (the residency of a
 * started workgroup is the hardware's; whole-queue preemption between processes suspends and resumes all of a queue's waves together).  Then the
 * hand-shake times out, every other wave sees the error word and leaves at once, the call fails and the engine is POISONED: its replicas stopped at
 * different scans.  Every pte_* call on it fails with a message saying so, except pte_destroy and pte_set_state with state, chain and rng of all
 * replicas, which restores it (and discards the recorders of the round).
 *   *fused_calls: pte_run_scans calls that ran as one launch; *gate_aborts: launches that found a workgroup missing and fell back;
 *   *poisoned: 0 / 1.  NULLs are skipped. */
int pte_scan_loop_stats(const pte_engine *h, int64_t *fused_calls, int64_t *gate_aborts, int32_t *poisoned);

/* index process of the local slots: replica[scan][K], chain[scan][K] (global ids). */
int pte_get_index_process_shard(const pte_engine *h, int64_t *replica, int64_t *chain, int64_t *n_scans);
int pte_get_replica_ids(const pte_engine *h, int64_t *out /*K*/);

/* Measurement hooks (bench.py): per-kernel HIP-event timing accumulated on the engine's stream
 * over pte_run_scans calls since the last reset.  kernel: 0 = explore, 1 = swap (both empty while pte_run_scans runs as one fused launch:
 * 4 = that launch, pte_scan_loop_info says how many scans it held); 2 = k_init (create_replicas), timed once
 * at pte_create and not touched by pte_timing_reset; 3 = the boundary exchange of a chain-sharded engine (events around
 * ncclGroupStart .. ncclGroupEnd on the engine's stream: from "messages packed" to "messages landed", one sample per even scan).
 * enable: 0 off, 1 every kernel, 2 the explore kernels only (an event pair costs ~10 us of stream time per launch). */
int pte_timing_reset(pte_engine *h, int enable);
int pte_timing_get(const pte_engine *h, int kernel, double *total_ms, int64_t *launches);
/* the individual launch durations behind pte_timing_get (min / median / max of the timed region); out_ms may be NULL */
int pte_timing_get_samples(const pte_engine *h, int kernel, double *out_ms, int64_t capacity, int64_t *n_out);

/* The conventions of Julia's Random stdlib that no fixture from a live Julia pins yet (include/pte_rng_policy.h: the
 * ziggurat tail formula, the bit rand(rng, Bool) takes).  One word per device, read by every engine of this process on
 * that device; the default is PTE_RNG_POLICY_DEFAULT.  Never read from the environment. */
int pte_set_rng_policy(int32_t device, uint32_t policy);
int pte_get_rng_policy(int32_t device, uint32_t *policy);

/* RNG building blocks exposed for parity tests of the device samplers: fill `n` draws from the
 * stream (seed, gamma) on the device, in the reference's sequential order.
 * kind: 0 = rand (Float64 in [0,1)), 1 = randn, 2 = randexp, 3 = rand(rng, Bool) as 0.0 / 1.0.  Returns the advanced stream. */
int pte_test_rng_fill(int32_t device, uint64_t *seed_gamma /*2, in-out*/, int32_t kind, int64_t n, double *out);
/* The quotient procedure of the Langevin-family kernels, element by element (round 6): the preconditioner's diagonal and the funnel's sigma divide
 * thousands of values each, so a / b is evaluated as Markstein's correctly rounded q' = fma(fma(-q, b, a), r, q), q = a r, r = RN(1 / b), and as the
 * division itself wherever the theorem does not apply (quotient estimate outside [2^-900, 2^900], zero, non-finite; divisor with an extreme exponent
 * or an all-ones significand).  out[i] must equal the IEEE quotient a[i] / b[i] the reference computes (src/explorers/hamiltonian_dynamics.jl:60-76,
 * `./ M`) bit for bit; took_division[i] = 1 where the guards chose the division. */
int pte_test_quotient(int32_t device, const double *a, const double *b, int64_t n, double *out, int32_t *took_division);
/* sqr_norm of each row of x [rows][d] with the engine's fixed reduction tree. */
int pte_test_sqr_norm(int32_t device, const double *x, int64_t rows, int64_t d, double *out);

#ifdef __cplusplus
}
#endif
#endif /* PTE_H */
