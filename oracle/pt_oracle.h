/* pt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the explore-then-swap hot path of Pigeons.jl v0.4.10
 * (reference tree at /root/reference, pure Julia).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product library (pigeons.jl_amd/csrc) never links, includes or calls it.
 *
 * PARITY UNPINNED: the reference cannot be executed in the build container (no
 * Julia) and its tests hold no bit-level golden vectors for this path, so this
 * restatement is pinned only by (i) the reference's RNG-free / analytic known
 * answers (tests/test_oracle_kat.py) and (ii) the public SplitMix64 vector and the
 * four recalled ziggurat table entries.  Third-party arithmetic restated from
 * published algorithms: SplittableRandoms.jl 0.1 (Java SplittableRandom),
 * Julia Random stdlib rand/randn/randexp and rand(rng, a:b) (SamplerRangeNDL), OnlineStatsBase 1.x Mean/Variance/Sum,
 * Interpolations.jl FritschCarlsonMonotonicInterpolation, LogExpFunctions logaddexp.
 *
 * All indices crossing this API are 0-based (chain 0 = reference, chain N-1 =
 * target; replica r = reference replica_index r+1).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG: SplittableRandoms.jl + Julia Random samplers ------------------- */
typedef struct { uint64_t seed, gamma; } po_rng;
po_rng   po_rng_new(uint64_t seed);          /* SplittableRandom(seed)            */
uint64_t po_rng_next_u64(po_rng *r);         /* rand(rng, UInt64)                  */
po_rng   po_rng_split(po_rng *r);            /* split(rng)                         */
double   po_rand(po_rng *r);                 /* rand(rng)    :: Float64 in [0,1)   */
double   po_randn(po_rng *r);                /* randn(rng)   (ziggurat)            */
double   po_randexp(po_rng *r);              /* randexp(rng) (ziggurat)            */
int      po_rand_bool_pub(po_rng *r);        /* rand(rng, Bool)                    */
/* the oracle's own ziggurat tables (built at load time in binary128): which = 0 ki, 1 wi, 2 fi, 3 ke, 4 we, 5 fe */
void     po_zig_table(int which, void *out /*256 x 8 bytes*/);
void     po_zig_install(int which, const void *in /*256 x 8 bytes*/);
/* include/pte_rng_policy.h: process-wide; returns != 0 on an invalid policy */
int      po_set_rng_policy(uint32_t policy);
int      po_set_libm_nudge(int mode);        /* tests only: exp / log of the Langevin family one ulp up (1), down (2), alternating (3) */
uint32_t po_get_rng_policy(void);

/* ---- numerics ------------------------------------------------------------ */
double po_sqr_norm(const double *x, int64_t d);       /* fixed pairwise tree       */
double po_logaddexp(double a, double b);
/* Fritsch-Carlson monotone cubic: build (m,c,dd from knots x,y; n>=2) / evaluate */
void   po_fc_build(const double *x, const double *y, int64_t n, double *m, double *c, double *dd);
double po_fc_eval(const double *x, const double *y, const double *m, const double *c,
                  const double *dd, int64_t n, double t);

/* ---- SliceSampler on Bool / Integer / mixed states (SliceSampler.jl:43-95, 128-142, 188-189) -----
 * Stand-alone (no po_pt): the device has no target with Bool / Integer coordinates and refuses them, so these methods are restated
 * behind a log-potential call-back, for tier 3.  State coordinates are doubles (Integer: exact below 2^53; Bool: 0.0 / 1.0). */
#include <stddef.h>
enum { PO_COORD_FLOAT64 = 0, PO_COORD_INTEGER = 1, PO_COORD_BOOL = 2 };
typedef double (*po_logpotential_fn)(const double *state, int64_t d, void *ctx);
typedef struct po_slice_params { double w; int32_t p, n_passes, max_iter; } po_slice_params;       /* SliceSampler.jl:8-20 */
typedef struct po_slice_stats { double acc_mean; int64_t acc_n; double steps_sum; int64_t steps_n; } po_slice_stats;   /* explorer_acceptance_pr (Mean), explorer_n_steps (Sum) */
int64_t po_rand_range(po_rng *r, int64_t a, int64_t b);          /* rand(rng, a:b), Int64: Random.SamplerRangeNDL */
/* one step!(::SliceSampler) on `state`; kind[c] in PO_COORD_* (NULL = all Float64); stats may be NULL; returns != 0 with a message in err */
int     po_slice_step_mixed(po_rng *rng, double *state, const int32_t *kind, int64_t d, const po_slice_params *h,
                            po_logpotential_fn lp, void *lp_ctx, po_slice_stats *stats, char *err, size_t errlen);

/* ---- parallel tempering --------------------------------------------------- */
enum { PO_TARGET_MVN = 0, PO_TARGET_TEST_SWAPPER = 1, PO_TARGET_FUNNEL = 2, PO_TARGET_ISING = 3 };
enum { PO_EXPLORER_NONE = 0, PO_EXPLORER_TOY = 1, PO_EXPLORER_SLICE = 2, PO_EXPLORER_AUTOMALA = 3, PO_EXPLORER_ISING = 4,
       PO_EXPLORER_MALA = 5 /* src/explorers/MALA.jl (am_* fields, fixed step size) */ };

typedef struct po_config {
    int64_t  n_chains;
    int64_t  dim;
    uint64_t seed;
    int32_t  target;            /* PO_TARGET_*                                     */
    int32_t  explorer;          /* PO_EXPLORER_*                                   */
    double   p0, p1;            /* MVN: precision0/1. TestSwapper: p0 = accept pr.
                                   Funnel: p0 = precision of the normal reference.
                                   Ising: p0 = beta of the target IsingLogPotential; dim = base_length^2,
                                   states are 0/1 spins stored as doubles; slice_n_passes = n_steps of IsingMetropolis */
    /* SliceSampler (src/explorers/SliceSampler.jl:8-20) */
    double   slice_w;
    int32_t  slice_p, slice_n_passes, slice_max_iter;
    /* AutoMALA (src/explorers/AutoMALA.jl:29-68) */
    int32_t  am_base_n_refresh;
    double   am_exponent_n_refresh;
    double   am_step_size;
    int32_t  am_preconditioner;  /* 0 identity, 1 diagonal, 2 mix-diagonal         */
    double   am_p0, am_p1;       /* mix proportions (1/3, 1/3)                     */
    /* recorders */
    int32_t  record_round_trip, record_index_process, record_online;
    int32_t  n_threads;          /* OpenMP threads over replicas in explore!       */
    /* chain sharding (test-only restatement of the build's multi-GPU protocol):
       this instance owns chains [rank*N/world, (rank+1)*N/world) and the replicas at them */
    int32_t  rank, world_size;
    /* traces (target chain [state; lp] per scan) and energy_ac1 */
    int32_t  record_traces, record_energy_ac1;
    /* Compose(explorer, explorer2) (src/explorers/Compose.jl:5-19); 0 = single explorer */
    int32_t  explorer2;
    /* StabilizedPT (src/tempering/StabilizedPT.jl) with the fixed reference on both legs
       (inputs.variational == nothing): n_chains fixed-leg chains + n_chains_variational variational-leg chains */
    int64_t  n_chains_variational;
    /* GaussianReference (src/variational/GaussianReference.jl:4-74) on the funnel path: 0 = none, else its
       first_tuning_round (default 6).  From that round on the variational leg (all chains when there is one leg)
       runs InterpolatingPath(GaussianReference(mean, std of the target chains), target). */
    int32_t  variational_first_tuning_round;
} po_config;

typedef struct po_pt po_pt;

void        po_default_config(po_config *cfg);
po_pt      *po_create(const po_config *cfg);
void        po_destroy(po_pt *pt);
const char *po_last_error(const po_pt *pt);

/* One full round (src/pt/pigeons.jl:17-19): 2^round scans, reduce, adapt.       */
int         po_run_round(po_pt *pt);
/* Pieces of the above, for timing and fine-grained tests. */
int         po_begin_round(po_pt *pt);                 /* round += 1, scan = 0     */
int         po_run_scans(po_pt *pt, int64_t n_scans);  /* explore!+communicate!    */
int         po_end_round(po_pt *pt);                   /* reduce_recorders!, adapt */
int64_t     po_round(const po_pt *pt);
void        po_set_round(po_pt *pt, int64_t round);   /* resume from a checkpoint: shared.iterators.round (src/pt/checkpoint.jl:19-54) */

/* ---- chain-sharded operation (world_size >= 1): same calls as include/pte.h's two-phase swap ---- */
int     po_shard_explore(po_pt *pt, int64_t scan);
int     po_shard_swap_begin(po_pt *pt, int64_t scan, double *stats_out /*4*/, int32_t *active_out /*2*/);
int     po_shard_swap_finish(po_pt *pt, int64_t scan, const double *nbr_stats /*4*/, int32_t *accepted_out /*2*/);
int64_t po_shard_payload_words(const po_pt *pt);                 /* d + 5 */
void    po_shard_export(po_pt *pt, int side, double *buf);
void    po_shard_import(po_pt *pt, int side, const double *buf);
int     po_shard_reduce(po_pt *pt);                              /* merge local recorders, reset */
void    po_shard_info(const po_pt *pt, int64_t *c0, int64_t *K, int64_t *n_pairs);
void    po_shard_replica_ids(const po_pt *pt, int64_t *out /*K*/);
int64_t po_shard_index_process(const po_pt *pt, int64_t *replica, int64_t *chain);   /* [scan][K]; returns n_scans */

/* State (replica order; local slot order for shards). rng: 2 words per replica (seed, gamma). */
void po_get_states(const po_pt *pt, double *x, int64_t *chain, uint64_t *rng);
void po_set_states(po_pt *pt, const double *x, const int64_t *chain, const uint64_t *rng);   /* NULLs are skipped */
void po_get_schedule(const po_pt *pt, double *betas);
void po_set_schedule(po_pt *pt, const double *betas);

/* Reduced recorders of the last completed round. */
void    po_get_swap_pr(const po_pt *pt, double *mean, int64_t *n);              /* N-1      */
void    po_get_log_sum_ratio(const po_pt *pt, double *up, int64_t *up_n,
                             double *dn, int64_t *dn_n);                         /* N-1 each */
void    po_get_round_trip(const po_pt *pt, int64_t *restarts, int64_t *trips);
int64_t po_get_index_process(const po_pt *pt, int64_t *out);  /* [replica][scan]; returns n_scans */
void    po_get_explorer_stats(const po_pt *pt, double *acc_mean, int64_t *acc_n,
                              double *steps_sum, int64_t *steps_n);              /* N each   */
void    po_get_am_stats(const po_pt *pt, double *factor_mean, int64_t *factor_n,
                        double *rev_mean, int64_t *rev_n);                       /* N each   */
int64_t po_get_online(const po_pt *pt, double *mean, double *var);               /* d each; returns n */
void    po_get_online_lp(const po_pt *pt, double *mean, double *var, int64_t *n); /* entry d+1 of `online`: the log density */
void    po_get_energy_ac1(const po_pt *pt, double *cor /*N*/, int64_t *n /*N*/, double *raw /*5N or NULL*/);
int64_t po_get_traces(const po_pt *pt, double *out /*[scan][d+1]*/);              /* returns the number of scans */
void    po_get_stepping_stone(const po_pt *pt, double *pair);                    /* 2        */
double  po_get_global_barrier(const po_pt *pt);
double  po_get_global_barrier_variational(const po_pt *pt);
int     po_get_variational(const po_pt *pt, double *mean, double *std);   /* returns 1 when the reference is active */   /* StabilizedPT.jl:117-119 */
double  po_cumulative_barrier(const po_pt *pt, double beta);
double  po_get_step_size(const po_pt *pt);
int64_t po_get_target_std(const po_pt *pt, double *out);                         /* d; returns 0 if `nothing` */
void    po_set_explorer_adaptation(po_pt *pt, double step_size, const double *target_std);

#ifdef __cplusplus
}
#endif
#endif
