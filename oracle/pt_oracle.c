/* pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pt_oracle.h for the scope and the "PARITY UNPINNED" statement.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Build: see oracle/Makefile (-ffp-contract=off:
 * Julia never contracts a*b+c).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"
#include "../include/pte_rng_policy.h"

#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ========================================================================== */
/* RNG                                                                        */
/* ========================================================================== */

/* SplittableRandoms.jl 0.1 == Java 8 SplittableRandom (SplitMix64).  Not under
 * /root/reference; call sites src/replicas/replicas.jl:88, src/utils/misc.jl:28-30. */
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t mix_gamma(uint64_t z) {
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL;
    z = (z ^ (z >> 33)) | 1ULL;
    int n = __builtin_popcountll(z ^ (z >> 1));
    return (n < 24) ? (z ^ 0xaaaaaaaaaaaaaaaaULL) : z;
}
po_rng po_rng_new(uint64_t seed) {
    po_rng r = { seed, 0x9e3779b97f4a7c15ULL };
    return r;
}
uint64_t po_rng_next_u64(po_rng *r) {
    r->seed += r->gamma;
    return mix64(r->seed);
}
po_rng po_rng_split(po_rng *r) {
    po_rng c;
    c.seed = po_rng_next_u64(r);       /* nextLong()            */
    r->seed += r->gamma;               /* nextSeed()            */
    c.gamma = mix_gamma(r->seed);
    return c;
}

/* Julia Random stdlib, generic AbstractRNG path: rand(rng) = CloseOpen12 - 1.0
 * built from the low 52 bits of one UInt64 draw. Call sites e.g.
 * src/swap/pair_swapper.jl:46, src/explorers/SliceSampler.jl:110,130,188. */
#define MASK52 0x000fffffffffffffULL
static inline double u52_to_unit(uint64_t u) {
    union { uint64_t u; double d; } v;
    v.u = (u & MASK52) | 0x3ff0000000000000ULL;
    return v.d - 1.0;
}
double po_rand(po_rng *r) { return u52_to_unit(po_rng_next_u64(r)); }

/* The 256-layer ziggurat tables of Random/src/normal.jl (`ki, wi, fi, ke, we, fe`; 0-based here: ZIG_KI[i] = ki[i+1]).
 * The oracle does NOT read the product's generated header (pigeons.jl_amd/csrc/zig_tables.h, 60-digit mpmath): it derives
 * the tables when the library is loaded, by its own route -- the randmtzig recurrence in IEEE binary128 (libquadmath),
 * section areas computed from R -- so that a wrong entry in either place shows up as a difference
 * (tests/test_zig_tables.py compares all 6 x 256 entries; the same test documents where numpy's embedded
 * double-precision tables of the same construction differ in the last bits).  PARITY vs Julia's literal tables: UNPINNED
 * until tools/gen_golden.jl has been run (it dumps them; tools/import_tables.py installs them). */
static uint64_t ZIG_KI[256], ZIG_KE[256];
static double ZIG_WI[256], ZIG_FI[256], ZIG_WE[256], ZIG_FE[256];
static uint64_t ZIG_DERIVED[6][256];      /* the binary128 derivation, kept when po_zig_install replaces the active tables */
static double ZIG_NOR_R, ZIG_NOR_INV_R, ZIG_EXP_R;
static uint32_t g_rng_policy = PTE_RNG_POLICY_DEFAULT;

__attribute__((constructor)) static void zig_build(void) {
    typedef __float128 q;
    const q RN = strtoflt128("3.6541528853610088", NULL), RE = strtoflt128("7.69711747013104972", NULL);
    ZIG_NOR_R = (double)RN; ZIG_NOR_INV_R = 1.0 / ZIG_NOR_R; ZIG_EXP_R = (double)RE;
    {   /* normal: f(x) = exp(-x^2/2), mantissa 2^51, area = R f(R) + sqrt(pi/2) erfc(R / sqrt 2) */
        const q M = ldexpq(1.0Q, 51);
        const q area = RN * expq(-RN * RN / 2) + sqrtq(M_PIq / 2) * erfcq(RN / sqrtq(2.0Q));
        q x1 = RN, fx1 = expq(-x1 * x1 / 2);
        ZIG_WI[255] = (double)(x1 / M); ZIG_FI[255] = (double)fx1;
        ZIG_KI[0] = (uint64_t)floorq(x1 * fx1 / area * M);
        ZIG_WI[0] = (double)(area / fx1 / M); ZIG_FI[0] = 1.0;
        for (int i = 254; i > 0; i--) {
            q x = sqrtq(-2 * logq(area / x1 + fx1));
            ZIG_KI[i + 1] = (uint64_t)floorq(x / x1 * M);
            ZIG_WI[i] = (double)(x / M);
            fx1 = expq(-x * x / 2);
            ZIG_FI[i] = (double)fx1;
            x1 = x;
        }
        ZIG_KI[1] = 0;
    }
    {   /* exponential: f(x) = exp(-x), mantissa 2^52, area = R f(R) + f(R) */
        const q M = ldexpq(1.0Q, 52);
        const q area = RE * expq(-RE) + expq(-RE);
        q x1 = RE, fx1 = expq(-x1);
        ZIG_WE[255] = (double)(x1 / M); ZIG_FE[255] = (double)fx1;
        ZIG_KE[0] = (uint64_t)floorq(x1 * fx1 / area * M);
        ZIG_WE[0] = (double)(area / fx1 / M); ZIG_FE[0] = 1.0;
        for (int i = 254; i > 0; i--) {
            q x = -logq(area / x1 + fx1);
            ZIG_KE[i + 1] = (uint64_t)floorq(x / x1 * M);
            ZIG_WE[i] = (double)(x / M);
            fx1 = expq(-x);
            ZIG_FE[i] = (double)fx1;
            x1 = x;
        }
        ZIG_KE[1] = 0;
    }
    memcpy(ZIG_DERIVED[0], ZIG_KI, 2048); memcpy(ZIG_DERIVED[1], ZIG_WI, 2048); memcpy(ZIG_DERIVED[2], ZIG_FI, 2048);
    memcpy(ZIG_DERIVED[3], ZIG_KE, 2048); memcpy(ZIG_DERIVED[4], ZIG_WE, 2048); memcpy(ZIG_DERIVED[5], ZIG_FE, 2048);
}
/* which: 0 ki, 1 wi, 2 fi, 3 ke, 4 we, 5 fe = the ACTIVE tables; 8 + which = the binary128 derivation; out: 256 x 8 bytes */
void po_zig_table(int which, void *out) {
    const void *src[6] = { ZIG_KI, ZIG_WI, ZIG_FI, ZIG_KE, ZIG_WE, ZIG_FE };
    if (which >= 0 && which < 6) memcpy(out, src[which], 256 * 8);
    else if (which >= 8 && which < 14) memcpy(out, ZIG_DERIVED[which - 8], 256 * 8);
}
/* tools/import_tables.py path: install tables taken from a live Julia (tests/golden/reference_pigeons.json) */
void po_zig_install(int which, const void *in) {
    void *dst[6] = { ZIG_KI, ZIG_WI, ZIG_FI, ZIG_KE, ZIG_WE, ZIG_FE };
    if (which >= 0 && which < 6) memcpy(dst[which], in, 256 * 8);
}
/* Sensitivity hook (tests only, tests/test_gpu_benchmarked_shapes.py): exp / log of the Langevin family -- the funnel density and the
 * AutoMALA / MALA bounds and acceptance ratios -- nudged by one ulp: 0 as libm returns them, 1 one ulp up, 2 one ulp down, 3 up and
 * down alternating per call.  glibc (here) and ocml (device) exp / log are both within an ulp of the truth but not bit-identical; the
 * test shows that the device's residual against this oracle at BASELINE configs[2] lies inside the band this nudge opens. */
static int g_libm_nudge = 0;
static _Thread_local unsigned g_libm_calls = 0;
int po_set_libm_nudge(int mode) { if (mode < 0 || mode > 3) return 1; g_libm_nudge = mode; return 0; }
static inline double libm_nudge(double v) {
    if (g_libm_nudge == 0 || !isfinite(v)) return v;
    const int up = g_libm_nudge == 1 || (g_libm_nudge == 3 && ((g_libm_calls++) & 1u));
    return nextafter(v, up ? INFINITY : -INFINITY);
}
static inline double po_exp(double x) { return libm_nudge(exp(x)); }
static inline double po_log(double x) { return libm_nudge(log(x)); }

int po_set_rng_policy(uint32_t policy) {
    if (policy & ~PTE_RNG_POLICY_VALID_MASK) return 1;
    g_rng_policy = policy;
    return 0;
}
uint32_t po_get_rng_policy(void) { return g_rng_policy; }
/* the tail draw of both ziggurats: -log(rand) or -log1p(-rand), include/pte_rng_policy.h */
static inline double zig_tail_neglog(double u) { return (g_rng_policy & PTE_RNG_TAIL_LOG1P) ? -log1p(-u) : -log(u); }

/* randn: Random/src/normal.jl `randn` + `randn_unlikely` (256-layer ziggurat). */
double po_randn(po_rng *r) {
    for (;;) {
        uint64_t u = po_rng_next_u64(r) & MASK52;
        int64_t rabs = (int64_t)(u >> 1);
        int idx = (int)(rabs & 0xFF);
        double x = (double)((u & 1) ? -rabs : rabs) * ZIG_WI[idx];
        if ((uint64_t)rabs < ZIG_KI[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = ZIG_NOR_INV_R * zig_tail_neglog(po_rand(r));
                double yy = zig_tail_neglog(po_rand(r));
                if (yy + yy > xx * xx)
                    return ((rabs >> 8) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
            }
        } else if ((ZIG_FI[idx - 1] - ZIG_FI[idx]) * po_rand(r) + ZIG_FI[idx] < exp(-0.5 * x * x)) {
            return x;
        }
        /* else: return randn(rng) -> loop */
    }
}

/* randexp: Random/src/normal.jl `randexp` + `randexp_unlikely`. */
double po_randexp(po_rng *r) {
    for (;;) {
        uint64_t ri = po_rng_next_u64(r) & MASK52;
        int idx = (int)(ri & 0xFF);
        double x = (double)ri * ZIG_WE[idx];
        if (ri < ZIG_KE[idx]) return x;
        if (idx == 0) return ZIG_EXP_R + zig_tail_neglog(po_rand(r));
        if ((ZIG_FE[idx - 1] - ZIG_FE[idx]) * po_rand(r) + ZIG_FE[idx] < exp(-x)) return x;
    }
}

/* ========================================================================== */
/* numerics                                                                   */
/* ========================================================================== */

static int64_t next_pow2(int64_t n) { int64_t p = 1; while (p < n) p <<= 1; return p; }

/* sqr_norm(x) = sum(abs2, x)  (src/utils/misc.jl:10).  Julia's reduction order
 * is compiler dependent (@simd); the build fixes ONE association, used by the
 * oracle and by every HIP kernel: the balanced binary tree over the leaves
 * x_i^2 in natural order, zero-padded to the next power of two
 * (node[i] = node[2i] + node[2i+1]).  x + 0.0 is exact, so padding is inert. */
double po_sqr_norm(const double *x, int64_t d) {
    /* The balanced tree evaluated without a scratch array: 8-leaf subtrees in registers, their sums merged through
     * a binary-counter stack (level l holds a completed subtree of 8 * 2^l leaves).  Zero padding never has to be
     * materialised: a subtree of zeros adds 0.0, which is exact, so a partial right edge just moves up the tree.
     * Same association as node[i] = node[2i] + node[2i+1] -- tests/test_oracle_kat.py::test_sqr_norm_tree pins the bits. */
    if (d <= 0) return 0.0;
    double stack[64];
    int level[64];
    int top = 0;
    int64_t i = 0;
    for (; i + 8 <= d; i += 8) {
        const double *v = x + i;
        double s = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                   ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        int l = 0;
        while (top > 0 && level[top - 1] == l) { s = stack[--top] + s; l++; }
        stack[top] = s; level[top] = l; top++;
    }
    if (i < d) {                                   /* ragged last block: missing leaves are 0.0 */
        double q[8];
        for (int k = 0; k < 8; k++) q[k] = (i + k < d) ? x[i + k] * x[i + k] : 0.0;
        double s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        int l = 0;
        while (top > 0 && level[top - 1] == l) { s = stack[--top] + s; l++; }
        stack[top] = s; level[top] = l; top++;
    }
    double acc = stack[--top];                     /* the right edge climbs (adding exact zeros) until it meets its left sibling */
    while (top > 0) acc = stack[--top] + acc;
    return acc;
}

/* LogExpFunctions.logaddexp (0.3.x): max + log1pexp(-|x-y|); used by
 * src/recorders/LogSum.jl:11,16. */
double po_logaddexp(double x, double y) {
    double delta = (x == y) ? 0.0 : fabs(x - y);
    double m = (x > y) ? x : y;
    double nd = -delta;
    double t = (nd <= -37.0) ? exp(nd) : log1p(exp(nd));
    return m + t;
}

/* Interpolations.jl FritschCarlsonMonotonicInterpolation (monotonic.jl), used by
 * src/tempering/adaptation.jl:61,85.  m: n tangents; c,dd: n-1 coefficients. */
void po_fc_build(const double *x, const double *y, int64_t n, double *m, double *c, double *dd) {
    double *D = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n - 1 : 1));
    for (int64_t k = 0; k < n - 1; k++) {
        double Dk = (y[k + 1] - y[k]) / (x[k + 1] - x[k]);
        D[k] = Dk;
        if (k == 0) m[k] = Dk;
        else if (D[k - 1] * Dk <= 0.0) m[k] = 0.0;
        else m[k] = (D[k - 1] + Dk) / 2.0;
    }
    m[n - 1] = D[n - 2];
    for (int64_t k = 0; k < n - 1; k++) {
        double Dk = D[k];
        if (Dk == 0.0) { m[k] = 0.0; m[k + 1] = 0.0; continue; }
        double al = m[k] / Dk, be = m[k + 1] / Dk;
        double tau = 3.0 / sqrt(al * al + be * be);
        if (tau < 1.0) { m[k] = tau * al * Dk; m[k + 1] = tau * be * Dk; }
    }
    for (int64_t k = 0; k < n - 1; k++) {
        double xd = x[k + 1] - x[k];
        c[k] = (3.0 * D[k] - 2.0 * m[k] - m[k + 1]) / xd;
        dd[k] = (m[k] + m[k + 1] - 2.0 * D[k]) / (xd * xd);
    }
    free(D);
}
double po_fc_eval(const double *x, const double *y, const double *m, const double *c,
                  const double *dd, int64_t n, double t) {
    /* k = searchsortedfirst(knots, t); if k > 1: k -= 1  (1-based) */
    int64_t k = 0;
    while (k < n && x[k] < t) k++;      /* 0-based index of first knot >= t */
    if (k > 0) k -= 1;
    if (k > n - 2) k = n - 2;
    double xd = t - x[k];
    return y[k] + m[k] * xd + c[k] * xd * xd + dd[k] * xd * xd * xd;
}

/* ========================================================================== */
/* OnlineStatsBase statistics (Mean / Sum / Variance / LogSum) and recorders   */
/* ========================================================================== */
typedef struct { double mu; int64_t n; } po_mean;
typedef struct { double value; int64_t n; } po_logsum;
typedef struct { double sum; int64_t n; } po_sum;
typedef struct { double s2, mu; int64_t n; } po_var;

static inline void mean_fit(po_mean *o, double x) {      /* mu += (1/n)(x - mu) */
    o->n += 1;
    o->mu = o->mu + (1.0 / (double)o->n) * (x - o->mu);
}
static inline void mean_merge(po_mean *a, const po_mean *b) {
    if (b->n == 0) return;                /* GroupBy: key absent in b            */
    if (a->n == 0) { *a = *b; return; }   /* GroupBy: key absent in a -> copy    */
    a->n += b->n;
    a->mu = a->mu + ((double)b->n / (double)a->n) * (b->mu - a->mu);
}
static inline void sum_fit(po_sum *o, double x) { o->n += 1; o->sum += x; }
static inline void sum_merge(po_sum *a, const po_sum *b) {
    if (b->n == 0) return;
    if (a->n == 0) { *a = *b; return; }
    a->n += b->n; a->sum += b->sum;
}
static inline void var_fit(po_var *o, double x) {
    double mu = o->mu;
    o->n += 1;
    double g = 1.0 / (double)o->n;
    o->mu = o->mu + g * (x - o->mu);
    o->s2 = o->s2 + g * ((x - o->mu) * (x - mu) - o->s2);
}
static inline void var_merge(po_var *a, const po_var *b) {
    if (b->n == 0) return;
    if (a->n == 0) { *a = *b; return; }
    a->n += b->n;
    double g = (double)b->n / (double)a->n;
    double delta = b->mu - a->mu;
    a->s2 = (a->s2 + g * (b->s2 - a->s2)) + delta * delta * g * (1.0 - g);
    a->mu = a->mu + g * (b->mu - a->mu);
}
static inline double var_value(const po_var *o) {
    return o->n > 1 ? o->s2 * ((double)o->n / (double)(o->n - 1)) : 1.0;
}
/* src/recorders/LogSum.jl:1-24 */
static inline void logsum_fit(po_logsum *o, double y) { o->value = po_logaddexp(o->value, y); o->n += 1; }
static inline void logsum_merge(po_logsum *a, const po_logsum *b) {
    if (b->n == 0) return;
    if (a->n == 0) { *a = *b; return; }
    a->value = po_logaddexp(a->value, b->value);
    a->n += b->n;
}

/* src/recorders/RoundTripRecorder.jl:4-54 */
/* OnlineStatsBase 1.x `CovMatrix(2)` (third-party, not under /root/reference; the reference pins
 * OnlineStatsBase = "1", Project.toml:91): b = running mean, A = running mean of x x' (upper triangle),
 * both with the EqualWeight smoothing a += (1/n)(x - a); merge smooths with n2/(n1+n2);
 * value = (A - b b') n/(n-1); cor = D^-1/2 value D^-1/2. */
typedef struct { int64_t n; double b[2]; double A[3]; } po_cov2;
static inline void cov2_fit(po_cov2 *o, double x0, double x1) {
    o->n += 1;
    const double g = 1.0 / (double)o->n;
    o->b[0] += g * (x0 - o->b[0]); o->b[1] += g * (x1 - o->b[1]);
    o->A[0] += g * (x0 * x0 - o->A[0]); o->A[1] += g * (x0 * x1 - o->A[1]); o->A[2] += g * (x1 * x1 - o->A[2]);
}
static inline void cov2_merge(po_cov2 *a, const po_cov2 *b) {
    if (b->n == 0) return;
    a->n += b->n;
    const double g = (double)b->n / (double)a->n;
    for (int i = 0; i < 3; i++) a->A[i] += g * (b->A[i] - a->A[i]);
    for (int i = 0; i < 2; i++) a->b[i] += g * (b->b[i] - a->b[i]);
}
static inline double cov2_cor12(const po_cov2 *o) {      /* cor(o)[1,2], energy_ac1s (recorder.jl:162-173) */
    const double bes = (double)o->n / (double)(o->n - 1);
    const double c00 = (o->A[0] - o->b[0] * o->b[0]) * bes, c01 = (o->A[1] - o->b[0] * o->b[1]) * bes,
                 c11 = (o->A[2] - o->b[1] * o->b[1]) * bes;
    const double v0 = 1.0 / sqrt(c00), v1 = 1.0 / sqrt(c11);
    return (c01 * v1) * v0;
}

typedef struct { int64_t n_tempered_restarts, n_round_trips, state; } po_round_trip;
static inline void round_trip_record(po_round_trip *r, int is_ref, int is_target) {
    if (r->state == 0 && is_ref) r->state = 1;
    else if (r->state == 1 && is_target) { r->state = 2; r->n_tempered_restarts += 1; }
    else if (r->state == 2 && is_ref) { r->state = 1; r->n_round_trips += 1; }
}

/* One replica's `recorders` NamedTuple (src/recorders/recorders.jl:37-70), as
 * dense arrays keyed by chain / pair; a key is "absent" iff its n == 0. */
typedef struct {
    po_mean   *swap_pr;     /* [N-1] key (c,c+1)   src/recorders/recorder.jl:60   */
    po_logsum *lsr_up;      /* [N-1] key (c,c+1)   recorder.jl:87                 */
    po_logsum *lsr_dn;      /* [N-1] key (c+1,c)                                  */
    po_mean   *expl_acc;    /* [N]   explorer_acceptance_pr  recorder.jl:67       */
    po_sum    *expl_steps;  /* [N]   explorer_n_steps        recorder.jl:74       */
    po_mean   *am_factors;  /* [N]   src/explorers/AutoMALA.jl:277                */
    po_mean   *rev_rate;    /* [N]   AutoMALA.jl:294                              */
    po_round_trip rt;
    int64_t   *ip; int64_t ip_len, ip_cap;   /* index_process[replica_index]      */
    po_mean   *on_mean;     /* [d+1] _transformed_online / online, target chain; entry d = log density
                               (extract_sample(state::Array, lp) = [state; lp(state)], src/pt/state.jl:79) */
    po_var    *on_var;      /* [d+1]                                              */
    po_cov2   *eac;         /* [N] energy_ac1 = GroupBy(Int, CovMatrix(2)), recorder.jl:113 */
} po_recorders;

static void rec_alloc(po_recorders *r, int64_t N, int64_t d) {
    memset(r, 0, sizeof(*r));
    int64_t np = N > 1 ? N - 1 : 1;
    r->swap_pr = (po_mean *)calloc((size_t)np, sizeof(po_mean));
    r->lsr_up = (po_logsum *)calloc((size_t)np, sizeof(po_logsum));
    r->lsr_dn = (po_logsum *)calloc((size_t)np, sizeof(po_logsum));
    r->expl_acc = (po_mean *)calloc((size_t)N, sizeof(po_mean));
    r->expl_steps = (po_sum *)calloc((size_t)N, sizeof(po_sum));
    r->am_factors = (po_mean *)calloc((size_t)N, sizeof(po_mean));
    r->rev_rate = (po_mean *)calloc((size_t)N, sizeof(po_mean));
    r->on_mean = (po_mean *)calloc((size_t)(d + 1), sizeof(po_mean));
    r->on_var = (po_var *)calloc((size_t)(d + 1), sizeof(po_var));
    r->eac = (po_cov2 *)calloc((size_t)N, sizeof(po_cov2));
}
static void rec_empty(po_recorders *r, int64_t N, int64_t d) {
    int64_t np = N > 1 ? N - 1 : 1;
    memset(r->swap_pr, 0, sizeof(po_mean) * (size_t)np);
    for (int64_t i = 0; i < np; i++) {
        r->lsr_up[i].value = -INFINITY; r->lsr_up[i].n = 0;
        r->lsr_dn[i].value = -INFINITY; r->lsr_dn[i].n = 0;
    }
    memset(r->expl_acc, 0, sizeof(po_mean) * (size_t)N);
    memset(r->expl_steps, 0, sizeof(po_sum) * (size_t)N);
    memset(r->am_factors, 0, sizeof(po_mean) * (size_t)N);
    memset(r->rev_rate, 0, sizeof(po_mean) * (size_t)N);
    memset(&r->rt, 0, sizeof(r->rt));
    r->ip_len = 0;
    memset(r->on_mean, 0, sizeof(po_mean) * (size_t)(d + 1));
    memset(r->on_var, 0, sizeof(po_var) * (size_t)(d + 1));
    memset(r->eac, 0, sizeof(po_cov2) * (size_t)N);
}
static void rec_free(po_recorders *r) {
    free(r->swap_pr); free(r->lsr_up); free(r->lsr_dn); free(r->expl_acc); free(r->expl_steps);
    free(r->am_factors); free(r->rev_rate); free(r->ip); free(r->on_mean); free(r->on_var); free(r->eac);
}
/* merge_recorders (src/recorders/recorders.jl:122-130): a <- merge(a, b).
 * index_process dicts have disjoint keys (one per replica) and are collected by
 * the caller; the merged RoundTripRecorder has state 0 (RoundTripRecorder.jl:36-41). */
static void rec_merge(po_recorders *a, const po_recorders *b, int64_t N, int64_t d) {
    for (int64_t i = 0; i + 1 < N; i++) {
        mean_merge(&a->swap_pr[i], &b->swap_pr[i]);
        logsum_merge(&a->lsr_up[i], &b->lsr_up[i]);
        logsum_merge(&a->lsr_dn[i], &b->lsr_dn[i]);
    }
    for (int64_t i = 0; i < N; i++) {
        mean_merge(&a->expl_acc[i], &b->expl_acc[i]);
        sum_merge(&a->expl_steps[i], &b->expl_steps[i]);
        mean_merge(&a->am_factors[i], &b->am_factors[i]);
        mean_merge(&a->rev_rate[i], &b->rev_rate[i]);
        cov2_merge(&a->eac[i], &b->eac[i]);
    }
    a->rt.n_tempered_restarts += b->rt.n_tempered_restarts;
    a->rt.n_round_trips += b->rt.n_round_trips;
    a->rt.state = 0;
    for (int64_t i = 0; i <= d; i++) {
        mean_merge(&a->on_mean[i], &b->on_mean[i]);
        var_merge(&a->on_var[i], &b->on_var[i]);
    }
}

/* ========================================================================== */
/* PT structures                                                              */
/* ========================================================================== */
typedef struct { double log_ratio, uniform; } swap_stat_t_fwd;
/* src/replicas/Replica.jl:5-30 */
typedef struct {
    double  *state;
    int64_t  chain;          /* 0-based */
    po_rng   rng;
    po_recorders rec;
    int64_t  replica_index;  /* 0-based */
    int64_t  aux;            /* Ising: cached sum_pair_products (IsingState, examples/ising.jl:18-22) */
    /* AutoMALA scratch (src/explorers/Augmentation.jl buffers) */
    double  *buf;
} po_replica;

struct po_pt {
    po_config cfg;
    int64_t N, d;
    int64_t K, c0;               /* shard: local chains [c0, c0+K); K == N, c0 == 0 when unsharded */
    swap_stat_t_fwd *shard_stat; /* [K] SwapStats of the local chains between the two swap phases   */
    int64_t *shard_ip_replica, *shard_ip_chain; int64_t shard_ip_len, shard_ip_cap;  /* [scan][K]   */
    po_replica *replicas;        /* by replica_index (by local slot for shards)   */
    int64_t *replica_of_chain;   /* the sorted Vector{Replica} view (swap.jl:7)   */
    double  *betas;              /* Schedule.grids                                */
    int64_t round, scan;         /* Iterators (src/pt/Iterators.jl:9-25)          */
    /* explorer adaptation state (AutoMALA) */
    double   step_size;
    double  *target_std;         /* estimated_target_std_deviations or NULL       */
    /* reduced recorders of the last round + what adapt() derived from them       */
    po_recorders reduced;
    int64_t  reduced_n_scans;
    int64_t *reduced_ip;         /* [replica][scan]                               */
    int64_t  reduced_ip_cap;
    double   global_barrier;
    double   global_barrier_var;  /* two legs: barrier of the variational leg (global_barrier is the fixed leg's) */
    double  *sched_var, *sched_fix;   /* two legs: the legs' own schedules, reference -> target */
    double  *vmean, *vstd; int v_active;   /* GaussianReference of the variational leg, once activated */
    double  *cb_x, *cb_y, *cb_m, *cb_c, *cb_d;  /* cumulative barrier interpolant */
    int      cb_valid;
    double   stepping_stone[2];
    int64_t  traces_n;
    double  *traces; int64_t traces_cap;   /* [scan][d+1] target-chain samples of the current round (recorder.jl:27,39-43) */
    double  *reduced_traces; int64_t reduced_traces_n;
    char     err[512];
    int      failed;
};

const char *po_last_error(const po_pt *pt) { return pt->err; }
int64_t po_round(const po_pt *pt) { return pt->round; }
/* resume: PT(exec_folder; round) restores shared.iterators (src/pt/checkpoint.jl:19-54); replicas / schedule / explorer
 * adaptation come in through po_set_states / po_set_schedule / po_set_explorer_adaptation */
void po_set_round(po_pt *pt, int64_t round) { pt->round = round; pt->scan = 0; }

void po_default_config(po_config *c) {
    memset(c, 0, sizeof(*c));
    c->n_chains = 10; c->dim = 2; c->seed = 1;          /* src/pt/Inputs.jl:14-20 */
    c->target = PO_TARGET_MVN; c->explorer = PO_EXPLORER_TOY;
    c->p0 = 1.0; c->p1 = 10.0;                          /* ScaledPrecisionNormalPath.jl:43-44 */
    c->slice_w = 10.0; c->slice_p = 20; c->slice_n_passes = 3; c->slice_max_iter = 1024;
    c->am_base_n_refresh = 3; c->am_exponent_n_refresh = 0.35; c->am_step_size = 1.0;
    c->am_preconditioner = 2; c->am_p0 = 1.0 / 3.0; c->am_p1 = 1.0 / 3.0;
    c->record_round_trip = 1; c->record_index_process = 1; c->record_online = 0;
    c->record_traces = 0; c->record_energy_ac1 = 0; c->explorer2 = PO_EXPLORER_NONE; c->n_chains_variational = 0; c->variational_first_tuning_round = 0;
    c->n_threads = 1;
    c->rank = 0; c->world_size = 1;
}

/* ---- the path: log potentials along the ladder ---------------------------- */
/* precision(path, beta), src/paths/ScaledPrecisionNormalPath.jl:45-46 */
static inline double mvn_precision(const po_pt *pt, double beta) {
    return (1.0 - beta) * pt->cfg.p0 + beta * pt->cfg.p1;
}
/* ScaledPrecisionNormalLogPotential(x), ScaledPrecisionNormalPath.jl:19-20 */
static inline double mvn_lp(double prec, const double *x, int64_t d) {
    return (-0.5 * prec) * po_sqr_norm(x, d);
}
/* tree_sum: the build's fixed association (balanced binary tree, natural order, zero padded) for
 * any vector of terms; po_sqr_norm is tree_sum of the squares. */
static double tree_sum(const double *t, int64_t d) {
    if (d <= 0) return 0.0;
    int64_t P = next_pow2(d);
    if (P == 1) return t[0];
    double stackbuf[1024];
    double *a = (P / 2 <= 1024) ? stackbuf : (double *)malloc(sizeof(double) * (size_t)(P / 2));
    for (int64_t i = 0; i < P / 2; i++) {
        double l = (2 * i < d) ? t[2 * i] : 0.0;
        double r = (2 * i + 1 < d) ? t[2 * i + 1] : 0.0;
        a[i] = l + r;
    }
    for (int64_t len = P / 2; len > 1; len /= 2)
        for (int64_t i = 0; i < len / 2; i++) a[i] = a[2 * i] + a[2 * i + 1];
    double s = a[0];
    if (a != stackbuf) free(a);
    return s;
}

/* Neal's funnel (reference test/supporting/dimensional-analysis.jl:33-48):
 *   z[1] ~ Normal(0, 3), z[i] ~ Normal(0, exp(z[1]/2)),  logpdf(Normal(mu,sigma), x) = -(zval^2 + log2pi)/2 - log(sigma)
 * (Distributions.jl / StatsFuns normlogpdf).  The reference accumulates the d terms sequentially;
 * the build sums them with the fixed tree (rounding-level difference).  If `g` != NULL the analytic
 * gradient is written (the reference differentiates the same expression with ForwardDiff). */
#define PO_LOG2PI 1.8378770664093453
static double funnel_lp_grad(const double *z, int64_t d, double *g, double *terms) {
    const double y = z[0];
    const double zv = y / 3.0;
    terms[0] = -(zv * zv + PO_LOG2PI) / 2.0 - log(3.0);
    const double sigma = po_exp(y / 2.0);
    const double logsigma = po_log(sigma);
    for (int64_t i = 1; i < d; i++) {
        double zi = z[i] / sigma;
        terms[i] = -(zi * zi + PO_LOG2PI) / 2.0 - logsigma;
    }
    double lp = tree_sum(terms, d);
    if (g) {
        /* d/dz_i = -z_i / sigma^2 ;  d/dy = -y/9 + sum_i ((z_i/sigma)^2 - 1) / 2 */
        for (int64_t i = 1; i < d; i++) { double zi = z[i] / sigma; g[i] = -(zi / sigma); terms[i] = (zi * zi - 1.0) / 2.0; }
        terms[0] = -(y / 9.0);
        g[0] = tree_sum(terms, d);
    }
    return lp;
}

/* The two end points of the path for chain evaluation.
 * MVN: ScaledPrecisionNormalPath is its own path (no interpolation).
 * FUNNEL: InterpolatingPath(ScaledPrecisionNormalLogPotential(p0, d), Funnel(d)) with the
 * LinearInterpolator (src/paths/InterpolatingPath.jl:25-27). */
/* does chain's path start at the variational reference?  (the variational leg of StabilizedPT, or the only leg) */
static inline int chain_uses_variational(const po_pt *pt, int64_t chain) {
    return pt->v_active && (pt->cfg.n_chains_variational > 0 ? chain < pt->cfg.n_chains_variational : 1);
}
/* gaussian_logdensity, GaussianReference.jl:43-49 (sequential sum, as there) */
static double gaussian_logdensity(const double *x, const double *mean, const double *sd, int64_t d) {
    double log_pdf = 0.0;
    for (int64_t i = 0; i < d; i++)
        log_pdf += -0.5 * log(2.0 * M_PI * (sd[i] * sd[i])) - 1.0 / (2.0 * (sd[i] * sd[i])) * ((x[i] - mean[i]) * (x[i] - mean[i]));
    return log_pdf;
}
/* the reference end of the interpolated (funnel) path at `chain` */
static inline double path_ref_lp(const po_pt *pt, int64_t chain, const double *x) {
    return chain_uses_variational(pt, chain) ? gaussian_logdensity(x, pt->vmean, pt->vstd, pt->d) : mvn_lp(pt->cfg.p0, x, pt->d);
}
static double lp_at_chain_buf(const po_pt *pt, int64_t chain, const double *x, double *scratch) {
    const double beta = pt->betas[chain];
    switch (pt->cfg.target) {
    case PO_TARGET_MVN: return mvn_lp(mvn_precision(pt, beta), x, pt->d);
    case PO_TARGET_FUNNEL: {
        /* InterpolatedLogPotential(x), src/paths/InterpolatedLogPotential.jl:9-16 */
        if (beta == 0.0) return path_ref_lp(pt, chain, x);
        if (beta == 1.0) return funnel_lp_grad(x, pt->d, NULL, scratch);
        double ref = path_ref_lp(pt, chain, x), tgt = funnel_lp_grad(x, pt->d, NULL, scratch);
        return (1.0 - beta) * ref + beta * tgt;
    }
    default: return NAN;
    }
}
/* log_potentials[chain](x) (src/tempering/NonReversiblePT.jl:72, src/schedules/discretize.jl:6-7) */
static double lp_at_chain(const po_pt *pt, int64_t chain, const double *x) {
    if (pt->cfg.target == PO_TARGET_FUNNEL) {
        double stackbuf[1024];
        double *t = pt->d <= 1024 ? stackbuf : (double *)malloc(sizeof(double) * (size_t)pt->d);
        double v = lp_at_chain_buf(pt, chain, x, t);
        if (t != stackbuf) free(t);
        return v;
    }
    return lp_at_chain_buf(pt, chain, x, NULL);
}
/* LogDensityProblems.logdensity_and_gradient on the chain's log potential:
 * MVN: src/paths/ScaledPrecisionNormalPath.jl:30-34; interpolated: InterpolatedAD,
 * src/explorers/BufferedAD.jl:98-112 (no beta == 0 / 1 short-circuit there). */
static double lp_grad_at_chain(const po_pt *pt, int64_t chain, const double *x, double *grad, double *scratch) {
    const int64_t d = pt->d;
    const double beta = pt->betas[chain];
    if (pt->cfg.target == PO_TARGET_MVN) {
        double prec = mvn_precision(pt, beta);
        double logdens = mvn_lp(prec, x, d);
        for (int64_t i = 0; i < d; i++) grad[i] = (-prec) * x[i];
        return logdens;
    }
    double logdens = 0.0;
    double l = path_ref_lp(pt, chain, x);
    logdens += l * (1.0 - beta);
    if (chain_uses_variational(pt, chain))      /* BufferedAD{GaussianReference}, GaussianReference.jl:66-73 */
        for (int64_t i = 0; i < d; i++) grad[i] = (-1.0 / (pt->vstd[i] * pt->vstd[i]) * (x[i] - pt->vmean[i])) * (1.0 - beta);
    else
        for (int64_t i = 0; i < d; i++) grad[i] = ((-pt->cfg.p0) * x[i]) * (1.0 - beta);
    double *g2 = scratch, *terms = scratch + d;
    l = funnel_lp_grad(x, d, g2, terms);
    logdens += l * beta;
    for (int64_t i = 0; i < d; i++) grad[i] = grad[i] + g2[i] * beta;
    return logdens;
}
/* LogDensityProblems.logdensity(::InterpolatedAD) BufferedAD.jl:89-94 == (1-beta)*l1 + beta*l2, no short-circuit */
static double lp_ad_at_chain(const po_pt *pt, int64_t chain, const double *x, double *scratch) {
    if (pt->cfg.target == PO_TARGET_MVN) return mvn_lp(mvn_precision(pt, pt->betas[chain]), x, pt->d);
    const double beta = pt->betas[chain];
    double l1 = path_ref_lp(pt, chain, x), l2 = funnel_lp_grad(x, pt->d, NULL, scratch);
    return (1.0 - beta) * l1 + beta * l2;
}

/* ---- DEO swap graph -------------------------------------------------------- */
/* partner_chain(::OddEven, chain), src/swap/OddEven.jl:23-31, 1-based there.
 * `even` = iseven(scan) (src/swap/DEO.jl:12). */
static int64_t partner_chain(int64_t N, int even, int64_t chain0) {
    int64_t chain = chain0 + 1;
    int chain_even = (chain % 2 == 0);
    int64_t proposed = chain + ((chain_even == even) ? 1 : -1);
    if (proposed == 0) return 0;
    if (proposed == N + 1) return N - 1;
    return proposed - 1;
}
/* One leg: deo (DEO.jl:13-14).  Two legs (n_var > 0): variational_deo -- references at both ends, targets in the
 * middle.  NB the reference has TWO target predicates: explore! asks the VariationalDEO (n_chains_var based,
 * VariationalDEO.jl:20-21), swap! / round trips ask the VariationalOddEven (n_chains_fixed based, OddEven.jl:47-48). */
#define NVAR(pt) ((pt)->cfg.n_chains_variational)
#define NFIX(pt) ((pt)->cfg.n_chains)
static inline int is_reference_pt(const po_pt *pt, int64_t c) {
    if (NVAR(pt) > 0) return c == 0 || c == pt->N - 1;
    return c == 0 && pt->N > 1;
}
static inline int is_target_explore(const po_pt *pt, int64_t c) {
    if (NVAR(pt) > 0) return c == NVAR(pt) - 1 || c == NVAR(pt);
    return c == pt->N - 1;
}
static inline int is_target_swap(const po_pt *pt, int64_t c) {
    if (NVAR(pt) > 0) return c == NFIX(pt) - 1 || c == NFIX(pt);
    return c == pt->N - 1;
}


/* ---- 2-D Ising model (reference examples/ising.jl) ------------------------------------------------
 * state: L x L spins in {false,true} (stored as 0.0 / 1.0 in the replica's state vector, row-major:
 * matrix[i,j] <-> state[(i-1)*L + (j-1)]), plus the cached sum_pair_products (r->aux).
 * log potential at a chain: InterpolatedLogPotential between IsingLogPotential(0.0, L) (default_reference,
 * ising.jl:77) and IsingLogPotential(beta, L) with the linear interpolator. */
static inline int ising_L(const po_pt *pt) { int L = (int)llround(sqrt((double)pt->d)); return L; }
static inline int ising_sign(double b) { return b != 0.0 ? +1 : -1; }
static inline int ising_wrap(int i, int L) { return i < 0 ? L - 1 : (i >= L ? 0 : i); }        /* wrap(), ising.jl:61-69, 0-based */
static inline int ising_sum_neighbours(const double *m, int L, int i, int j) {                /* ising.jl:57-60 */
    return ising_sign(m[ising_wrap(i - 1, L) * L + j]) + ising_sign(m[ising_wrap(i + 1, L) * L + j]) +
           ising_sign(m[i * L + ising_wrap(j - 1, L)]) + ising_sign(m[i * L + ising_wrap(j + 1, L)]);
}
static int64_t ising_recompute(const double *m, int L) {                                       /* ising.jl:27-35 */
    int64_t sum = 0;
    for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) sum += ising_sign(m[i * L + j]) * ising_sum_neighbours(m, L, i, j);
    return sum / 2;
}
static void ising_flip(po_replica *r, int L, int i, int j) {                                    /* flip!, ising.jl:38-46 */
    double *m = r->state;
    int before = ising_sign(m[i * L + j]) * ising_sum_neighbours(m, L, i, j);
    m[i * L + j] = m[i * L + j] != 0.0 ? 0.0 : 1.0;
    int after = ising_sign(m[i * L + j]) * ising_sum_neighbours(m, L, i, j);
    r->aux += (int64_t)(after - before);
}
/* InterpolatedLogPotential(state): ref = 0.0 * spp, target = beta_ising * spp (ising.jl:74) */
static double ising_lp(const po_pt *pt, int64_t chain, int64_t spp) {
    const double beta = pt->betas[chain];
    const double ref = 0.0 * (double)spp, tgt = pt->cfg.p0 * (double)spp;
    if (beta == 0.0) return ref;
    if (beta == 1.0) return tgt;
    return (1.0 - beta) * ref + beta * tgt;
}
/* rand(rng, Bool): bit k of one UInt64 draw, k from the policy (include/pte_rng_policy.h; default 0 = `% Bool`) -- UNPINNED */
static inline int po_rand_bool(po_rng *r) { return (int)((po_rng_next_u64(r) >> PTE_RNG_POLICY_BOOL_BIT(g_rng_policy)) & 1ULL); }
int po_rand_bool_pub(po_rng *r) { return po_rand_bool(r); }
static void ising_sample_iid(po_pt *pt, po_replica *r) {                                       /* iid_bernoulli!, ising.jl:49-58 */
    const int L = ising_L(pt);
    for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) r->state[i * L + j] = po_rand_bool(&r->rng) ? 1.0 : 0.0;
    r->aux = ising_recompute(r->state, L);
}
static void ising_step(po_pt *pt, po_replica *r) {                                             /* step!(::IsingMetropolis), ising.jl:96-116 */
    const int L = ising_L(pt);
    for (int k = 0; k < pt->cfg.slice_n_passes; k++)
        for (int i = 0; i < L; i++)
            for (int j = 0; j < L; j++) {
                double before = ising_lp(pt, r->chain, r->aux);
                ising_flip(r, L, i, j);
                double after = ising_lp(pt, r->chain, r->aux);
                double accept_ratio = exp(after - before);
                if (accept_ratio < 1 && po_rand(&r->rng) > accept_ratio) ising_flip(r, L, i, j);
            }
}

/* ========================================================================== */
/* explorers                                                                  */
/* ========================================================================== */
static void fail(po_pt *pt, const char *msg) {
#pragma omp critical
    { if (!pt->failed) { pt->failed = 1; snprintf(pt->err, sizeof(pt->err), "%s", msg); } }
}

/* rand!(rng, x, lp) / sample_iid! / ToyExplorer.step!
 * (src/targets/toy_mvn_target.jl:15-21, src/explorers/ToyExplorer.jl:7-12) */
static void mvn_sample_iid(po_pt *pt, po_replica *r) {
    /* funnel path: sample_iid!(::InterpolatedLogPotential) -> the reference end point (src/targets/target.jl:93-98) */
    if (pt->cfg.target == PO_TARGET_FUNNEL && chain_uses_variational(pt, r->chain)) {   /* sample_iid!(::GaussianReference), :33-40 */
        for (int64_t i = 0; i < pt->d; i++) r->state[i] = po_randn(&r->rng) * pt->vstd[i] + pt->vmean[i];
        return;
    }
    double prec = pt->cfg.target == PO_TARGET_FUNNEL ? pt->cfg.p0 : mvn_precision(pt, pt->betas[r->chain]);
    double sd = sqrt(prec);
    for (int64_t i = 0; i < pt->d; i++) r->state[i] = po_randn(&r->rng) / sd;
}

/* ---- SliceSampler (src/explorers/SliceSampler.jl) -------------------------- */
typedef struct { po_pt *pt; po_replica *r; int64_t chain; int64_t c; } slice_ctx;

static inline double slice_lp(slice_ctx *s) { return lp_at_chain(s->pt, s->chain, s->r->state); }

static inline int jl_isapprox(double x, double y) {  /* Base.isapprox defaults, rtol = sqrt(eps) */
    if (x == y) return 1;
    if (!isfinite(x) || !isfinite(y)) return 0;
    double ax = fabs(x), ay = fabs(y);
    return fabs(x - y) <= 1.4901161193847656e-8 * (ax > ay ? ax : ay);
}

/* slice_accept, SliceSampler.jl:192-237 */
static int slice_accept(slice_ctx *s, double new_position, double z, double L, double R,
                        double lp_L, double lp_R) {
    const po_config *h = &s->pt->cfg;
    double *ptr = &s->r->state[s->c];
    double old_position = *ptr;
    double Lhat = L, Rhat = R;
    int Rstale = 0, Lstale = 0, D = 0;
    while (Rhat - Lhat > 1.1 * h->slice_w) {
        double M = (Lhat + Rhat) / 2.0;
        if ((old_position < M && new_position >= M) || (old_position >= M && new_position < M)) D = 1;
        if (new_position < M) { Rhat = M; Rstale = 1; }
        else { Lhat = M; Lstale = 1; }
        if (D) {
            if (Lstale) { *ptr = Lhat; lp_L = slice_lp(s); Lstale = 0; }
            if (Rstale) { *ptr = Rhat; lp_R = slice_lp(s); Rstale = 0; }
            if (z >= lp_L && z >= lp_R) {
                *ptr = old_position;
                mean_fit(&s->r->rec.expl_acc[s->chain], 0.0);
                return 0;
            }
        }
    }
    *ptr = old_position;
    mean_fit(&s->r->rec.expl_acc[s->chain], 1.0);
    return 1;
}

/* slice_sample_coord! (generic Float64 case), SliceSampler.jl:89-95 with
 * slice_double :97-126, initialize_slice_endpoints :129-133, slice_shrink! :144-186 */
static int slice_sample_coord(slice_ctx *s, double *cached_lp) {
    const po_config *h = &s->pt->cfg;
    po_rng *rng = &s->r->rng;
    double *ptr = &s->r->state[s->c];
    double z = *cached_lp - po_randexp(rng);
    /* slice_double */
    double old_position = *ptr;
    double L = old_position - h->slice_w * po_rand(rng);
    double R = L + h->slice_w;
    int K = h->slice_p;
    *ptr = L; double potent_L = slice_lp(s);
    *ptr = R; double potent_R = slice_lp(s);
    while (K > 0 && (z < potent_L || z < potent_R)) {
        double V = po_rand(rng);
        if (V <= 0.5) { L = L - (R - L); *ptr = L; potent_L = slice_lp(s); }
        else { R = R + (R - L); *ptr = R; potent_R = slice_lp(s); }
        K -= 1;
    }
    sum_fit(&s->r->rec.expl_steps[s->chain], (double)(h->slice_p - K));
    *ptr = old_position;
    /* slice_shrink! */
    double Lbar = L, Rbar = R, new_lp = 0.0;
    for (int n = 1; n <= h->slice_max_iter; n++) {
        double new_position = Lbar + po_rand(rng) * (Rbar - Lbar);   /* draw_new_position :188 */
        *ptr = new_position;
        new_lp = slice_lp(s);
        int consider = z < new_lp;
        *ptr = old_position;
        if (consider && slice_accept(s, new_position, z, L, R, potent_L, potent_R)) {
            *ptr = new_position;
            sum_fit(&s->r->rec.expl_steps[s->chain], (double)n);
            *cached_lp = new_lp;
            return 0;
        }
        if (new_position < *ptr) Lbar = new_position; else Rbar = new_position;
        if (jl_isapprox(Lbar, Rbar)) {
            *ptr = old_position;
            sum_fit(&s->r->rec.expl_steps[s->chain], (double)n);
            *cached_lp = slice_lp(s);
            return 0;
        }
    }
    fail(s->pt, "SliceSampler: maximum number of iterations reached");
    return 1;
}

/* step!(::SliceSampler) :24-30, slice_sample! :43-62, cached_log_potential :32-41 */
static int slice_step(po_pt *pt, po_replica *r) {
    slice_ctx s = { pt, r, r->chain, 0 };
    double cached_lp = -INFINITY;
    for (int pass = 0; pass < pt->cfg.slice_n_passes; pass++) {
        if (cached_lp == -INFINITY) {
            cached_lp = slice_lp(&s);
            if (cached_lp == -INFINITY) { fail(pt, "SliceSampler: initialized outside the support"); return 1; }
        }
        for (int64_t c = 0; c < pt->d; c++) {
            s.c = c;
            if (slice_sample_coord(&s, &cached_lp)) return 1;
            if (!isfinite(cached_lp)) { fail(pt, "SliceSampler: invalid log density after update"); return 1; }
        }
    }
    return 0;
}


/* ---- SliceSampler on Bool / Integer / mixed states (SliceSampler.jl:43-95, 128-142, 188-189) --------------------------------
 * SURVEY.md 8 row a8's last sub-row.  The reference dispatches slice_sample_coord! on typeof(pointer[]) PER COORDINATE (:47-48), so a
 * state may mix Float64, Integer and Bool coordinates (its DynamicPPL models do).  The device has no target with such coordinates (closed
 * enum of log-potential families; pte_create refuses), so this part of the restatement stands alone behind a log-potential call-back
 * instead of a po_pt: state coordinates are held as doubles (integers exact below 2^53, Bool = 0.0 / 1.0), `kind` says which method a
 * coordinate takes.  The Float64 method here is the same arithmetic as slice_sample_coord above (tests/test_oracle_slice_mixed.py holds
 * the two bit for bit against each other).
 *
 * rand(rng, a:b) on Int64 ranges: Julia's Random stdlib (>= 1.5; Project.toml compat julia = "1.8") builds SamplerRangeNDL for every
 * AbstractRNG and BitInteger64 element type -- Lemire's "nearly division-less" sampler over rand(rng, UInt64): s = b - a + 1 (mod 2^64),
 * m = x * s as UInt128, low = m mod 2^64; if low < s { t = (2^64 - s) mod s; redraw while low < t }; result = a + (s == 0 ? x : m >> 64).
 * The stdlib is not under /root/reference: restated from the published algorithm (Random/src/generation.jl, "SamplerRangeNDL") -- UNPINNED
 * like the other stdlib samplers (pt_oracle.h). */
int64_t po_rand_range(po_rng *r, int64_t a, int64_t b) {
    const uint64_t s = (uint64_t)b - (uint64_t)a + 1ULL;                    /* overflow ok: 0 = the full range */
    uint64_t x = po_rng_next_u64(r);
    unsigned __int128 m = (unsigned __int128)x * s;
    uint64_t low = (uint64_t)m;
    if (low < s) {
        const uint64_t t = (0ULL - s) % s;                                  /* mod(-s, s); s != 0 here because low < s */
        while (low < t) {
            x = po_rng_next_u64(r);
            m = (unsigned __int128)x * s;
            low = (uint64_t)m;
        }
    }
    return (int64_t)((s == 0 ? x : (uint64_t)(m >> 64)) + (uint64_t)a);
}

typedef struct {
    po_rng *rng; double *state; int64_t d, c;
    const po_slice_params *h;
    po_logpotential_fn lp; void *lp_ctx;
    po_slice_stats *st;
} mixed_ctx;
static inline double mixed_lp(mixed_ctx *s) { return s->lp(s->state, s->d, s->lp_ctx); }
static inline void mixed_rec_steps(mixed_ctx *s, double v) { if (s->st) { s->st->steps_sum += v; s->st->steps_n += 1; } }
static inline void mixed_rec_acc(mixed_ctx *s, double v) {                /* Mean(): mu += (1/n)(x - mu), as mean_fit */
    if (s->st) { s->st->acc_n += 1; s->st->acc_mean += (1.0 / (double)s->st->acc_n) * (v - s->st->acc_mean); }
}

/* slice_accept :192-237 -- one body for Float64 and Integer coordinates: with an Integer pointer Lhat / Rhat start as Int64 and turn
 * into Float64 at the first M = (Lhat + Rhat) / 2.0; R - L = w 2^k with integer w keeps every M integral, so the stores into the
 * Integer pointer convert exactly, and every comparison (Int against Float64) is exact below 2^53: doubles carry it. */
static int mixed_slice_accept(mixed_ctx *s, double new_position, double z, double L, double R, double lp_L, double lp_R) {
    double *ptr = &s->state[s->c];
    const double old_position = *ptr;
    double Lhat = L, Rhat = R;
    int Rstale = 0, Lstale = 0, D = 0;
    while (Rhat - Lhat > 1.1 * s->h->w) {
        const double M = (Lhat + Rhat) / 2.0;
        if ((old_position < M && new_position >= M) || (old_position >= M && new_position < M)) D = 1;
        if (new_position < M) { Rhat = M; Rstale = 1; }
        else { Lhat = M; Lstale = 1; }
        if (D) {
            if (Lstale) { *ptr = Lhat; lp_L = mixed_lp(s); Lstale = 0; }
            if (Rstale) { *ptr = Rhat; lp_R = mixed_lp(s); Rstale = 0; }
            if (z >= lp_L && z >= lp_R) { *ptr = old_position; mixed_rec_acc(s, 0.0); return 0; }
        }
    }
    *ptr = old_position;
    mixed_rec_acc(s, 1.0);
    return 1;
}

/* slice_sample_coord!(..., ::Type{Bool}) :65-86: the full conditional, ONE density evaluation, ONE rand(rng); records nothing */
static void mixed_coord_bool(mixed_ctx *s, double *cached_lp) {
    double *ptr = &s->state[s->c];
    double lp0, lp1;
    if (*ptr != 0.0) { lp1 = *cached_lp; *ptr = 0.0; lp0 = mixed_lp(s); }
    else             { lp0 = *cached_lp; *ptr = 1.0; lp1 = mixed_lp(s); }
    const double prob_ratio = exp(lp1 - lp0);
    const double prob_zero = 1.0 / (1.0 + prob_ratio);                      /* inv(1 + prob_ratio) */
    if (po_rand(s->rng) < prob_zero) { *ptr = 0.0; *cached_lp = lp0; }
    else                             { *ptr = 1.0; *cached_lp = lp1; }
}

/* slice_sample_coord!(..., ::Type) :89-95 on an Integer coordinate: slice_double :97-126 with initialize_slice_endpoints(::Integer)
 * :136-142 (L = current - rand(rng, 0:width), width = ceil(T, w), w must be integral), slice_shrink! :144-186 with
 * draw_new_position(::Integer, ::Integer) = rand(rng, L:R) :189 and Lbar ≈ Rbar, which is == on Integers (Base.isapprox, rtol = 0). */
static int mixed_coord_integer(mixed_ctx *s, double *cached_lp, char *err, size_t errlen) {
    const po_slice_params *h = s->h;
    double *ptr = &s->state[s->c];
    if (h->w != floor(h->w) || !isfinite(h->w)) {
        snprintf(err, errlen, "for integer variables, the width should be an integer. Got: %g", h->w);   /* @assert :137 */
        return 1;
    }
    const double z = *cached_lp - po_randexp(s->rng);
    const int64_t old_position = (int64_t)*ptr;
    const int64_t width = (int64_t)ceil(h->w);
    int64_t L = old_position - po_rand_range(s->rng, 0, width);
    int64_t R = L + width;
    int K = h->p;
    *ptr = (double)L; double potent_L = mixed_lp(s);
    *ptr = (double)R; double potent_R = mixed_lp(s);
    while (K > 0 && (z < potent_L || z < potent_R)) {
        const double V = po_rand(s->rng);
        if (V <= 0.5) { L = L - (R - L); *ptr = (double)L; potent_L = mixed_lp(s); }
        else          { R = R + (R - L); *ptr = (double)R; potent_R = mixed_lp(s); }
        K -= 1;
    }
    mixed_rec_steps(s, (double)(h->p - K));
    *ptr = (double)old_position;
    int64_t Lbar = L, Rbar = R;
    for (int n = 1; n <= h->max_iter; n++) {
        const int64_t new_position = po_rand_range(s->rng, Lbar, Rbar);
        *ptr = (double)new_position;
        const double new_lp = mixed_lp(s);
        const int consider = z < new_lp;
        *ptr = (double)old_position;
        if (consider && mixed_slice_accept(s, (double)new_position, z, (double)L, (double)R, potent_L, potent_R)) {
            *ptr = (double)new_position;
            mixed_rec_steps(s, (double)n);
            *cached_lp = new_lp;
            return 0;
        }
        if (new_position < old_position) Lbar = new_position; else Rbar = new_position;
        if (Lbar == Rbar) {
            *ptr = (double)old_position;
            mixed_rec_steps(s, (double)n);
            *cached_lp = mixed_lp(s);
            return 0;
        }
    }
    snprintf(err, errlen, "SliceSampler: maximum number of iterations reached");
    return 1;
}

/* the Float64 method, as slice_sample_coord above (same statements, the log potential through the call-back) */
static int mixed_coord_float(mixed_ctx *s, double *cached_lp, char *err, size_t errlen) {
    const po_slice_params *h = s->h;
    double *ptr = &s->state[s->c];
    const double z = *cached_lp - po_randexp(s->rng);
    const double old_position = *ptr;
    double L = old_position - h->w * po_rand(s->rng);
    double R = L + h->w;
    int K = h->p;
    *ptr = L; double potent_L = mixed_lp(s);
    *ptr = R; double potent_R = mixed_lp(s);
    while (K > 0 && (z < potent_L || z < potent_R)) {
        const double V = po_rand(s->rng);
        if (V <= 0.5) { L = L - (R - L); *ptr = L; potent_L = mixed_lp(s); }
        else          { R = R + (R - L); *ptr = R; potent_R = mixed_lp(s); }
        K -= 1;
    }
    mixed_rec_steps(s, (double)(h->p - K));
    *ptr = old_position;
    double Lbar = L, Rbar = R;
    for (int n = 1; n <= h->max_iter; n++) {
        const double new_position = Lbar + po_rand(s->rng) * (Rbar - Lbar);
        *ptr = new_position;
        const double new_lp = mixed_lp(s);
        const int consider = z < new_lp;
        *ptr = old_position;
        if (consider && mixed_slice_accept(s, new_position, z, L, R, potent_L, potent_R)) {
            *ptr = new_position;
            mixed_rec_steps(s, (double)n);
            *cached_lp = new_lp;
            return 0;
        }
        if (new_position < *ptr) Lbar = new_position; else Rbar = new_position;
        if (jl_isapprox(Lbar, Rbar)) {
            *ptr = old_position;
            mixed_rec_steps(s, (double)n);
            *cached_lp = mixed_lp(s);
            return 0;
        }
    }
    snprintf(err, errlen, "SliceSampler: maximum number of iterations reached");
    return 1;
}

/* step!(::SliceSampler) :24-30 over slice_sample!(h, state::AbstractVector, ...) :43-62, dispatching per coordinate on kind[c] */
int po_slice_step_mixed(po_rng *rng, double *state, const int32_t *kind, int64_t d, const po_slice_params *h,
                        po_logpotential_fn lp, void *lp_ctx, po_slice_stats *stats, char *err, size_t errlen) {
    mixed_ctx s = { rng, state, d, 0, h, lp, lp_ctx, stats };
    char dummy[8];
    if (!err) { err = dummy; errlen = sizeof(dummy); }
    err[0] = 0;
    double cached_lp = -INFINITY;
    for (int pass = 0; pass < h->n_passes; pass++) {
        if (cached_lp == -INFINITY) {                                       /* cached_log_potential :32-41 */
            cached_lp = mixed_lp(&s);
            if (cached_lp == -INFINITY) { snprintf(err, errlen, "SliceSampler: initialized outside the support"); return 1; }
        }
        for (int64_t c = 0; c < d; c++) {
            s.c = c;
            switch (kind ? kind[c] : PO_COORD_FLOAT64) {
            case PO_COORD_BOOL:    mixed_coord_bool(&s, &cached_lp); break;
            case PO_COORD_INTEGER: if (mixed_coord_integer(&s, &cached_lp, err, errlen)) return 1; break;
            case PO_COORD_FLOAT64: if (mixed_coord_float(&s, &cached_lp, err, errlen)) return 1; break;
            default: snprintf(err, errlen, "po_slice_step_mixed: coordinate %lld has kind %d", (long long)c, (int)kind[c]); return 1;
            }
            if (!isfinite(cached_lp)) { snprintf(err, errlen, "SliceSampler: invalid log density after update"); return 1; }
        }
    }
    return 0;
}

/* ---- AutoMALA (src/explorers/AutoMALA.jl, src/explorers/hamiltonian_dynamics.jl) -------------- */
typedef struct {
    po_pt *pt; po_replica *r; int64_t chain;
    double *momentum, *precond, *start, *state_before, *momentum_before, *grad, *scratch;
} am_ctx;

static inline double am_log_joint(am_ctx *a) {          /* log_joint(target, state, momentum) :48-49 */
    return lp_ad_at_chain(a->pt, a->chain, a->r->state, a->scratch) - 0.5 * po_sqr_norm(a->momentum, a->pt->d);
}
/* hamiltonian_dynamics! with n_steps = 1 (:59-102) */
static int am_leap_frog(am_ctx *a, double step_size) {
    const int64_t d = a->pt->d;
    double *x = a->r->state, *p = a->momentum, *M = a->precond, *g = a->grad;
    double logp = lp_grad_at_chain(a->pt, a->chain, x, g, a->scratch);
    for (int64_t i = 0; i < d; i++) g[i] = g[i] / M[i];
    (void)logp;
    const double half = step_size / 2;
    for (int64_t i = 0; i < d; i++) p[i] = p[i] + half * g[i];
    for (int64_t i = 0; i < d; i++) x[i] = x[i] + step_size * (p[i] / M[i]);
    logp = lp_grad_at_chain(a->pt, a->chain, x, g, a->scratch);
    for (int64_t i = 0; i < d; i++) g[i] = g[i] / M[i];
    double current = logp - 0.5 * po_sqr_norm(p, d);
    if (!isfinite(current)) return 0;
    for (int64_t i = 0; i < d; i++) p[i] = p[i] + half * g[i];
    if (!isfinite(po_sqr_norm(p, d))) return 0;
    return 1;
}
/* auto_step_size (:184-214) with log_joint_difference_function (:250-275), grow/shrink (:216-248) */
static int am_auto_step_size(am_ctx *a, double step_size, double lower, double upper, int *exponent_out) {
    const int64_t d = a->pt->d;
    memcpy(a->state_before, a->r->state, sizeof(double) * (size_t)d);
    memcpy(a->momentum_before, a->momentum, sizeof(double) * (size_t)d);
    const double h_before = am_log_joint(a);
#define AM_DIFF(eps, out) do { am_leap_frog(a, (eps)); double h_after_ = am_log_joint(a);                 \
        memcpy(a->r->state, a->state_before, sizeof(double) * (size_t)d);                                 \
        memcpy(a->momentum, a->momentum_before, sizeof(double) * (size_t)d); (out) = h_after_ - h_before; } while (0)
    double diff;
    AM_DIFF(step_size, diff);
    int n_steps = 0, exponent = 0;
    if (!isfinite(diff) || diff < lower) {
        int n = 1;
        for (;;) {
            step_size /= 2.0;
            AM_DIFF(step_size, diff);
            if (step_size == 0.0) { fail(a->pt, "AutoMALA: could not find a positive step size"); return 1; }
            if (diff > lower) { n_steps = n; exponent = -n; break; }
            n++;
        }
    } else if (diff > upper) {
        int n = 1;
        for (;;) {
            step_size *= 2.0;
            AM_DIFF(step_size, diff);
            if (!isfinite(diff) || diff < upper) { n_steps = n; exponent = n - 1; break; }
            n++;
        }
    }
#undef AM_DIFF
    sum_fit(&a->r->rec.expl_steps[a->chain], (double)(1 + n_steps));
    mean_fit(&a->r->rec.am_factors[a->chain], ldexp(1.0, exponent));
    *exponent_out = exponent;
    return 0;
}
/* build_preconditioner! (src/explorers/Preconditioner.jl:57-77) */
static void am_build_preconditioner(po_pt *pt, po_replica *r, double *precond) {
    const int64_t d = pt->d;
    const po_config *cfg = &pt->cfg;
    if (pt->target_std == NULL || cfg->am_preconditioner == 0) {
        for (int64_t i = 0; i < d; i++) precond[i] = 1.0;
    } else if (cfg->am_preconditioner == 1) {
        for (int64_t i = 0; i < d; i++) precond[i] = pt->target_std[i] == 0.0 ? 1.0 : 1.0 / pt->target_std[i];
    } else {
        double u = po_rand(&r->rng);
        if (u <= cfg->am_p0) {
            for (int64_t i = 0; i < d; i++) precond[i] = pt->target_std[i] == 0.0 ? 1.0 : 1.0 / pt->target_std[i];
        } else if (u <= cfg->am_p0 + cfg->am_p1) {
            for (int64_t i = 0; i < d; i++) precond[i] = 1.0;
        } else {
            double mix = po_rand(&r->rng), rmix = 1.0 - mix;
            for (int64_t i = 0; i < d; i++) precond[i] = pt->target_std[i] == 0.0 ? 1.0 : mix + rmix / pt->target_std[i];
        }
    }
}
/* mala! (src/explorers/MALA.jl:74-97): fixed step size, one leapfrog, always Metropolis-Hastings */
static int mala_step(po_pt *pt, po_replica *r) {
    const int64_t d = pt->d;
    const po_config *cfg = &pt->cfg;
    am_ctx a = { pt, r, r->chain, r->buf, r->buf + d, r->buf + 2 * d, r->buf + 3 * d, r->buf + 4 * d, r->buf + 5 * d, r->buf + 6 * d };
    am_build_preconditioner(pt, r, a.precond);
    const int n_refresh = cfg->am_base_n_refresh * (int)ceil(pow((double)d, cfg->am_exponent_n_refresh));
    for (int it = 0; it < n_refresh; it++) {
        memcpy(a.start, r->state, sizeof(double) * (size_t)d);
        for (int64_t i = 0; i < d; i++) a.momentum[i] = po_randn(&r->rng);
        const double init_joint_log = am_log_joint(&a);
        if (!isfinite(init_joint_log)) { fail(pt, "MALA can only be called on a configuration of positive density."); return 1; }
        am_leap_frog(&a, cfg->am_step_size);
        for (int64_t i = 0; i < d; i++) a.momentum[i] = a.momentum[i] * -1.0;
        const double e = po_exp(am_log_joint(&a) - init_joint_log);
        double probability = e < 1.0 ? e : 1.0;
        if (isnan(e)) probability = e;
        mean_fit(&r->rec.expl_acc[a.chain], probability);
        if (po_rand(&r->rng) < probability) { /* accept */ }
        else memcpy(r->state, a.start, sizeof(double) * (size_t)d);
        sum_fit(&r->rec.expl_steps[a.chain], 1.0);
    }
    return 0;
}
/* step! -> _extract_commons_and_run! (:84-104) -> auto_mala! (:106-182) */
static int automala_step(po_pt *pt, po_replica *r) {
    const int64_t d = pt->d;
    const po_config *cfg = &pt->cfg;
    am_ctx a = { pt, r, r->chain, r->buf, r->buf + d, r->buf + 2 * d, r->buf + 3 * d, r->buf + 4 * d, r->buf + 5 * d, r->buf + 6 * d };
    const int use_mh = (pt->scan != 1);
    am_build_preconditioner(pt, r, a.precond);
    const int n_refresh = cfg->am_base_n_refresh * (int)ceil(pow((double)d, cfg->am_exponent_n_refresh));
    for (int it = 0; it < n_refresh; it++) {
        memcpy(a.start, r->state, sizeof(double) * (size_t)d);
        for (int64_t i = 0; i < d; i++) a.momentum[i] = po_randn(&r->rng);
        const double init_joint_log = am_log_joint(&a);
        if (!isfinite(init_joint_log)) { fail(pt, "AutoMALA can only be called on a configuration of positive density."); return 1; }
        double ua = po_rand(&r->rng), ub = po_rand(&r->rng);
        double lower = po_log(ua < ub ? ua : ub), upper = po_log(ua < ub ? ub : ua);
        int proposed_exponent, reversed_exponent;
        if (am_auto_step_size(&a, pt->step_size, lower, upper, &proposed_exponent)) return 1;
        const double proposed_step_size = pt->step_size * ldexp(1.0, proposed_exponent);
        am_leap_frog(&a, proposed_step_size);
        if (use_mh) {
            for (int64_t i = 0; i < d; i++) a.momentum[i] = a.momentum[i] * -1.0;
            if (am_auto_step_size(&a, pt->step_size, lower, upper, &reversed_exponent)) return 1;
            int passed = (reversed_exponent == proposed_exponent);
            mean_fit(&r->rec.rev_rate[a.chain], passed ? 1.0 : 0.0);
            double probability = 0.0;
            if (passed) {
                double final_joint_log = am_log_joint(&a);
                double e = po_exp(final_joint_log - init_joint_log);
                probability = e < 1.0 ? e : 1.0;      /* min(1.0, NaN) is NaN in Julia; rand < NaN is false either way */
                if (isnan(e)) probability = e;
            }
            mean_fit(&r->rec.expl_acc[a.chain], probability);
            if (po_rand(&r->rng) < probability) { /* accept */ }
            else memcpy(r->state, a.start, sizeof(double) * (size_t)d);
        }
    }
    return 0;
}

/* explore!(pt, replica, explorer), src/pt/pigeons.jl:101-132 */
static inline double lp_of_replica(const po_pt *pt, const po_replica *r) {   /* find_log_potential(replica, ...)(replica.state) */
    return pt->cfg.target == PO_TARGET_ISING ? ising_lp(pt, r->chain, r->aux) : lp_at_chain(pt, r->chain, r->state);
}
static inline int uses_gradient_sampler(const po_config *c) {
    return c->explorer == PO_EXPLORER_AUTOMALA || c->explorer == PO_EXPLORER_MALA || c->explorer2 == PO_EXPLORER_AUTOMALA || c->explorer2 == PO_EXPLORER_MALA;
}
static inline int64_t traces_row_width(const po_pt *pt) {   /* extended: all chains; two legs: the two target chains */
    return (pt->cfg.record_traces == 2 ? pt->N : (pt->cfg.n_chains_variational > 0 ? 2 : 1)) * (pt->d + 1);
}
/* called from the serial part of a scan, before the (threaded) explore loop */
static void traces_reserve(po_pt *pt) {
    if (!pt->cfg.record_traces) return;
    const int64_t t = pt->scan - 1, row = traces_row_width(pt);
    if (t >= pt->traces_cap) {
        const int64_t ncap = 2 * (t + 1);
        pt->traces = (double *)realloc(pt->traces, sizeof(double) * (size_t)(ncap * row));
        memset(pt->traces + pt->traces_cap * row, 0, sizeof(double) * (size_t)((ncap - pt->traces_cap) * row));
        pt->traces_cap = ncap;
    }
    if (t + 1 > pt->traces_n) pt->traces_n = t + 1;
}
static int explore_replica_inner(po_pt *pt, po_replica *r);
static int explore_replica(po_pt *pt, po_replica *r) {
    if (pt->cfg.target == PO_TARGET_TEST_SWAPPER) return 0;   /* state nothing, step! no-op (pair_swapper.jl:137-143) */
    const int ac = pt->cfg.record_energy_ac1;
    const double before = ac ? lp_of_replica(pt, r) : 0.0;    /* eval_if_ac_requested, pigeons.jl:134-137 */
    if (explore_replica_inner(pt, r)) return 1;
    if (ac) cov2_fit(&r->rec.eac[r->chain], before, lp_of_replica(pt, r));   /* process_ac!, :139-143 */
    if (pt->cfg.record_traces && pt->cfg.target != PO_TARGET_ISING &&
        (pt->cfg.record_traces == 2 || is_target_explore(pt, r->chain))) {   /* pigeons.jl:116-125; == 2: inputs.extended_traces */
        const int64_t w = pt->d + 1, t = pt->scan - 1;
        const int64_t row = traces_row_width(pt);                /* capacity ensured by traces_reserve (serial) */
        double *dst = pt->traces + t * row + (pt->cfg.record_traces == 2 ? r->chain * w : (r->chain == NVAR(pt) && NVAR(pt) > 0 ? w : 0));
        memcpy(dst, r->state, sizeof(double) * (size_t)pt->d);
        dst[pt->d] = lp_of_replica(pt, r);
    }
    return 0;
}
static int explore_replica_inner(po_pt *pt, po_replica *r) {
    const int64_t N = pt->N;
    if (pt->cfg.target == PO_TARGET_ISING) {
        if (is_reference_pt(pt, r->chain)) ising_sample_iid(pt, r);
        else ising_step(pt, r);
        return 0;
    }
    if (is_reference_pt(pt, r->chain)) {
        mvn_sample_iid(pt, r);
    } else {
        const int32_t kinds[2] = { pt->cfg.explorer, pt->cfg.explorer2 };      /* step!(::Compose), Compose.jl:16-19 */
        for (int k = 0; k < 2; k++) {
            if (k == 1 && kinds[1] == PO_EXPLORER_NONE) break;
            switch (kinds[k]) {
            case PO_EXPLORER_TOY:   mvn_sample_iid(pt, r); break;
            case PO_EXPLORER_SLICE: if (slice_step(pt, r)) return 1; break;
            case PO_EXPLORER_AUTOMALA: if (automala_step(pt, r)) return 1; break;
            case PO_EXPLORER_MALA:  if (mala_step(pt, r)) return 1; break;
            case PO_EXPLORER_NONE:  break;
            default: fail(pt, "oracle: explorer not implemented"); return 1;
            }
        }
    }
    if (is_target_explore(pt, r->chain) && (pt->cfg.record_online || pt->cfg.variational_first_tuning_round > 0 ||
                                            (uses_gradient_sampler(&pt->cfg) && pt->cfg.am_preconditioner != 0))) {
        for (int64_t i = 0; i < pt->d; i++) {       /* OnlineStateRecorder.jl:87-96 */
            mean_fit(&r->rec.on_mean[i], r->state[i]);
            var_fit(&r->rec.on_var[i], r->state[i]);
        }
        if (pt->cfg.record_online) {                /* :online sees extract_sample = [state; lp(state)], state.jl:79 */
            const double lp = lp_at_chain(pt, r->chain, r->state);
            mean_fit(&r->rec.on_mean[pt->d], lp);
            var_fit(&r->rec.on_var[pt->d], lp);
        }
    }
    return 0;
}

/* ========================================================================== */
/* communicate!  (src/pt/pigeons.jl:64-69, src/swap/swap.jl:6-39,106-126)       */
/* ========================================================================== */
typedef struct { double log_ratio, uniform; } swap_stat_t;   /* pair_swapper.jl:8-11 */

/* swap_stat, pair_swapper.jl:42-47 (TestSwapper :128) */
static int swap_stat(po_pt *pt, po_replica *r, int64_t partner, swap_stat_t *out) {
    if (pt->cfg.target == PO_TARGET_TEST_SWAPPER) {
        out->log_ratio = 0.0;
        out->uniform = po_rand(&r->rng);
        return 0;
    }
    /* log_unnormalized_ratio(lps, partner, my_chain, state), log_potentials.jl:43-51 */
    double lp_num, lp_den;
    if (pt->cfg.target == PO_TARGET_ISING) { lp_num = ising_lp(pt, partner, r->aux); lp_den = ising_lp(pt, r->chain, r->aux); }
    else { lp_num = lp_at_chain(pt, partner, r->state); lp_den = lp_at_chain(pt, r->chain, r->state); }
    double ans = lp_num - lp_den;
    if (isnan(ans)) { fail(pt, "Got NaN log-unnormalized ratio"); return 1; }
    out->log_ratio = ans;
    out->uniform = po_rand(&r->rng);
    return 0;
}
static inline double swap_acceptance_probability(const swap_stat_t *a, const swap_stat_t *b) {
    double e = exp(a->log_ratio + b->log_ratio);            /* pair_swapper.jl:88 */
    return e < 1.0 ? e : 1.0;
}
static int swap_decision(const po_pt *pt, int64_t c1, const swap_stat_t *s1, int64_t c2, const swap_stat_t *s2) {
    double uniform = c1 < c2 ? s1->uniform : s2->uniform;
    if (pt->cfg.target == PO_TARGET_TEST_SWAPPER) return uniform < pt->cfg.p0;   /* :135-138 */
    return uniform < swap_acceptance_probability(s1, s2);                          /* :81-85  */
}
/* _swap!, swap.jl:106-126 */
static void half_swap(po_pt *pt, po_replica *r, const swap_stat_t *mine, const swap_stat_t *theirs, int64_t partner) {
    const int64_t N = pt->N;
    int64_t my_chain = r->chain;
    if (pt->cfg.record_index_process) {
        if (r->rec.ip_len == r->rec.ip_cap) {
            r->rec.ip_cap = r->rec.ip_cap ? 2 * r->rec.ip_cap : 64;
            r->rec.ip = (int64_t *)realloc(r->rec.ip, sizeof(int64_t) * (size_t)r->rec.ip_cap);
        }
        r->rec.ip[r->rec.ip_len++] = r->chain;
    }
    if (pt->cfg.record_round_trip) round_trip_record(&r->rec.rt, is_reference_pt(pt, r->chain), is_target_swap(pt, r->chain));
    if (my_chain == partner) return;
    int do_swap = swap_decision(pt, my_chain, mine, partner, theirs);
    if (my_chain < partner && pt->cfg.target != PO_TARGET_TEST_SWAPPER) {
        /* record_swap_stats!, pair_swapper.jl:59-66 */
        double acc = swap_acceptance_probability(mine, theirs);
        mean_fit(&r->rec.swap_pr[my_chain], acc);
        logsum_fit(&r->rec.lsr_up[my_chain], mine->log_ratio);
        logsum_fit(&r->rec.lsr_dn[my_chain], theirs->log_ratio);
    }
    if (do_swap) r->chain = partner;
}
static int communicate(po_pt *pt) {
    const int64_t N = pt->N;
    int even = (pt->scan % 2 == 0);
    for (int64_t my_chain = 0; my_chain < N; my_chain++) {
        po_replica *me = &pt->replicas[pt->replica_of_chain[my_chain]];
        int64_t partner = partner_chain(N, even, my_chain);
        if (partner >= my_chain) {
            po_replica *other = &pt->replicas[pt->replica_of_chain[partner]];
            swap_stat_t mine, theirs;
            if (swap_stat(pt, me, partner, &mine)) return 1;
            if (partner == my_chain) theirs = mine;
            else if (swap_stat(pt, other, my_chain, &theirs)) return 1;
            half_swap(pt, me, &mine, &theirs, partner);
            if (partner != my_chain) half_swap(pt, other, &theirs, &mine, my_chain);
        }
    }
    for (int64_t r = 0; r < N; r++) pt->replica_of_chain[pt->replicas[r].chain] = r;   /* resort_replicas! */
    return 0;
}

/* ========================================================================== */
/* round loop                                                                 */
/* ========================================================================== */
/* concatenate_log_potentials (StabilizedPT.jl:67-69): variational leg reference -> target, then the fixed leg reversed */
static void assemble_betas(po_pt *pt) {
    const int64_t nv = pt->cfg.n_chains_variational, nf = pt->cfg.n_chains;
    for (int64_t i = 0; i < nv; i++) pt->betas[i] = pt->sched_var[i];
    for (int64_t i = 0; i < nf; i++) pt->betas[nv + i] = pt->sched_fix[nf - 1 - i];
}
po_pt *po_create(const po_config *cfg) {
    po_pt *pt = (po_pt *)calloc(1, sizeof(po_pt));
    pt->cfg = *cfg;
    const int64_t N = pt->N = cfg->n_chains + (cfg->n_chains_variational > 0 ? cfg->n_chains_variational : 0);   /* Inputs.jl:128 */
    const int64_t d = pt->d = (cfg->target == PO_TARGET_TEST_SWAPPER) ? 0 : cfg->dim;
    const int world = cfg->world_size > 0 ? cfg->world_size : 1;
    if (world > 1 && cfg->n_chains_variational > 0) { snprintf(pt->err, sizeof pt->err, "oracle: two-leg tempering is not sharded"); pt->failed = 1; return pt; }
    const int64_t K = pt->K = N / world;
    pt->c0 = K * cfg->rank;
    pt->shard_stat = (swap_stat_t_fwd *)calloc((size_t)K, sizeof(swap_stat_t_fwd));
    pt->replicas = (po_replica *)calloc((size_t)N, sizeof(po_replica));
    pt->replica_of_chain = (int64_t *)calloc((size_t)N, sizeof(int64_t));
    pt->betas = (double *)calloc((size_t)N, sizeof(double));
    /* equally_spaced_schedule, src/schedules/Schedule.jl:36-44 (range elements i/(N-1)) */
    if (cfg->n_chains_variational > 0) {                     /* StabilizedPT(inputs), StabilizedPT.jl:37-51 */
        const int64_t nv = cfg->n_chains_variational, nf = cfg->n_chains;
        pt->sched_var = (double *)calloc((size_t)nv, sizeof(double));
        pt->sched_fix = (double *)calloc((size_t)nf, sizeof(double));
        for (int64_t i = 0; i < nv; i++) pt->sched_var[i] = nv == 1 ? 1.0 : ((i == nv - 1) ? 1.0 : (double)i / (double)(nv - 1));
        for (int64_t i = 0; i < nf; i++) pt->sched_fix[i] = nf == 1 ? 1.0 : ((i == nf - 1) ? 1.0 : (double)i / (double)(nf - 1));
        assemble_betas(pt);
    } else if (N == 1) pt->betas[0] = 1.0;
    else for (int64_t i = 0; i < N; i++) pt->betas[i] = (i == N - 1) ? 1.0 : (double)i / (double)(N - 1);
    pt->step_size = cfg->am_step_size;
    /* _create_locals, src/replicas/replicas.jl:87-98; split_slice, src/utils/misc.jl:21-31 */
    po_rng master = po_rng_new(cfg->seed);
    for (int64_t g = 0; g < pt->c0; g++) (void)po_rng_split(&master);   /* split_slice burns the streams left of the slice */
    for (int64_t i = 0; i < K; i++) {
        po_replica *r = &pt->replicas[i];
        r->rng = po_rng_split(&master);
        r->chain = pt->c0 + i; r->replica_index = pt->c0 + i;
        pt->replica_of_chain[i] = i;
        r->state = (double *)calloc((size_t)(d > 0 ? d : 1), sizeof(double));
        r->buf = (double *)calloc((size_t)(8 * (d > 0 ? d : 1)), sizeof(double));
        rec_alloc(&r->rec, N, d);
        rec_empty(&r->rec, N, d);
        if (cfg->target == PO_TARGET_ISING) r->aux = ising_recompute(r->state, (int)llround(sqrt((double)d)));   /* falses(L, L), ising.jl:85 */
        /* funnel: initialization = zeros(dim) (test/supporting/dimensional-analysis.jl:24) -- calloc above */
        if (cfg->target == PO_TARGET_MVN) {
            /* initialization, src/targets/toy_mvn_target.jl:10-11 */
            double s = sqrt(cfg->p1);
            for (int64_t k = 0; k < d; k++) r->state[k] = po_randn(&r->rng);
            for (int64_t k = 0; k < d; k++) r->state[k] = r->state[k] / s;
        }
    }
    rec_alloc(&pt->reduced, N, d);
    rec_empty(&pt->reduced, N, d);
    pt->cb_x = (double *)calloc((size_t)N * 5, sizeof(double));
    pt->cb_y = pt->cb_x + N; pt->cb_m = pt->cb_y + N; pt->cb_c = pt->cb_m + N; pt->cb_d = pt->cb_c + N;
    return pt;
}
void po_destroy(po_pt *pt) {
    if (!pt) return;
    for (int64_t i = 0; i < pt->K; i++) { free(pt->replicas[i].state); free(pt->replicas[i].buf); rec_free(&pt->replicas[i].rec); }
    free(pt->shard_stat); free(pt->shard_ip_replica); free(pt->shard_ip_chain);
    rec_free(&pt->reduced);
    free(pt->replicas); free(pt->replica_of_chain); free(pt->betas); free(pt->target_std);
    free(pt->reduced_ip); free(pt->cb_x); free(pt->traces); free(pt->reduced_traces); free(pt->sched_var); free(pt->sched_fix); free(pt->vmean); free(pt->vstd);
    free(pt);
}

/* next_round!, src/pt/Iterators.jl:27-35 */
int po_begin_round(po_pt *pt) { pt->round += 1; pt->scan = 0; return 0; }

/* the `while next_scan!` loop of run_one_round!, src/pt/pigeons.jl:49-52 */
int po_run_scans(po_pt *pt, int64_t n_scans) {
    const int64_t N = pt->N;
    for (int64_t t = 0; t < n_scans; t++) {
        pt->scan += 1;
        traces_reserve(pt);
        int nt = pt->cfg.n_threads > 1 ? pt->cfg.n_threads : 1;
        (void)nt;
        /* explore!: @threads static over replicas sorted by chain (pigeons.jl:82-85) */
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
        for (int64_t c = 0; c < N; c++) {
            if (!pt->failed) explore_replica(pt, &pt->replicas[pt->replica_of_chain[c]]);
        }
        if (pt->failed) return 1;
        if (communicate(pt)) return 1;
    }
    return 0;
}

/* rejections, src/tempering/adaptation.jl:109-112 */
static void rejections(const po_pt *pt, double *r) {
    for (int64_t i = 0; i + 1 < pt->N; i++)
        r[i] = 1.0 - (pt->reduced.swap_pr[i].n > 0 ? pt->reduced.swap_pr[i].mu : 0.5);
}

/* optimal_schedule(_generator), communication_barriers, adaptation.jl:56-93, for one leg: `sched` (n grid points,
 * reference -> target) is replaced; rej[n-1] are the rejection rates of its pairs in the same order.  The cumulative
 * barrier interpolant of the OLD schedule is kept in (cbx, cby, cbm, cbc, cbd) when cbx != NULL. */
static int adapt_leg(po_pt *pt, int64_t N, const double *rej_in, double *sched, double *barrier_out,
                     double *cbx, double *cby, double *cbm, double *cbc, double *cbd) {
    if (N == 1) return 0;                                   /* NonReversiblePT.jl:53-55 */
    double *rej = (double *)malloc(sizeof(double) * (size_t)N * 10);
    double *x = rej + N, *xn = x + N, *m = xn + N, *c = m + N, *dd = c + N, *newb = dd + N, *work = newb + N, *tx = work + N, *ty = tx + N;
    memcpy(rej, rej_in, sizeof(double) * (size_t)(N - 1));
    for (int64_t i = 0; i + 1 < N; i++) if (!(rej[i] >= 0.0)) { free(rej); fail(pt, "Bad intensities"); return 1; }
    /* communication_barriers on the OLD schedule */
    double *bx = cbx ? cbx : tx, *by = cby ? cby : ty;
    bx[0] = sched[0]; by[0] = 0.0;
    double acc = 0.0;
    for (int64_t i = 0; i + 1 < N; i++) { acc += rej[i]; bx[i + 1] = sched[i + 1]; by[i + 1] = acc; }
    *barrier_out = acc;
    if (cbx) po_fc_build(cbx, cby, N, cbm, cbc, cbd);
    /* optimal_schedule_generator */
    int nudged = 0;
    for (;;) {
        memcpy(work, rej, sizeof(double) * (size_t)(N - 1));
        if (nudged) for (int64_t i = 0; i + 1 < N; i++) work[i] = rej[i] + 1e-6;
        x[0] = 0.0; acc = 0.0;
        for (int64_t i = 0; i + 1 < N; i++) { acc += work[i]; x[i + 1] = acc; }
        double norm = x[N - 1];
        int dup = 0;
        for (int64_t i = 0; i < N; i++) xn[i] = x[i] / norm;
        for (int64_t i = 0; i + 1 < N; i++) if (xn[i] == xn[i + 1]) dup = 1;
        if (dup) {
            if (nudged) { free(rej); fail(pt, "optimal_schedule: duplicate knots after nudge"); return 1; }
            nudged = 1; continue;
        }
        break;
    }
    po_fc_build(xn, sched, N, m, c, dd);
    newb[0] = 0.0; newb[N - 1] = 1.0;
    for (int64_t i = 1; i + 1 < N; i++) newb[i] = po_fc_eval(xn, sched, m, c, dd, N, (double)i / (double)(N - 1));
    /* Schedule constructor asserts, src/schedules/Schedule.jl:19-23 */
    for (int64_t i = 0; i + 1 < N; i++) if (!(newb[i] < newb[i + 1])) { free(rej); fail(pt, "Invalid schedule"); return 1; }
    memcpy(sched, newb, sizeof(double) * (size_t)N);
    free(rej);
    return 0;
}
static int adapt_tempering(po_pt *pt) {
    const int64_t N = pt->N;
    double *rej = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    rejections(pt, rej);
    int rc;
    if (pt->cfg.n_chains_variational > 0) {
        /* adapt_tempering(::StabilizedPT), StabilizedPT.jl:53-65: each leg from its own pairs; the fixed leg's pairs are
         * (N-1,N), (N-2,N-1), ... read from its reference towards the target (fixed_leg_indices(indexer)[2:end]) */
        const int64_t nv = pt->cfg.n_chains_variational, nf = pt->cfg.n_chains;
        double *rf = (double *)malloc(sizeof(double) * (size_t)(nf + 1));
        for (int64_t k = 0; k + 1 < nf; k++) rf[k] = rej[N - 2 - k];
        rc = adapt_leg(pt, nv, rej, pt->sched_var, &pt->global_barrier_var, NULL, NULL, NULL, NULL, NULL);
        if (!rc) rc = adapt_leg(pt, nf, rf, pt->sched_fix, &pt->global_barrier, pt->cb_x, pt->cb_y, pt->cb_m, pt->cb_c, pt->cb_d);
        if (!rc) { pt->cb_valid = 1; assemble_betas(pt); }
        free(rf);
    } else {
        rc = adapt_leg(pt, N, rej, pt->betas, &pt->global_barrier, pt->cb_x, pt->cb_y, pt->cb_m, pt->cb_c, pt->cb_d);
        if (!rc && N > 1) pt->cb_valid = 1;
    }
    free(rej);
    return rc;
}

/* stepping_stone_pair, src/evidence/stepping_stone.jl:28-43 */
static void stepping_stone(po_pt *pt) {
    double e1 = 0.0, e2 = 0.0;
    /* two legs: only the variational leg's keys (stepping_stone_keys(::StabilizedPT), stepping_stone.jl:53-65) */
    const int64_t np = pt->cfg.n_chains_variational > 0 ? pt->cfg.n_chains_variational : pt->N;
    for (int64_t i = 0; i + 1 < np; i++) {
        if (pt->reduced.lsr_up[i].n > 0) e1 += pt->reduced.lsr_up[i].value - log((double)pt->reduced.lsr_up[i].n);
        if (pt->reduced.lsr_dn[i].n > 0) e2 += pt->reduced.lsr_dn[i].value - log((double)pt->reduced.lsr_dn[i].n);
    }
    pt->stepping_stone[0] = e1; pt->stepping_stone[1] = -e2;
}

/* reduce_recorders! (src/recorders/recorders.jl:88-120) with the binary tree of
 * all_reduce_deterministically over replica indices (src/mpi_utils/Entangler.jl:188-251),
 * then adapt (src/pt/pigeons.jl:152-162). */
int po_end_round(po_pt *pt) {
    const int64_t N = pt->N, d = pt->d;
    int64_t n_scans = pt->scan;
    /* collect index_process first (disjoint keys) */
    if (pt->cfg.record_index_process) {
        if (pt->reduced_ip_cap < N * n_scans) {
            pt->reduced_ip_cap = N * n_scans;
            pt->reduced_ip = (int64_t *)realloc(pt->reduced_ip, sizeof(int64_t) * (size_t)pt->reduced_ip_cap);
        }
        for (int64_t r = 0; r < N; r++)
            memcpy(pt->reduced_ip + r * n_scans, pt->replicas[r].rec.ip, sizeof(int64_t) * (size_t)pt->replicas[r].rec.ip_len);
    }
    pt->reduced_n_scans = n_scans;
    { free(pt->reduced_traces); pt->reduced_traces = pt->traces; pt->traces = NULL; pt->traces_cap = 0;
      pt->reduced_traces_n = pt->traces_n; pt->traces_n = 0; }
    for (int64_t s = 1; s < N; s *= 2)
        for (int64_t i = 0; i + s < N; i += 2 * s)
            rec_merge(&pt->replicas[i].rec, &pt->replicas[i + s].rec, N, d);
    rec_empty(&pt->reduced, N, d);
    rec_merge(&pt->reduced, &pt->replicas[0].rec, N, d);
    pt->reduced.rt.state = 0;
    for (int64_t r = 0; r < N; r++) rec_empty(&pt->replicas[r].rec, N, d);
    pt->scan = 0;                                           /* Iterators.jl:43-45 */
    if (pt->cfg.target != PO_TARGET_TEST_SWAPPER) {
        stepping_stone(pt);                                 /* report uses pre-adapt recorders */
        if (adapt_tempering(pt)) return 1;
    }
    if (pt->cfg.variational_first_tuning_round > 0 && pt->round >= pt->cfg.variational_first_tuning_round &&
        pt->cfg.target == PO_TARGET_FUNNEL) {
        /* update_path_if_needed -> update_reference! (variational.jl:28-41, GaussianReference.jl:24-31): mean / std of
         * the target chains' _transformed_online statistics; the path's reference becomes the GaussianReference */
        if (!pt->vmean) { pt->vmean = (double *)calloc((size_t)d, sizeof(double)); pt->vstd = (double *)calloc((size_t)d, sizeof(double)); }
        for (int64_t i = 0; i < d; i++) { pt->vmean[i] = pt->reduced.on_mean[i].mu; pt->vstd[i] = sqrt(var_value(&pt->reduced.on_var[i])); }
        pt->v_active = 1;
    }
    if (uses_gradient_sampler(&pt->cfg)) {
        /* adapt_explorer(::AutoMALA), src/explorers/AutoMALA.jl:70-79; (::MALA) MALA.jl:63-69 adapts the preconditioner only */
        if (pt->cfg.am_preconditioner != 0) {               /* adapt_preconditioner, Preconditioner.jl:54-55 */
            if (!pt->target_std) pt->target_std = (double *)calloc((size_t)(d > 0 ? d : 1), sizeof(double));
            for (int64_t i = 0; i < d; i++) pt->target_std[i] = sqrt(var_value(&pt->reduced.on_var[i]));
        }
        double acc = 0.0; int64_t cnt = 0;
        for (int64_t c = 0; c < N; c++) if (pt->reduced.am_factors[c].n > 0) { acc += pt->reduced.am_factors[c].mu; cnt++; }
        if (cnt > 0) pt->step_size = pt->step_size * (acc / (double)cnt);      /* no am_factors without AutoMALA */
    }
    return 0;
}

int po_run_round(po_pt *pt) {
    po_begin_round(pt);
    if (po_run_scans(pt, (int64_t)1 << pt->round)) return 1;   /* n_scans_in_round, Iterators.jl:49 */
    return po_end_round(pt);
}

/* ========================================================================== */
/* getters                                                                    */
/* ========================================================================== */
void po_get_states(const po_pt *pt, double *x, int64_t *chain, uint64_t *rng) {
    for (int64_t r = 0; r < pt->K; r++) {
        if (x && pt->d > 0) memcpy(x + r * pt->d, pt->replicas[r].state, sizeof(double) * (size_t)pt->d);
        if (chain) chain[r] = pt->replicas[r].chain;
        if (rng) { rng[2 * r] = pt->replicas[r].rng.seed; rng[2 * r + 1] = pt->replicas[r].rng.gamma; }
    }
}
/* restore replicas from a checkpoint (src/pt/checkpoint.jl:19-54): state, chain (0-based), rng in replica / slot order */
void po_set_states(po_pt *pt, const double *x, const int64_t *chain, const uint64_t *rng) {
    for (int64_t s = 0; s < pt->K; s++) {
        po_replica *r = &pt->replicas[s];
        if (x && pt->d > 0) memcpy(r->state, x + s * pt->d, sizeof(double) * (size_t)pt->d);
        if (chain) { r->chain = chain[s]; pt->replica_of_chain[chain[s] - pt->c0] = s; }
        if (rng) { r->rng.seed = rng[2 * s]; r->rng.gamma = rng[2 * s + 1]; }
        if (x && pt->cfg.target == PO_TARGET_ISING) r->aux = ising_recompute(r->state, (int)llround(sqrt((double)pt->d)));
    }
}
void po_get_schedule(const po_pt *pt, double *b) { memcpy(b, pt->betas, sizeof(double) * (size_t)pt->N); }
void po_set_schedule(po_pt *pt, const double *b) {
    memcpy(pt->betas, b, sizeof(double) * (size_t)pt->N);
    if (pt->cfg.n_chains_variational > 0) {
        const int64_t nv = pt->cfg.n_chains_variational, nf = pt->cfg.n_chains;
        for (int64_t i = 0; i < nv; i++) pt->sched_var[i] = b[i];
        for (int64_t i = 0; i < nf; i++) pt->sched_fix[nf - 1 - i] = b[nv + i];
    }
}
void po_get_swap_pr(const po_pt *pt, double *mean, int64_t *n) {
    for (int64_t i = 0; i + 1 < pt->N; i++) { mean[i] = pt->reduced.swap_pr[i].mu; n[i] = pt->reduced.swap_pr[i].n; }
}
void po_get_log_sum_ratio(const po_pt *pt, double *up, int64_t *up_n, double *dn, int64_t *dn_n) {
    for (int64_t i = 0; i + 1 < pt->N; i++) {
        up[i] = pt->reduced.lsr_up[i].value; up_n[i] = pt->reduced.lsr_up[i].n;
        dn[i] = pt->reduced.lsr_dn[i].value; dn_n[i] = pt->reduced.lsr_dn[i].n;
    }
}
void po_get_round_trip(const po_pt *pt, int64_t *restarts, int64_t *trips) {
    *restarts = pt->reduced.rt.n_tempered_restarts; *trips = pt->reduced.rt.n_round_trips;
}
int64_t po_get_index_process(const po_pt *pt, int64_t *out) {
    if (out && pt->reduced_ip) memcpy(out, pt->reduced_ip, sizeof(int64_t) * (size_t)(pt->N * pt->reduced_n_scans));
    return pt->reduced_n_scans;
}
void po_get_explorer_stats(const po_pt *pt, double *acc_mean, int64_t *acc_n, double *steps_sum, int64_t *steps_n) {
    for (int64_t i = 0; i < pt->N; i++) {
        acc_mean[i] = pt->reduced.expl_acc[i].mu; acc_n[i] = pt->reduced.expl_acc[i].n;
        steps_sum[i] = pt->reduced.expl_steps[i].sum; steps_n[i] = pt->reduced.expl_steps[i].n;
    }
}
void po_get_am_stats(const po_pt *pt, double *fm, int64_t *fn, double *rm, int64_t *rn) {
    for (int64_t i = 0; i < pt->N; i++) {
        fm[i] = pt->reduced.am_factors[i].mu; fn[i] = pt->reduced.am_factors[i].n;
        rm[i] = pt->reduced.rev_rate[i].mu; rn[i] = pt->reduced.rev_rate[i].n;
    }
}
/* energy_ac1s(pt) (recorder.jl:156-173): cor[c] of (lp before, lp after) the exploration step at chain c; raw = {b0,b1,A00,A01,A11} */
void po_get_energy_ac1(const po_pt *pt, double *cor, int64_t *n, double *raw) {
    for (int64_t c = 0; c < pt->N; c++) {
        const po_cov2 *o = &pt->reduced.eac[c];
        if (cor) cor[c] = o->n > 1 ? cov2_cor12(o) : NAN;
        if (n) n[c] = o->n;
        if (raw) { raw[5 * c] = o->b[0]; raw[5 * c + 1] = o->b[1]; raw[5 * c + 2] = o->A[0]; raw[5 * c + 3] = o->A[1]; raw[5 * c + 4] = o->A[2]; }
    }
}
/* traces of the last round: out[scan][d+1] = [state; log density] of the target chain (record_traces == 2: out[scan][chain][d+1],
 * all N chains; rows of chains held by other shards stay zero); returns the number of scans */
int64_t po_get_traces(const po_pt *pt, double *out) {
    if (out && pt->reduced_traces) memcpy(out, pt->reduced_traces, sizeof(double) * (size_t)(pt->reduced_traces_n * traces_row_width(pt)));
    return pt->reduced_traces_n;
}
void po_get_online_lp(const po_pt *pt, double *mean, double *var, int64_t *n) {
    *mean = pt->reduced.on_mean[pt->d].mu; *var = var_value(&pt->reduced.on_var[pt->d]); *n = pt->reduced.on_mean[pt->d].n;
}
int64_t po_get_online(const po_pt *pt, double *mean, double *var) {
    for (int64_t i = 0; i < pt->d; i++) { mean[i] = pt->reduced.on_mean[i].mu; var[i] = var_value(&pt->reduced.on_var[i]); }
    return pt->d > 0 ? pt->reduced.on_mean[0].n : 0;
}
void po_get_stepping_stone(const po_pt *pt, double *pair) { pair[0] = pt->stepping_stone[0]; pair[1] = pt->stepping_stone[1]; }
double po_get_global_barrier(const po_pt *pt) { return pt->global_barrier; }
double po_get_global_barrier_variational(const po_pt *pt) { return pt->global_barrier_var; }
int po_get_variational(const po_pt *pt, double *mean, double *std) {
    if (pt->v_active) { memcpy(mean, pt->vmean, sizeof(double) * (size_t)pt->d); memcpy(std, pt->vstd, sizeof(double) * (size_t)pt->d); }
    return pt->v_active;
}
double po_cumulative_barrier(const po_pt *pt, double beta) {
    if (!pt->cb_valid) return NAN;
    return po_fc_eval(pt->cb_x, pt->cb_y, pt->cb_m, pt->cb_c, pt->cb_d, pt->N, beta);
}
double po_get_step_size(const po_pt *pt) { return pt->step_size; }
int64_t po_get_target_std(const po_pt *pt, double *out) {
    if (!pt->target_std) return 0;
    memcpy(out, pt->target_std, sizeof(double) * (size_t)pt->d);
    return pt->d;
}
void po_set_explorer_adaptation(po_pt *pt, double step_size, const double *target_std) {
    pt->step_size = step_size;
    if (target_std) {
        if (!pt->target_std) pt->target_std = (double *)calloc((size_t)(pt->d > 0 ? pt->d : 1), sizeof(double));
        memcpy(pt->target_std, target_std, sizeof(double) * (size_t)pt->d);
    }
}

/* ========================================================================== */
/* chain-sharded operation: test-only restatement of the multi-GPU protocol    */
/* (DESIGN.md 9).  Recorders stay with the slot (rank-local accumulators, as in */
/* the HIP engine); state, rng, replica index and round-trip state travel.      */
/* ========================================================================== */
static void shard_active(const po_pt *pt, int even, int32_t *active) {
    active[0] = (pt->c0 > 0 && partner_chain(pt->N, even, pt->c0) == pt->c0 - 1) ? 1 : 0;
    active[1] = (pt->c0 + pt->K < pt->N && partner_chain(pt->N, even, pt->c0 + pt->K - 1) == pt->c0 + pt->K) ? 1 : 0;
}
void po_shard_info(const po_pt *pt, int64_t *c0, int64_t *K, int64_t *n_pairs) {
    *c0 = pt->c0; *K = pt->K; *n_pairs = (pt->c0 + pt->K < pt->N) ? pt->K : pt->K - 1;
}
int po_shard_explore(po_pt *pt, int64_t scan) {
    pt->scan = scan;
    traces_reserve(pt);
    for (int64_t cl = 0; cl < pt->K; cl++)
        if (explore_replica(pt, &pt->replicas[pt->replica_of_chain[cl]])) return 1;
    return 0;
}
int po_shard_swap_begin(po_pt *pt, int64_t scan, double *stats_out, int32_t *active_out) {
    const int64_t N = pt->N, K = pt->K;
    int even = (scan % 2 == 0);
    pt->scan = scan;
    if (pt->shard_ip_len + K > pt->shard_ip_cap) {
        pt->shard_ip_cap = pt->shard_ip_cap ? 2 * pt->shard_ip_cap : 64 * K;
        pt->shard_ip_replica = (int64_t *)realloc(pt->shard_ip_replica, sizeof(int64_t) * (size_t)pt->shard_ip_cap);
        pt->shard_ip_chain = (int64_t *)realloc(pt->shard_ip_chain, sizeof(int64_t) * (size_t)pt->shard_ip_cap);
    }
    for (int64_t cl = 0; cl < K; cl++) {
        int64_t slot = pt->replica_of_chain[cl];
        po_replica *r = &pt->replicas[slot];
        swap_stat_t st;
        if (swap_stat(pt, r, partner_chain(N, even, pt->c0 + cl), &st)) return 1;
        pt->shard_stat[cl].log_ratio = st.log_ratio; pt->shard_stat[cl].uniform = st.uniform;
        pt->shard_ip_replica[pt->shard_ip_len + slot] = r->replica_index;
        pt->shard_ip_chain[pt->shard_ip_len + slot] = r->chain;
        if (pt->cfg.record_round_trip) round_trip_record(&r->rec.rt, is_reference_pt(pt, r->chain), is_target_swap(pt, r->chain));
    }
    pt->shard_ip_len += K;
    stats_out[0] = pt->shard_stat[0].log_ratio; stats_out[1] = pt->shard_stat[0].uniform;
    stats_out[2] = pt->shard_stat[K - 1].log_ratio; stats_out[3] = pt->shard_stat[K - 1].uniform;
    shard_active(pt, even, active_out);
    return 0;
}
int po_shard_swap_finish(po_pt *pt, int64_t scan, const double *nbr, int32_t *accepted) {
    const int64_t N = pt->N, K = pt->K;
    int even = (scan % 2 == 0);
    accepted[0] = accepted[1] = 0;
    for (int64_t cl = 0; cl < K; cl++) {
        int64_t c = pt->c0 + cl, pc = partner_chain(N, even, c);
        if (pc == c) continue;
        po_replica *r = &pt->replicas[pt->replica_of_chain[cl]];
        int local = (pc >= pt->c0 && pc < pt->c0 + K);
        int side = pc < c ? 0 : 1;
        swap_stat_t mine = { pt->shard_stat[cl].log_ratio, pt->shard_stat[cl].uniform }, theirs;
        if (local) { theirs.log_ratio = pt->shard_stat[pc - pt->c0].log_ratio; theirs.uniform = pt->shard_stat[pc - pt->c0].uniform; }
        else { theirs.log_ratio = nbr[2 * side]; theirs.uniform = nbr[2 * side + 1]; }
        int do_swap = swap_decision(pt, c, &mine, pc, &theirs);
        if (c < pc && pt->cfg.target != PO_TARGET_TEST_SWAPPER) {
            double acc = swap_acceptance_probability(&mine, &theirs);
            mean_fit(&r->rec.swap_pr[c], acc);
            logsum_fit(&r->rec.lsr_up[c], mine.log_ratio);
            logsum_fit(&r->rec.lsr_dn[c], theirs.log_ratio);
        }
        if (do_swap) { if (local) r->chain = pc; else accepted[side] = 1; }
    }
    for (int64_t s = 0; s < K; s++) pt->replica_of_chain[pt->replicas[s].chain - pt->c0] = s;
    return 0;
}
int64_t po_shard_payload_words(const po_pt *pt) { return pt->d + 6; }
/* payload: [0..d) state, d: (unused: the HIP engine ships sum x^2 here), d+1,d+2: rng, d+3: replica id, d+4: round-trip state */
void po_shard_export(po_pt *pt, int side, double *buf) {
    po_replica *r = &pt->replicas[pt->replica_of_chain[side == 0 ? 0 : pt->K - 1]];
    memcpy(buf, r->state, sizeof(double) * (size_t)pt->d);
    buf[pt->d] = pt->cfg.target == PO_TARGET_ISING ? (double)r->aux : (pt->d > 0 ? po_sqr_norm(r->state, pt->d) : 0.0);
    uint64_t w[4] = { r->rng.seed, r->rng.gamma, (uint64_t)r->replica_index, (uint64_t)r->rec.rt.state };
    memcpy(buf + pt->d + 1, w, sizeof w);
}
void po_shard_import(po_pt *pt, int side, const double *buf) {
    po_replica *r = &pt->replicas[pt->replica_of_chain[side == 0 ? 0 : pt->K - 1]];
    memcpy(r->state, buf, sizeof(double) * (size_t)pt->d);
    uint64_t w[4];
    memcpy(w, buf + pt->d + 1, sizeof w);
    r->rng.seed = w[0]; r->rng.gamma = w[1]; r->replica_index = (int64_t)w[2]; r->rec.rt.state = (int64_t)w[3];
    if (pt->cfg.target == PO_TARGET_ISING) r->aux = (int64_t)buf[pt->d];
}
int po_shard_reduce(po_pt *pt) {
    const int64_t N = pt->N, d = pt->d;
    rec_empty(&pt->reduced, N, d);
    for (int64_t s = 0; s < pt->K; s++) rec_merge(&pt->reduced, &pt->replicas[s].rec, N, d);
    for (int64_t s = 0; s < pt->K; s++) {
        int64_t st = pt->replicas[s].rec.rt.state;          /* the state machine resets every round ...  */
        rec_empty(&pt->replicas[s].rec, N, d);
        (void)st;                                           /* ... including its state (RoundTripRecorder.jl:30-34) */
    }
    pt->reduced_n_scans = pt->shard_ip_len / pt->K;
    { free(pt->reduced_traces); pt->reduced_traces = pt->traces; pt->traces = NULL; pt->traces_cap = 0;
      pt->reduced_traces_n = pt->traces_n; pt->traces_n = 0; }
    return 0;
}
void po_shard_replica_ids(const po_pt *pt, int64_t *out) { for (int64_t s = 0; s < pt->K; s++) out[s] = pt->replicas[s].replica_index; }
int64_t po_shard_index_process(const po_pt *pt, int64_t *replica, int64_t *chain) {
    int64_t n = pt->reduced_n_scans;
    if (replica) memcpy(replica, pt->shard_ip_replica, sizeof(int64_t) * (size_t)(n * pt->K));
    if (chain) memcpy(chain, pt->shard_ip_chain, sizeof(int64_t) * (size_t)(n * pt->K));
    ((po_pt *)pt)->shard_ip_len = 0;
    return n;
}
