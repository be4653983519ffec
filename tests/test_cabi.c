/* test_cabi.c -- a plain C99 program against include/pte.h, linked to libpte.so: proves that the drop-in boundary is C
 * (not merely ctypes-compatible) by running BASELINE configs[0] through it -- toy_mvn_target(2), n_chains = 10, n_rounds = 5,
 * SliceSampler, seed 1 (the Pigeons.jl quickstart) -- round by round against tests/golden/cabi_c1.txt (written by
 * tools/gen_golden.py from the CPU oracle: the schedule in force during each round, that round's swap acceptance and index
 * process, the final replicas).  Integers exact, swap acceptance 1e-9 relative, states 1e-12 relative.
 *
 *   gcc -std=c99 -Wall -Wextra -Werror -pedantic -I include tests/test_cabi.c -L pigeons.jl_amd/lib -lpte -Wl,-rpath,... -lm
 *   ./test_cabi tests/golden/cabi_c1.txt          exit code 0 = every comparison held
 * Built and run by tests/test_cabi.py (the compile + link on any machine, the run on a GPU). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pte.h"

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++failures; } } while (0)
#define PTE_OK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, pte_last_error(h)); return 2; } } while (0)

static int expect_word(FILE *f, const char *w) {
    char buf[64];
    if (fscanf(f, "%63s", buf) != 1 || strcmp(buf, w) != 0) { fprintf(stderr, "fixture: expected '%s', got '%s'\n", w, buf); return 0; }
    return 1;
}
static int read_doubles(FILE *f, double *out, int64_t n) {
    char buf[64];
    for (int64_t i = 0; i < n; ++i) { if (fscanf(f, "%63s", buf) != 1) return 0; out[i] = strtod(buf, NULL); }   /* C99 hex floats */
    return 1;
}
static int read_i64(FILE *f, int64_t *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { long long v; if (fscanf(f, "%lld", &v) != 1) return 0; out[i] = (int64_t)v; }
    return 1;
}
static int read_u64(FILE *f, uint64_t *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { unsigned long long v; if (fscanf(f, "%llu", &v) != 1) return 0; out[i] = (uint64_t)v; }
    return 1;
}
static int close_rel(double a, double b, double rtol) { return fabs(a - b) <= rtol * fmax(fabs(a), fabs(b)) || (a == b); }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s tests/golden/cabi_c1.txt\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    long long N, d, seed, R;
    if (!expect_word(f, "config") || fscanf(f, "%lld %lld %lld %lld", &N, &d, &seed, &R) != 4) return 2;

    pte_engine *h = NULL;
    pte_config cfg;
    if (pte_default_config(&cfg) != 0) { fprintf(stderr, "pte_default_config failed\n"); return 2; }
    CHECK(cfg.struct_size == sizeof(pte_config) && cfg.abi_version == PTE_ABI_VERSION, "default config: size %u abi %u", cfg.struct_size, cfg.abi_version);
    cfg.target = PTE_TARGET_MVN_SCALED_PRECISION;
    cfg.explorer = PTE_EXPLORER_SLICE;
    cfg.n_chains = N; cfg.dim = d; cfg.seed = (uint64_t)seed;
    cfg.max_scans_per_round = (int64_t)1 << R;
    cfg.record_flags = PTE_RECORD_ROUND_TRIP | PTE_RECORD_INDEX_PROCESS;
    if (pte_create(&cfg, &h) != 0) { fprintf(stderr, "pte_create failed: %s\n", pte_last_error(NULL)); return 3; }
    printf("kernel: %s\n", pte_kernel_name(h));

    double *sched = malloc(sizeof(double) * (size_t)N), *mean = malloc(sizeof(double) * (size_t)N), *want_mean = malloc(sizeof(double) * (size_t)N);
    int64_t *cnt = malloc(sizeof(int64_t) * (size_t)N), *want_cnt = malloc(sizeof(int64_t) * (size_t)N);
    int64_t *ip = malloc(sizeof(int64_t) * (size_t)(N << R)), *want_ip = malloc(sizeof(int64_t) * (size_t)(N << R));
    for (long long r = 1; r <= R; ++r) {
        long long rr;
        if (!expect_word(f, "round") || fscanf(f, "%lld", &rr) != 1 || rr != r) return 2;
        const int64_t scans = (int64_t)1 << r;
        if (!expect_word(f, "schedule") || !read_doubles(f, sched, N)) return 2;
        PTE_OK(pte_set_schedule(h, sched, N));                       /* discretize(adapt_tempering(...)): the host's job, here replayed */
        PTE_OK(pte_run_scans(h, 1, scans));                          /* while next_scan!(pt): explore!; communicate! */
        PTE_OK(pte_reduce(h));                                       /* reduce_recorders! */
        PTE_OK(pte_get_swap_acceptance(h, mean, cnt));
        int64_t got_scans = 0;
        PTE_OK(pte_get_index_process(h, ip, &got_scans));
        if (!expect_word(f, "swap_mean") || !read_doubles(f, want_mean, N - 1)) return 2;
        if (!expect_word(f, "swap_n") || !read_i64(f, want_cnt, N - 1)) return 2;
        if (!expect_word(f, "index_process") || !read_i64(f, want_ip, N * scans)) return 2;
        CHECK(got_scans == scans, "round %lld: %lld scans recorded, expected %lld", r, (long long)got_scans, (long long)scans);
        for (long long i = 0; i < N - 1; ++i) {
            CHECK(cnt[i] == want_cnt[i], "round %lld pair %lld: swap count %lld != %lld", r, i, (long long)cnt[i], (long long)want_cnt[i]);
            CHECK(close_rel(mean[i], want_mean[i], 1e-9), "round %lld pair %lld: swap acceptance %.17g != %.17g", r, i, mean[i], want_mean[i]);
        }
        int bad = 0;
        for (long long i = 0; i < N * scans; ++i) bad += ip[i] != want_ip[i];
        CHECK(bad == 0, "round %lld: index process differs in %d entries", r, bad);
    }
    int64_t *chain = malloc(sizeof(int64_t) * (size_t)N), *want_chain = malloc(sizeof(int64_t) * (size_t)N);
    uint64_t *rng = malloc(sizeof(uint64_t) * 2 * (size_t)N), *want_rng = malloc(sizeof(uint64_t) * 2 * (size_t)N);
    double *x = malloc(sizeof(double) * (size_t)(N * d)), *want_x = malloc(sizeof(double) * (size_t)(N * d));
    PTE_OK(pte_get_state(h, x, chain, rng));
    if (!expect_word(f, "final_chain") || !read_i64(f, want_chain, N)) return 2;
    if (!expect_word(f, "final_rng") || !read_u64(f, want_rng, 2 * N)) return 2;
    if (!expect_word(f, "final_state") || !read_doubles(f, want_x, N * d)) return 2;
    for (long long i = 0; i < N; ++i) {
        CHECK(chain[i] == want_chain[i], "replica %lld: chain %lld != %lld", i, (long long)chain[i], (long long)want_chain[i]);
        CHECK(rng[2 * i] == want_rng[2 * i] && rng[2 * i + 1] == want_rng[2 * i + 1], "replica %lld: rng state differs", i);
    }
    for (long long i = 0; i < N * d; ++i) CHECK(close_rel(x[i], want_x[i], 1e-12), "state[%lld] = %.17g != %.17g", i, x[i], want_x[i]);
    int64_t restarts = -1, trips = -1;
    PTE_OK(pte_get_round_trip(h, &restarts, &trips));
    CHECK(restarts >= 0 && trips >= 0, "round trip counters %lld %lld", (long long)restarts, (long long)trips);
    PTE_OK(pte_destroy(h));
    fclose(f);
    if (failures) { fprintf(stderr, "test_cabi: %d comparisons failed\n", failures); return 1; }
    printf("test_cabi: C1 (toy_mvn_target(%lld), n_chains = %lld, %lld rounds, SliceSampler) matches the fixture through the C ABI\n", d, N, R);
    return 0;
}
