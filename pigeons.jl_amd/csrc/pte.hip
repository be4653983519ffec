// pte.hip -- host side of the C ABI declared in include/pte.h (libpte.so).
// Owns device memory, one HIP stream per engine, launches the kernels of pte_kernels.hpp.
// There is no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pte.h"
#include "pte_kernels.hpp"
// The headers of the earlier SliceSampler generations hold helpers the default kernel shares (draw buffer, filtered
// predicates, exact sequential fallback, window tables); their kernels are templates and are only instantiated --
// i.e. only exist in the library -- in the test build (-DPTE_TEST_KERNELS, libpte_test.so), see launch_explorer_kind.
#include "pte_slice2.hpp"
#include "pte_slice5.hpp"
#include "pte_slice7.hpp"
#include "pte_slice8.hpp"
#ifdef PTE_SPLIT_LANGEVIN          // the product build: the Langevin-family kernels are the library's second translation unit (pte_langevin.hip)
#include "pte_automala_params.hpp"
#else                              // tools / development builds: one translation unit
#include "pte_langevin_launch.hpp"
#endif
#include "pte_ising.hpp"
#if defined(PTE_PROFILE_AM)               // debug builds only (tools/prof_automala.py): 12 words per wave, section times of k_explore_automala
#define PTE_WAVE_PROFILE_WORDS 12
#elif defined(PTE_PROFILE_WAVES)          // debug builds only (tools/prof_waves.py): per-wave start / end / placement of the explore kernel
#define PTE_WAVE_PROFILE_WORDS 4
#else
#define PTE_WAVE_PROFILE_WORDS 0
#endif
#include "pte_comm.hpp"

using namespace pte;

namespace {

thread_local std::string g_create_error;

struct Snapshot {   // reduced recorders of the last round, host side
    std::vector<double> swap_mean; std::vector<int64_t> swap_n;
    std::vector<double> lsr_up, lsr_dn; std::vector<int64_t> lsr_n;
    int64_t restarts = 0, trips = 0;
    std::vector<double> acc_mean, steps_sum; std::vector<int64_t> acc_n, steps_n;
    std::vector<double> on_mean, on_var; int64_t on_n = 0;   // d + 1 entries: [state; log density] (two legs: both targets merged)
    std::vector<double> eac_cor, eac_raw; std::vector<int64_t> eac_n;   // energy_ac1 per local chain
    std::vector<double> traces; int64_t traces_n = 0;        // [scan][d+1]
    std::vector<int32_t> ip_chain, ip_replica;   // [scan][slot]
    int64_t n_scans = 0;
};

}  // namespace

struct pte_engine {
    pte_config cfg{};
    EngineDev dev{};
    hipStream_t stream = nullptr;
    int nlu = 0;
    int slice_impl = 8, slice_m = 4;   // from pte_config.debug_kernel: 8 offset speculation, straight-line (default) | 1 plain sequential; test build: 2, 5, 7
    int ising_impl = 0;                // 0 lane-speculative bit-packed sweep (default) | test build: 1 scalar bit-packed, 2 byte lattice
    // transport behind the ABI (pte_comm_*)
    int comm_kind = 0;                 // 0 none, 1 RCCL communicator over the ranks
    ncclComm_t nccl = nullptr;
    int n_ranks_seen = 0;
    double *own_msg = nullptr;         // engine-owned message buffers [4][d+8] when the caller set none
    double *d_coll = nullptr;          // [16] device staging of the small host collectives
    hipEvent_t ev_pack = nullptr, ev_copied = nullptr;   // in-process group transport (pte_group_run_scans)
    int64_t N = 0, d = 0;          // global chains, state dimension
    int64_t K = 0, c0 = 0;         // local chains [c0, c0+K)
    int world = 1, rank = 0;
    int32_t *slot_map[2] = {nullptr, nullptr};   // ping-pong buffers of slot_of_chain (two-phase swap)
    int slot_cur = 0;
    double *d_payload = nullptr;   // staging buffer for boundary export/import
    double *msg_send[2] = {nullptr, nullptr}, *msg_recv[2] = {nullptr, nullptr};   // device-resident exchange (caller-owned)
    int64_t *d_napplied = nullptr;  // [2] boundary swaps applied on the device path
    double *d_vref = nullptr;       // [5 d] GaussianReference: mean, std, c0, i2, gf
    int32_t *d_vuse = nullptr;      // [N]   chains whose path starts at it
    std::vector<double> betas;
    std::vector<void *> allocs;
    double *d_nhp = nullptr, *d_sd = nullptr, *d_nprec = nullptr, *d_beta = nullptr, *d_target_std = nullptr;
    bool have_target_std = false;
    double step_size = 1.0;
    int am_n_refresh = 0;
    std::vector<double> fac_mean, rev_mean; std::vector<int64_t> fac_n, rev_n;
    int64_t scans_in_round = 0;      // scans run since the last pte_reduce
    Snapshot snap;
    std::string err;
    // timing
    int timing = 0;                  // 0 off, 1 every kernel, 2 explore kernels only
    bool ev_open = false;
    struct Ev { hipEvent_t a, b; int kernel; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> ev_pool;
    double init_ms = -1.0;            // duration of k_init (create_replicas), -1: not launched (Ising, TestSwapper)
    bool ev_ext = false, ev_ext_done = false;   // the open bracket's events ride on its ONE kernel launch (PTE_LAUNCH1)
    double t_ms[5] = {0, 0, 0, 0, 0};    // by kind: 0 explore, 1 swap, (2 = k_init: init_ms), 3 boundary exchange (ncclGroupStart .. ncclGroupEnd), 4 the fused scan loop (one launch per pte_run_scans)
    int64_t t_n[5] = {0, 0, 0, 0, 0};
    std::vector<float> t_samples[5];  // per-launch durations since the last reset (spread of the timed region)
    int64_t t_scans4 = 0;             // scans inside the timed launches of kind 4
    // one launch per pte_run_scans (k_scans_*: pte_kernels.hpp "ScanLoop"): pairwise hand-shakes instead of a launch boundary per scan
    bool fused_allowed = true;        // pte_config.debug_kernel & PTE_KERNEL_TWO_LAUNCHES clears it
    int n_cus = -1;                   // compute units of the device (lazily)
    bool fused_wg_allowed = true;     // ... & PTE_KERNEL_SCAN_LOOP_ONE_CHAIN clears it
    int fused_wg = 1;                 // chains (waves) per workgroup of the scan-loop kernel this engine launches
    int64_t fused_limit = -1;         // workgroups of the scan-loop kernel the device holds at once (-1: not asked yet, 0: not available)
    unsigned long long *hs_flag = nullptr; double *hs_pub = nullptr;   // [K], [K][2][4]
    unsigned long long hs_epoch = 0;  // epochs handed out so far (monotone; pte_set_state re-bases the flags of a poisoned engine to it)
    // forward progress of that launch (round 6; pte_kernels.hpp scan_loop_gate): no workgroup starts before all have arrived, or the call falls back
    unsigned long long *hs_gate = nullptr;                 // device [2]: arrivals so far, decision word of the last launch
    unsigned long long gate_seq = 0, gate_arrived = 0;     // launches so far; workgroups they held (every workgroup of a finished launch has arrived)
    int64_t fused_calls = 0, gate_aborts = 0;              // pte_run_scans calls that ran as one launch; launches that found a workgroup missing and fell back
    int64_t fused_skip = 0, fused_backoff = 0;             // after an abort the next `fused_skip` calls go straight to explore + swap launches (1, 2, 4, ... 256)
    bool poisoned = false;            // a device error inside the one-launch scan loop: chains stopped at different scans, only pte_set_state / pte_destroy accepted
    int test_fault = 0;               // test build only: PTE_KERNEL_TEST_DEAD_CHAIN (1), PTE_KERNEL_TEST_LATE_WORKGROUP (2)
};

namespace {

int fail(pte_engine *h, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return 1;
}
#define HIP_OK(h, call)                                                                          \
    do { hipError_t e_ = (call);                                                                 \
         if (e_ != hipSuccess) return fail(h, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

template <typename T>
int dev_alloc(pte_engine *h, T **p, size_t n, bool zero = true) {
    void *q = nullptr;
    size_t bytes = sizeof(T) * (n ? n : 1);
    HIP_OK(h, hipMalloc(&q, bytes));
    h->allocs.push_back(q);
    if (zero) HIP_OK(h, hipMemsetAsync(q, 0, bytes, h->stream));
    *p = (T *)q;
    return 0;
}

int next_pow2_log(int64_t n) { int l = 0; while (((int64_t)1 << l) < n) ++l; return l; }

// One kernel launch that may carry the timing events of the bracket it stands in (time_begin(h, kind, true) ... time_end(h)): with
// hipExtLaunchKernelGGL the start / stop events take the kernel's own begin / end timestamps -- what rocprofv3's kernel trace reports --
// instead of bracketing it with two event-record commands, which add the stream's hand-over time around a ~85 us kernel (~7 us).
#define PTE_LAUNCH1(KERNEL, grid, block, shmem, stream, ...)                                                                      \
    do {                                                                                                                          \
        if (h->ev_open && h->ev_ext && !h->ev_ext_done) {                                                                         \
            hipExtLaunchKernelGGL(KERNEL, grid, block, shmem, stream, h->events.back().a, h->events.back().b, 0, __VA_ARGS__);    \
            h->ev_ext_done = true;                                                                                                \
        } else hipLaunchKernelGGL(KERNEL, grid, block, shmem, stream, __VA_ARGS__);                                               \
    } while (0)

#ifdef PTE_DEV_FEW_NLU   // development builds only (tools/build_variant.sh): the tree depths of d = 1024 and d = 4096, a fifth of the compile time
#define DISPATCH_NLU_M(nlu, KERNEL, MM, grid, block, stream, ...)                                 \
    switch (nlu) {                                                                               \
    case 4: PTE_LAUNCH1((KERNEL<4, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 6: PTE_LAUNCH1((KERNEL<6, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    default: fprintf(stderr, "PTE_DEV_FEW_NLU build: d must be 1024 or 4096\n"); abort();        \
    }
#define DISPATCH_NLU(nlu, KERNEL, grid, block, stream, ...)                                      \
    switch (nlu) {                                                                               \
    case 4: PTE_LAUNCH1(KERNEL<4>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 6: PTE_LAUNCH1(KERNEL<6>, grid, block, 0, stream, __VA_ARGS__); break;           \
    default: fprintf(stderr, "PTE_DEV_FEW_NLU build: d must be 1024 or 4096\n"); abort();        \
    }
#else
#define DISPATCH_NLU_M(nlu, KERNEL, MM, grid, block, stream, ...)                                 \
    switch (nlu) {                                                                               \
    case 0: PTE_LAUNCH1((KERNEL<0, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 1: PTE_LAUNCH1((KERNEL<1, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 2: PTE_LAUNCH1((KERNEL<2, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 3: PTE_LAUNCH1((KERNEL<3, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 4: PTE_LAUNCH1((KERNEL<4, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    case 5: PTE_LAUNCH1((KERNEL<5, MM>), grid, block, 0, stream, __VA_ARGS__); break;     \
    default: PTE_LAUNCH1((KERNEL<6, MM>), grid, block, 0, stream, __VA_ARGS__); break;    \
    }

#define DISPATCH_NLU(nlu, KERNEL, grid, block, stream, ...)                                      \
    switch (nlu) {                                                                               \
    case 0: PTE_LAUNCH1(KERNEL<0>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 1: PTE_LAUNCH1(KERNEL<1>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 2: PTE_LAUNCH1(KERNEL<2>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 3: PTE_LAUNCH1(KERNEL<3>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 4: PTE_LAUNCH1(KERNEL<4>, grid, block, 0, stream, __VA_ARGS__); break;           \
    case 5: PTE_LAUNCH1(KERNEL<5>, grid, block, 0, stream, __VA_ARGS__); break;           \
    default: PTE_LAUNCH1(KERNEL<6>, grid, block, 0, stream, __VA_ARGS__); break;          \
    }

#endif

// discretize(path, schedule): per-chain constants of ScaledPrecisionNormalLogPotential
// (reference src/paths/ScaledPrecisionNormalPath.jl:45-48, src/schedules/discretize.jl:6-7).
int upload_ladder(pte_engine *h) {
    const int64_t N = h->N;
    std::vector<double> nhp(N, 0.0), sd(N, 1.0), nprec(N, 0.0);
    if (h->cfg.target == PTE_TARGET_MVN_SCALED_PRECISION) {
        const double p0 = h->cfg.target_params[0], p1 = h->cfg.target_params[1];
        for (int64_t c = 0; c < N; ++c) {
            const double beta = h->betas[c];
            const double prec = (1.0 - beta) * p0 + beta * p1;
            nhp[c] = -0.5 * prec;
            nprec[c] = -prec;
            sd[c] = std::sqrt(prec);
        }
    } else if (h->cfg.target == PTE_TARGET_FUNNEL) {
        for (int64_t c = 0; c < N; ++c) sd[c] = std::sqrt(h->cfg.target_params[0]);   // reference end point
    }
    HIP_OK(h, hipMemcpyAsync(h->d_nhp, nhp.data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->d_sd, sd.data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->d_nprec, nprec.data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->d_beta, h->betas.data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int reset_recorders(pte_engine *h) {
    EngineDev &e = h->dev;
    const int64_t N = h->K, np = h->K;       // per-shard sizes: K slots, K pairs keyed by their lower chain
    HIP_OK(h, hipMemsetAsync(e.swap_sum, 0, sizeof(double) * np, h->stream));
    HIP_OK(h, hipMemsetAsync(e.swap_n, 0, sizeof(int64_t) * np, h->stream));
    std::vector<double> ninf(np, -INFINITY);
    HIP_OK(h, hipMemcpyAsync(e.lsr_up, ninf.data(), sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(e.lsr_dn, ninf.data(), sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemsetAsync(e.lsr_n, 0, sizeof(int64_t) * np, h->stream));
    HIP_OK(h, hipMemsetAsync(e.rt_state, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.rt_restarts, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.rt_trips, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.expl_acc_sum, 0, sizeof(double) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.expl_acc_n, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.expl_steps_sum, 0, sizeof(double) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.expl_steps_n, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.am_fac_sum, 0, sizeof(double) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.am_fac_n, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.am_rev_sum, 0, sizeof(double) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.am_rev_n, 0, sizeof(int64_t) * N, h->stream));
    const int64_t dd = h->d > 0 ? h->d : 1;
    HIP_OK(h, hipMemsetAsync(e.on_mean, 0, sizeof(double) * 2 * (h->d + 1), h->stream));
    HIP_OK(h, hipMemsetAsync(e.on_m2, 0, sizeof(double) * 2 * (h->d + 1), h->stream));
    HIP_OK(h, hipMemsetAsync(e.eac, 0, sizeof(double) * 5 * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.eac_n, 0, sizeof(int64_t) * N, h->stream));
    HIP_OK(h, hipMemsetAsync(e.on_n, 0, 2 * sizeof(int64_t), h->stream));
    if (e.am_log)       // 0x7f7f = "no search here"
        HIP_OK(h, hipMemsetAsync(e.am_log, 0x7F, sizeof(int16_t) * (size_t)(h->cfg.max_scans_per_round * h->K * e.am_log_cap), h->stream));
    if (e.eac_log) HIP_OK(h, hipMemsetAsync(e.eac_log, 0xFF, sizeof(double) * 2 * (size_t)(h->cfg.max_scans_per_round * h->K), h->stream));      // all-ones = "no explore step recorded here"
    if (e.swap_log)     // all-ones words = "this pair was idle at this scan" (no log ratio has that bit pattern: a NaN ratio is ERR_NAN_RATIO)
        HIP_OK(h, hipMemsetAsync(e.swap_log, 0xFF, sizeof(double) * 2 * (size_t)(h->cfg.max_scans_per_round * h->K), h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));   // `ninf` must outlive the copies
    h->scans_in_round = 0;
    return 0;
}

int check_device_error(pte_engine *h) {
    int32_t err[4] = {0, 0, 0, 0};
    HIP_OK(h, hipMemcpyAsync(err, h->dev.error, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    if (err[0] == ERR_NONE) return 0;
    HIP_OK(h, hipMemsetAsync(h->dev.error, 0, sizeof err, h->stream));
    switch (err[0]) {
    case ERR_NAN_RATIO: return fail(h, "Got NaN log-unnormalized ratio (chain %d)", err[1]);
    case ERR_SLICE_SUPPORT: return fail(h, "SliceSampler supports contrained target, but the sampler should be initialized in the support (chain %d)", err[1]);
    case ERR_SLICE_INVALID_LP: return fail(h, "Got an invalid log density after updating state at index %d (chain %d)", err[2], err[1]);
    case ERR_SLICE_MAX_ITER: return fail(h, "Maximum number of iterations reached in slice_shrink! (chain %d, index %d)", err[1], err[2]);
    case ERR_AM_DENSITY: return fail(h, "AutoMALA can only be called on a configuration of positive density. (chain %d)", err[1]);
    case ERR_AM_STEP: return fail(h, "Could not find a positive step size (chain %d)", err[1]);
    case ERR_HANDSHAKE_TIMEOUT: return fail(h, "pte_run_scans: chain %d gave up waiting for its swap partner inside the one-launch scan loop (every workgroup had arrived: "
                                               "a wave died or was descheduled for seconds)", err[1]);
    default: return fail(h, "device error %d", err[0]);
    }
}

// h->timing: 0 off, 1 every kernel, 2 the explore kernels only (an event pair costs ~10 us of stream time per launch)
// on_launch: the bracket holds exactly ONE kernel launch, written with PTE_LAUNCH1 / DISPATCH_NLU*, which then carries the events
void time_begin(pte_engine *h, int kernel, bool on_launch = false) {
    h->ev_open = false; h->ev_ext = false; h->ev_ext_done = false;
    if (!h->timing || (h->timing == 2 && kernel != 0 && kernel != 4)) return;
    pte_engine::Ev ev; ev.kernel = kernel;
    if (h->ev_pool.size() >= 2) {
        ev.a = h->ev_pool.back(); h->ev_pool.pop_back();
        ev.b = h->ev_pool.back(); h->ev_pool.pop_back();
    } else { hipEventCreate(&ev.a); hipEventCreate(&ev.b); }
    if (!on_launch) hipEventRecord(ev.a, h->stream);
    h->events.push_back(ev);
    h->ev_open = true; h->ev_ext = on_launch;
}
void time_end(pte_engine *h) {
    if (!h->ev_open) return;
    if (h->ev_ext && !h->ev_ext_done) hipEventRecord(h->events.back().a, h->stream);      // (no launch took them: an empty bracket)
    if (!(h->ev_ext && h->ev_ext_done)) hipEventRecord(h->events.back().b, h->stream);
    h->ev_open = false; h->ev_ext = false;
}
void time_collect(pte_engine *h) {
    for (auto &ev : h->events) {
        float ms = 0.f;
        hipEventSynchronize(ev.b);
        hipEventElapsedTime(&ms, ev.a, ev.b);
        h->t_ms[ev.kernel] += ms; h->t_n[ev.kernel] += 1;
        if (h->t_samples[ev.kernel].size() < (size_t)1 << 20) h->t_samples[ev.kernel].push_back(ms);
        h->ev_pool.push_back(ev.a); h->ev_pool.push_back(ev.b);
    }
    h->events.clear();
}


// k_*_langevin_mw steer their waves' priorities by the replicas' pace where that can help: more workgroups than compute units (they share SIMDs) and
// all of them resident at once (four per compute unit) -- one per compute unit the counter costs 3-6 %, with a second generation the late starters
// look slow and the steering costs 3-5 % (profiles/r06_langevin_mw.txt)
static int langevin_mw_paced(pte_engine *h) {
    if (h->n_cus < 0) { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess) cus = 0; (void)hipGetLastError(); h->n_cus = cus; }
    return (h->n_cus > 0 && h->K > (int64_t)h->n_cus && h->K <= 4 * (int64_t)h->n_cus) ? 1 : 0;
}

// one launch of the Langevin-family kernel (pte_automala_params.hpp); like PTE_LAUNCH1, the open timing bracket's events ride on it
static int launch_langevin(pte_engine *h, int E, int target, bool slice, bool full, int64_t N, const AmParams &ap) {
    LangevinLaunch L{E, target, slice, full, (unsigned)N, h->stream, false, nullptr, nullptr};
    L.one_wave16 = (h->cfg.debug_kernel & PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE) != 0;       // (test build only: pte_create refuses the flag otherwise)
    if (h->ev_open && h->ev_ext && !h->ev_ext_done) { L.ext = true; L.ev_a = h->events.back().a; L.ev_b = h->events.back().b; h->ev_ext_done = true; }
    if (E == 16 && !slice && !L.one_wave16) HIP_OK(h, hipMemsetAsync(h->dev.pace, 0, sizeof(unsigned int), h->stream));      // k_explore_langevin_mw: its workgroups count their refreshes here
    if (langevin_launch(L, h->dev, ap)) return fail(h, "this build holds no Langevin-family kernels (PTE_DEV_NO_LANGEVIN)");
    return 0;
}

int launch_explorer_kind(pte_engine *h, int64_t scan, int kind);
int launch_explore(pte_engine *h, int64_t scan) {
    (void)scan;
    const int64_t N = h->K;
    if ((h->cfg.record_flags & PTE_RECORD_TRACES) && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "traces buffer full: %lld scans since the last pte_reduce (max_scans_per_round = %lld)",
                    (long long)h->scans_in_round, (long long)h->cfg.max_scans_per_round);
    // the explore kernels write row `scans_in_round` of the step-size-search log too (PTE_RECORD_REFERENCE_REDUCTION with AutoMALA): refuse BEFORE the
    // launch, not at the swap that follows it (the scan after max_scans_per_round wrote K * cap int16 past the end of am_log)
    if (h->dev.am_log && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "step-size-search log full: %lld scans since the last pte_reduce (max_scans_per_round = %lld)",
                    (long long)h->scans_in_round, (long long)h->cfg.max_scans_per_round);
    if (h->dev.eac_log && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "energy log full: %lld scans since the last pte_reduce (max_scans_per_round = %lld)",
                    (long long)h->scans_in_round, (long long)h->cfg.max_scans_per_round);
    h->dev.trace_idx = h->scans_in_round;
    if (h->dev.eac_log && h->cfg.explorer != PTE_EXPLORER_NONE) {            // the energy pairs of the scan, logged around its explorer kernel(s) (k_log_energy)
        hipLaunchKernelGGL(k_log_energy, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->dev, 0);
        int rc;
        if (h->cfg.explorer2 == PTE_EXPLORER_NONE) { h->dev.compose_phase = 0; rc = launch_explorer_kind(h, scan, h->cfg.explorer); }
        else {
            h->dev.compose_phase = 1; rc = launch_explorer_kind(h, scan, h->cfg.explorer);
            if (!rc) { h->dev.compose_phase = 2; rc = launch_explorer_kind(h, scan, h->cfg.explorer2); }
            h->dev.compose_phase = 0;
        }
        if (rc) return rc;
        hipLaunchKernelGGL(k_log_energy, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->dev, 1);
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    if (h->cfg.explorer2 == PTE_EXPLORER_NONE) { h->dev.compose_phase = 0; return launch_explorer_kind(h, scan, h->cfg.explorer); }
    // Compose(first, second), src/explorers/Compose.jl:16-19: two kernels back to back on the replica's stream
    h->dev.compose_phase = 1;
    int rc = launch_explorer_kind(h, scan, h->cfg.explorer);
    h->dev.compose_phase = 2;
    if (!rc) rc = launch_explorer_kind(h, scan, h->cfg.explorer2);
    h->dev.compose_phase = 0;
    return rc;
}

int launch_explorer_kind(pte_engine *h, int64_t scan, int kind) {
    const int64_t N = h->K;
    switch (kind) {
    case PTE_EXPLORER_NONE: return 0;
    case PTE_EXPLORER_TOY:
        time_begin(h, 0, true);
        DISPATCH_NLU(h->nlu, k_explore_toy, dim3((unsigned)((N + NRM_WPB - 1) / NRM_WPB)), dim3(64 * NRM_WPB), h->stream, h->dev);
        time_end(h);
        break;
    case PTE_EXPLORER_SLICE:
        if (h->cfg.target == PTE_TARGET_FUNNEL) {
            // SliceSampler on the interpolated path: the register-resident kernel of the Langevin family in its slice mode
            // (full log potential per evaluation, as the reference's slice_sample! does for any log_potential)
            AmParams ap{};
            ap.slice = 1; ap.slice_w = h->cfg.slice_w; ap.slice_p = h->cfg.slice_p; ap.slice_n_passes = h->cfg.slice_n_passes;
            ap.slice_max_iter = h->cfg.slice_max_iter;
            ap.ref_prec = h->cfg.target_params[0]; ap.log3 = std::log(3.0);
            const int E = h->d <= 64 ? 1 : h->d <= 128 ? 2 : h->d <= 256 ? 4 : h->d <= 512 ? 8 : 16;
            time_begin(h, 0, true);
            if (launch_langevin(h, E, TGT_FUNNEL, true, false, N, ap)) return 1;
            time_end(h);
            break;
        }
        {
        SliceParams sp{h->cfg.slice_w, h->cfg.slice_p, h->cfg.slice_n_passes, h->cfg.slice_max_iter};
        time_begin(h, 0, true);
        if (h->slice_impl == 1) {          // PTE_KERNEL_SLICE_SEQUENTIAL: the plain sequential kernel (exact fallback, bisecting)
            DISPATCH_NLU(h->nlu, k_explore_slice, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp);
        } else if (h->slice_impl == 8) {
            // shrinkage steps for every hypothesis: PTE_S8_BS = 9, re-measured after every change of the round's cost (tools/bench_variant.py)
            // beyond two replicas per SIMD (PTE_S8_TWIN_FROM = 2048: the 512-draw kernel holds 189 VGPRs) the 10 KB / 128-VGPR variant keeps 16 per CU
            const bool fast_ok = sp.p > PTE_S8_BD && sp.p <= 20 && sp.max_iter >= PTE_S8_BS;       // what the FAST instantiation assumes (pte_slice8.hpp)
            if (N <= PTE_S8_TWIN_FROM && fast_ok) { DISPATCH_NLU_M(h->nlu, k_explore_slice8, PTE_S8_BS, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp); }
            else if (N <= PTE_S8_TWIN_FROM) { DISPATCH_NLU_M(h->nlu, k_explore_slice8_generic, PTE_S8_BS, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp); }
            else { DISPATCH_NLU_M(h->nlu, k_explore_slice8_lds10k, PTE_S8_BS, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp); }
        }
#ifdef PTE_TEST_KERNELS
        else if (h->slice_impl == 7) {
            S7Tune tn{2, 8, 2};
            DISPATCH_NLU(h->nlu, k_explore_slice7, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp, tn);
        } else if (h->slice_impl == 5) {
            DISPATCH_NLU_M(h->nlu, k_explore_slice5, 4, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp);
        } else if (h->slice_impl == 2) {
            DISPATCH_NLU_M(h->nlu, k_explore_slice2, 4, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp);
        }
#endif
        else return fail(h, "SliceSampler kernel %d is not in this build", h->slice_impl);   // pte_create validates; unreachable
        time_end(h);
        break;
    }
    case PTE_EXPLORER_MALA:
    case PTE_EXPLORER_AUTOMALA: {
        AmParams ap{};
        ap.mala = (kind == PTE_EXPLORER_MALA) ? 1 : 0;
        ap.step_size = ap.mala ? h->cfg.am_step_size : h->step_size; ap.n_refresh = h->am_n_refresh; ap.precond = h->cfg.am_preconditioner;
        ap.p0 = h->cfg.am_p0; ap.p1 = h->cfg.am_p1;
        ap.target_std = h->have_target_std ? h->d_target_std : nullptr;
        ap.use_mh = (scan != 1) ? 1 : 0;                 // AutoMALA.jl:87,96-102
        ap.ref_prec = h->cfg.target_params[0]; ap.log3 = std::log(3.0);
        ap.pace = h->d > 512 ? langevin_mw_paced(h) : 0;
        const int E = h->d <= 64 ? 1 : h->d <= 128 ? 2 : h->d <= 256 ? 4 : h->d <= 512 ? 8 : 16;
        const bool fun = h->cfg.target == PTE_TARGET_FUNNEL;
        time_begin(h, 0, true);
        const bool full = h->d == 64 * (int64_t)E;       // no ragged last block: the instantiation without per-lane validity masks
        if (launch_langevin(h, E, fun ? TGT_FUNNEL : TGT_MVN, false, full, N, ap)) return 1;
        time_end(h);
        break;
    }
    case PTE_EXPLORER_ISING_METROPOLIS: {
        IsingParams ip{(int)std::llround(std::sqrt((double)h->d)), h->cfg.slice_n_passes, h->cfg.target_params[0]};
        time_begin(h, 0, true);
        // L % 32 == 0: lane-speculative bit-packed sweep; other lattice sizes: the scalar byte-lattice kernel
        if (ip.L % 32 == 0 && h->ising_impl == 0) {
            if (ip.L == 32) PTE_LAUNCH1(k_explore_ising_spec<true>, dim3((unsigned)N), dim3(64), (size_t)(h->d / 8 + 8), h->stream, h->dev, ip);
            else            PTE_LAUNCH1(k_explore_ising_spec<false>, dim3((unsigned)N), dim3(64), (size_t)(h->d / 8 + 8), h->stream, h->dev, ip);
        }
#ifdef PTE_TEST_KERNELS
        else if (ip.L % 32 == 0 && h->ising_impl == 1)
            PTE_LAUNCH1(k_explore_ising_bits, dim3((unsigned)N), dim3(64), (size_t)(h->d / 8), h->stream, h->dev, ip);
#endif
        else
            PTE_LAUNCH1(k_explore_ising, dim3((unsigned)N), dim3(64), (size_t)h->d, h->stream, h->dev, ip);
        time_end(h);
        break;
    }
    default: return fail(h, "explorer %d is not implemented on the device", h->cfg.explorer);
    }
    HIP_OK(h, hipGetLastError());
    return 0;
}

int launch_swap(pte_engine *h, int64_t scan) {
    const int64_t N = h->K;
    if ((h->cfg.record_flags & PTE_RECORD_INDEX_PROCESS) && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "index_process buffer full: %lld scans since the last pte_reduce (max_scans_per_round = %lld)",
                    (long long)h->scans_in_round, (long long)h->cfg.max_scans_per_round);
    if (h->world != 1) return fail(h, "pte_swap / pte_run_scans need world_size == 1; sharded engines use pte_swap_begin / pte_swap_finish");
    const int even = (scan % 2 == 0) ? 1 : 0;          // create_swap_graph(::DEO), DEO.jl:12
    const unsigned block = 256, grid = (unsigned)((N + block - 1) / block);
    time_begin(h, 1, true);
    PTE_LAUNCH1(k_swap, dim3(grid), dim3(block), 0, h->stream, h->dev, even, h->scans_in_round);
    time_end(h);
    HIP_OK(h, hipGetLastError());
    h->scans_in_round += 1;
    return 0;
}

// boundary pair activity of this shard on the given graph: side 0 = (c0-1, c0), side 1 = (c0+K-1, c0+K)
void boundary_active(const pte_engine *h, int even, int32_t active[2]) {
    auto partner = [&](int64_t c) {
        const bool chain_even = ((c + 1) % 2 == 0);
        int64_t proposed = (c + 1) + ((chain_even == (even != 0)) ? 1 : -1);
        return (proposed == 0) ? (int64_t)0 : (proposed == h->N + 1 ? h->N - 1 : proposed - 1);
    };
    active[0] = (h->c0 > 0 && partner(h->c0) == h->c0 - 1) ? 1 : 0;
    active[1] = (h->c0 + h->K < h->N && partner(h->c0 + h->K - 1) == h->c0 + h->K) ? 1 : 0;
}

int run_scans_sharded(pte_engine *h, int64_t first_scan, int64_t n_scans);

// ---- one launch per pte_run_scans (k_scans_*; pte_kernels.hpp "ScanLoop") ----------------------------------------------------------------
// Which engines: one GPU holds the whole ladder (world_size == 1), SliceSampler on the scaled-precision MVN path with the default kernel
// generation, no Compose, every workgroup of the call resident at once (the hand-shakes spin: a workgroup that waits for a GPU slot
// would be waited for) -- asked of the runtime for the very instantiation that is launched -- and the shapes at which it was measured to win.
#ifdef PTE_DEV_FEW_NLU
#define OCC_NLU_MB(nlu, KERNEL, MM, BLOCK, out)                                                                        \
    switch (nlu) {                                                                                              \
    case 4: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<4, MM>, BLOCK, 0); break;                   \
    default: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<6, MM>, BLOCK, 0); break;                  \
    }
#else
#define OCC_NLU_MB(nlu, KERNEL, MM, BLOCK, out)                                                                        \
    switch (nlu) {                                                                                              \
    case 0: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<0, MM>, BLOCK, 0); break;                   \
    case 1: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<1, MM>, BLOCK, 0); break;                   \
    case 2: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<2, MM>, BLOCK, 0); break;                   \
    case 3: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<3, MM>, BLOCK, 0); break;                   \
    case 4: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<4, MM>, BLOCK, 0); break;                   \
    case 5: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<5, MM>, BLOCK, 0); break;                   \
    default: hipOccupancyMaxActiveBlocksPerMultiprocessor(&out, KERNEL<6, MM>, BLOCK, 0); break;                  \
    }
#endif

// 0: k_scans_slice8, 1: k_scans_slice8_generic -- the same choice launch_explorer_kind makes per scan
int fused_slice_variant(const pte_engine *h) {
    const bool fast_ok = h->cfg.slice_p > PTE_S8_BD && h->cfg.slice_p <= 20 && h->cfg.slice_max_iter >= PTE_S8_BS;
    return fast_ok ? 0 : 1;
}

// SliceSampler, measured on one box against the launch-per-scan loop (tools/r05_fused_shapes.sh, profiles/r05_fused_shapes.txt): at most ONE wave per SIMD
// and rows of at most 16 KB -- 1024 chains: d = 512 x1.055, d = 1024 x1.03, d = 2048 x1.00, d = 4096 x0.99 (every release writes back the
// XCD's dirty L2 lines, and 128 waves x 32 KB of freshly written rows are all of it); 2048 chains at d = 1024 x0.91 (a wave that polls shares
// its SIMD with a wave that works, and the fences cost per resident workgroup).  Elsewhere the loop of rounds 1-4 stays.
#define OCC_NLU_M(nlu, KERNEL, MM, out) OCC_NLU_MB(nlu, KERNEL, MM, 64, out)
// The shapes the one-kernel loop was MEASURED to win at, as build-time constants with the profile that produced them (re-measure before moving them):
#ifndef PTE_FUSED_SLICE_MAX_D
#define PTE_FUSED_SLICE_MAX_D 2048       // profiles/r05_fused_shapes.txt: 1024 chains, d = 512 x1.055, 1024 x1.03, 2048 x1.00, 4096 x0.99 against two launches per scan
#endif
#ifndef PTE_FUSED_LANGEVIN_MAX_D
#define PTE_FUSED_LANGEVIN_MAX_D 1024    // d <= 512: one wave per chain (k_scans_automala[_wg]; profiles/r05_am_wg_ab.txt); 512 < d <= 1024, MVN path: four waves per chain (k_scans_langevin_mw; profiles/r06_langevin_mw.txt)
#endif
// 0: not a fused kind; 1: SliceSampler on the MVN path (k_scans_slice8*); 2: AutoMALA / MALA on the MVN or funnel path (k_scans_automala)
int fused_kind(const pte_engine *h) {
    if (h->cfg.explorer2 != PTE_EXPLORER_NONE) return 0;
    if (h->cfg.explorer == PTE_EXPLORER_SLICE && h->cfg.target == PTE_TARGET_MVN_SCALED_PRECISION && h->slice_impl == 8) return h->d <= PTE_FUSED_SLICE_MAX_D ? 1 : 0;
    if ((h->cfg.explorer == PTE_EXPLORER_AUTOMALA || h->cfg.explorer == PTE_EXPLORER_MALA) &&
        (h->cfg.target == PTE_TARGET_MVN_SCALED_PRECISION || h->cfg.target == PTE_TARGET_FUNNEL)) {
        if (h->d > 512) {      // four waves per chain: the scaled-precision MVN path only (the funnel's loop measured no gain: pte_langevin_launch.hpp)
#ifndef PTE_DEV_MW_FUNNEL_LOOP
            if (h->cfg.target != PTE_TARGET_MVN_SCALED_PRECISION) return 0;
#endif
            if (h->cfg.debug_kernel & PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE) return 0;                 // (test build: the one-wave kernel with sixteen blocks per lane has no loop form)
        }
        return h->d <= PTE_FUSED_LANGEVIN_MAX_D ? 2 : 0;
    }
    return 0;
}
int langevin_E(const pte_engine *h) { return h->d <= 64 ? 1 : h->d <= 128 ? 2 : h->d <= 256 ? 4 : h->d <= 512 ? 8 : 16; }

bool fused_scans_eligible(pte_engine *h, int64_t n_scans) {
    if (!h->fused_allowed || h->world != 1 || n_scans < 1) return false;
    const int kind = fused_kind(h);
    if (kind == 0) return false;
    if (h->dev.eac_log) return false;          // the energy pairs are logged by launches around the explorer kernels (k_log_energy)
    // a recorder buffer that would overflow inside the call: the launch-per-scan loop reports it at the scan that overflows, as before
    if (((h->cfg.record_flags & (PTE_RECORD_TRACES | PTE_RECORD_INDEX_PROCESS | PTE_RECORD_REFERENCE_REDUCTION)) || h->dev.am_log || h->dev.swap_log) &&
        h->scans_in_round + n_scans > h->cfg.max_scans_per_round) return false;
    if (h->fused_limit < 0) {
        int per_cu = 0, cus = 0;
        if (kind == 1) {
            if (fused_slice_variant(h) == 0) { OCC_NLU_M(h->nlu, k_scans_slice8, PTE_S8_BS, per_cu); }
            else { OCC_NLU_M(h->nlu, k_scans_slice8_generic, PTE_S8_BS, per_cu); }
        } else {
            per_cu = langevin_scan_loop_blocks_per_cu(langevin_E(h), h->cfg.target == PTE_TARGET_FUNNEL ? TGT_FUNNEL : TGT_MVN, h->d == 64 * (int64_t)langevin_E(h));
        }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess) cus = 0;
        (void)hipGetLastError();
        // every workgroup resident (the hand-shakes spin) AND at most one wave per SIMD (4 SIMDs per CU)
        h->fused_limit = (int64_t)std::min(per_cu, 4) * (int64_t)cus;
        // the form with several consecutive chains per workgroup (intra-workgroup pairs shake hands through LDS), where the build has it and
        // the chains fit with one such workgroup per compute unit
        h->fused_wg = 1;
        if (kind == 2 && h->fused_wg_allowed) {
            const int wgn = langevin_scan_wg();
            const int per_cu_wg = wgn > 1 && wgn <= 4 ? langevin_scan_loop_blocks_per_cu(langevin_E(h), h->cfg.target == PTE_TARGET_FUNNEL ? TGT_FUNNEL : TGT_MVN, h->d == 64 * (int64_t)langevin_E(h), wgn) : 0;
            (void)hipGetLastError();
            if (per_cu_wg >= 1 && h->K <= (int64_t)wgn * cus) { h->fused_wg = wgn; h->fused_limit = std::max(h->fused_limit, (int64_t)wgn * cus); }
        }
        if (h->fused_limit > 0 && h->K <= h->fused_limit && !h->hs_flag) {
            if (dev_alloc(h, &h->hs_flag, (size_t)h->K) || dev_alloc(h, &h->hs_pub, (size_t)h->K * 8) || dev_alloc(h, &h->hs_gate, 2)) { h->fused_limit = 0; h->hs_flag = nullptr; h->err.clear(); }
            else hipStreamSynchronize(h->stream);
        }
    }
    return h->K <= h->fused_limit && h->hs_flag != nullptr;
}

// the loop of rounds 1-4: explore and swap launched per scan
int run_scans_two_launches(pte_engine *h, int64_t first_scan, int64_t n_scans) {
    for (int64_t s = first_scan; s < first_scan + n_scans; ++s) {
        if (launch_explore(h, s)) return 1;
        if (launch_swap(h, s)) return 1;
    }
    int rc = check_device_error(h);
    time_collect(h);
    return rc;
}

// One launch for all the scans of the call.  The launch starts with the residency gate (pte_kernels.hpp scan_loop_gate): if a workgroup is
// missing after 50 ms -- another engine, stream or process holds compute units -- every workgroup returns with NOTHING written, and the same
// scans run here as explore + swap launches: the call cannot hang and its result is the same either way (the reference's loop,
// src/pt/pigeons.jl:46-55, has no failure mode of this kind; neither has pte_run_scans).  After an abort the next calls skip the attempt (1, 2, 4, ...
// 256 calls: the device is evidently shared), pte_scan_loop_stats counts both.  What the gate does not cover: a wave that dies or is
// descheduled for more than 3 s AFTER every workgroup has arrived -- then the hand-shake times out, every other wave sees the error word and
// leaves, the call fails and the engine is poisoned (chains stopped at different scans) until pte_set_state.
int run_scans_fused(pte_engine *h, int64_t first_scan, int64_t n_scans) {
    const int64_t N = h->K;
    const int kind = fused_kind(h);
    const unsigned long long wgs = (unsigned long long)((kind == 2 && h->fused_wg > 1) ? (N + h->fused_wg - 1) / h->fused_wg : N);
    h->gate_seq += 1;
    ScanLoop sl{first_scan, n_scans, h->scans_in_round, h->hs_epoch, h->hs_flag, h->hs_pub, h->hs_gate, h->gate_seq, h->gate_arrived + wgs, h->test_fault};
    h->gate_arrived += wgs;                  // (whatever the gate decides: every workgroup of the launch arrives before the kernel ends)
    h->dev.compose_phase = 0; h->dev.trace_idx = h->scans_in_round;
    time_begin(h, 4, true);
    const bool timed = h->ev_open;
    if (kind == 1) {
        SliceParams sp{h->cfg.slice_w, h->cfg.slice_p, h->cfg.slice_n_passes, h->cfg.slice_max_iter};
        if (fused_slice_variant(h) == 0) { DISPATCH_NLU_M(h->nlu, k_scans_slice8, PTE_S8_BS, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp, sl); }
        else { DISPATCH_NLU_M(h->nlu, k_scans_slice8_generic, PTE_S8_BS, dim3((unsigned)N), dim3(64), h->stream, h->dev, sp, sl); }
    } else {
        // the parameters launch_explorer_kind gives the per-scan kernel; use_mh (scan != 1, AutoMALA.jl:87,96-102) is decided per scan inside
        AmParams ap{};
        ap.mala = (h->cfg.explorer == PTE_EXPLORER_MALA) ? 1 : 0;
        ap.step_size = ap.mala ? h->cfg.am_step_size : h->step_size; ap.n_refresh = h->am_n_refresh; ap.precond = h->cfg.am_preconditioner;
        ap.p0 = h->cfg.am_p0; ap.p1 = h->cfg.am_p1;
        ap.target_std = h->have_target_std ? h->d_target_std : nullptr;
        ap.use_mh = 1;
        ap.ref_prec = h->cfg.target_params[0]; ap.log3 = std::log(3.0);
        ap.pace = h->d > 512 ? langevin_mw_paced(h) : 0;
        const int E = langevin_E(h);
        LangevinLaunch L{E, h->cfg.target == PTE_TARGET_FUNNEL ? TGT_FUNNEL : TGT_MVN, false, h->d == 64 * (int64_t)E, (unsigned)N, h->stream, false, nullptr, nullptr, &sl, h->fused_wg};
        if (h->ev_open && h->ev_ext && !h->ev_ext_done) { L.ext = true; L.ev_a = h->events.back().a; L.ev_b = h->events.back().b; h->ev_ext_done = true; }
        if (E == 16) HIP_OK(h, hipMemsetAsync(h->dev.pace, 0, sizeof(unsigned int), h->stream));      // k_scans_langevin_mw: its workgroups count their refreshes here
        if (langevin_launch(L, h->dev, ap)) { time_end(h); return fail(h, "this build holds no fused Langevin-family kernel"); }
    }
    time_end(h);
    HIP_OK(h, hipGetLastError());
    unsigned long long gate[2] = {0, 0};
    HIP_OK(h, hipMemcpyAsync(gate, h->hs_gate, sizeof gate, hipMemcpyDeviceToHost, h->stream));
    int rc = check_device_error(h);          // synchronises the stream: `gate` has landed too
    if (gate[1] == ((h->gate_seq << 1) | 1ull)) {
        // not every workgroup was resident within the bound: nothing has run.  The aborted launch is no sample of the scan loop's timing.
        if (timed && !h->events.empty() && h->events.back().kernel == 4) {
            h->ev_pool.push_back(h->events.back().a); h->ev_pool.push_back(h->events.back().b); h->events.pop_back();
        }
        h->gate_aborts += 1;
        h->fused_backoff = h->fused_backoff ? std::min<int64_t>(256, 2 * h->fused_backoff) : 1;
        h->fused_skip = h->fused_backoff;
        if (rc) return rc;                   // (cannot happen: no workgroup has touched the error word)
        return run_scans_two_launches(h, first_scan, n_scans);
    }
    if (gate[1] != (h->gate_seq << 1) || gate[0] != h->gate_arrived) {
        h->poisoned = true;
        return fail(h, "pte_run_scans: the scan loop's residency gate is inconsistent (decision %llu, arrivals %llu; expected %llu, %llu)",
                    gate[1], gate[0], h->gate_seq << 1, h->gate_arrived);
    }
    h->fused_calls += 1; h->fused_backoff = 0;
    if (timed) h->t_scans4 += n_scans;
    h->hs_epoch += (unsigned long long)n_scans;
    h->scans_in_round += n_scans;
    time_collect(h);
    if (rc) {
        // a wave that hits an error leaves the loop, and so does every wave that then waits for it: the chains stand at different scans
        h->poisoned = true;
        h->err += "; the engine's replicas stopped at different scans and its recorders are void: only pte_set_state (state, chain and rng of every replica; "
                  "it also discards the round's recorders) or pte_destroy are accepted now";
    }
    return rc;
}

int poisoned_error(pte_engine *h, const char *what) {
    return fail(h, "%s: this engine is poisoned -- an earlier pte_run_scans failed inside the one-launch scan loop, its replicas stopped at different scans; "
                   "restore it with pte_set_state(state, chain, rng) or destroy it", what);
}
#define PTE_ALIVE(h, what) do { if ((h)->poisoned) return poisoned_error(h, what); } while (0)

}  // namespace

extern "C" {

int pte_default_config(pte_config *c) {
    if (!c) return 1;
    std::memset(c, 0, sizeof *c);
    c->struct_size = sizeof(pte_config);
    c->abi_version = PTE_ABI_VERSION;
    c->device = 0;
    c->target = PTE_TARGET_MVN_SCALED_PRECISION;
    c->explorer = PTE_EXPLORER_TOY;
    c->record_flags = PTE_RECORD_ROUND_TRIP | PTE_RECORD_INDEX_PROCESS;
    c->n_chains = 10; c->dim = 2; c->seed = 1;
    c->max_scans_per_round = 1024;
    c->target_params[0] = 1.0; c->target_params[1] = 10.0;
    c->slice_w = 10.0; c->slice_p = 20; c->slice_n_passes = 3; c->slice_max_iter = 1024;
    c->am_base_n_refresh = 3; c->am_exponent_n_refresh = 0.35; c->am_step_size = 1.0;
    c->am_p0 = 1.0 / 3.0; c->am_p1 = 1.0 / 3.0; c->am_preconditioner = 2;
    c->rank = 0; c->world_size = 1;
    return 0;
}

const char *pte_last_error(const pte_engine *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pte_create(const pte_config *cfg, pte_engine **out) {
    if (!cfg || !out) return fail(nullptr, "pte_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(pte_config) || cfg->abi_version != PTE_ABI_VERSION)
        return fail(nullptr, "pte_create: ABI mismatch (struct_size %u vs %zu, version %u vs %d)",
                    cfg->struct_size, sizeof(pte_config), cfg->abi_version, PTE_ABI_VERSION);
    if (cfg->n_chains < 1) return fail(nullptr, "pte_create: n_chains must be >= 1");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail(nullptr, "pte_create: bad rank / world_size");
    if (cfg->n_chains_variational < 0) return fail(nullptr, "pte_create: n_chains_variational must be >= 0");
    if (cfg->n_chains_variational > 0 && cfg->world_size != 1) return fail(nullptr, "pte_create: two-leg tempering (n_chains_variational > 0) runs on a single engine");
    if (cfg->n_chains % cfg->world_size != 0) return fail(nullptr, "pte_create: n_chains (%lld) must be a multiple of world_size (%d)", (long long)cfg->n_chains, cfg->world_size);
    const bool swapper = cfg->target == PTE_TARGET_TEST_SWAPPER;
    const bool funnel = cfg->target == PTE_TARGET_FUNNEL;
    const bool ising = cfg->target == PTE_TARGET_ISING;
    if (ising) {
        const int64_t L = (int64_t)std::llround(std::sqrt((double)cfg->dim));
        if (L < 2 || L * L != cfg->dim || cfg->dim > 65536) return fail(nullptr, "pte_create: Ising needs dim = base_length^2 <= 65536");
        if (cfg->explorer == PTE_EXPLORER_SLICE || cfg->explorer2 == PTE_EXPLORER_SLICE)   // spins are Bool coordinates: SliceSampler.jl:65-86 (and :136-142, :189 for Integer ones)
            return fail(nullptr, "pte_create: SliceSampler's Bool / Integer coordinate methods are not available on the device (its Float64 methods are); "
                                 "the Ising path is explored by IsingMetropolis only -- use the reference CPU path for Bool / Integer states");
        if (cfg->explorer != PTE_EXPLORER_ISING_METROPOLIS) return fail(nullptr, "pte_create: the Ising path is explored by IsingMetropolis only");
    } else if (cfg->explorer == PTE_EXPLORER_ISING_METROPOLIS) return fail(nullptr, "pte_create: IsingMetropolis needs the Ising target");
    if (!swapper && !funnel && !ising && cfg->target != PTE_TARGET_MVN_SCALED_PRECISION)
        return fail(nullptr, "pte_create: target %d has no device log-potential; use the reference CPU path", cfg->target);
    auto grad_based = [](int k) { return k == PTE_EXPLORER_AUTOMALA || k == PTE_EXPLORER_MALA; };
    const bool uses_grad = grad_based(cfg->explorer) || grad_based(cfg->explorer2);
    auto on_path = [&](int k) { return grad_based(k) || k == PTE_EXPLORER_SLICE; };
    if (funnel && !(on_path(cfg->explorer) && (cfg->explorer2 == PTE_EXPLORER_NONE || on_path(cfg->explorer2))))
        return fail(nullptr, "pte_create: the funnel path is implemented for AutoMALA / MALA / SliceSampler (and Compose of them); use the reference CPU path");
    if ((uses_grad || funnel) && (cfg->dim < 1 || cfg->dim > 1024))
        return fail(nullptr, "pte_create: AutoMALA / MALA (and every explorer of the funnel path) keep the replica in registers, dim must be in 1..1024 (got %lld)", (long long)cfg->dim);
    if (funnel && (cfg->debug_kernel & ~(PTE_KERNEL_FLAG_BITS | PTE_KERNEL_TEST_BITS)) != 0)
        return fail(nullptr, "pte_create: debug_kernel %d is not available on the funnel path (one register-resident kernel serves it)", cfg->debug_kernel);
    if (cfg->explorer2 != PTE_EXPLORER_NONE) {           // Compose(first, second)
        auto composable = [&](int k) { return k == PTE_EXPLORER_SLICE || grad_based(k); };
        if (!composable(cfg->explorer) || !composable(cfg->explorer2))
            return fail(nullptr, "pte_create: Compose is available for SliceSampler / AutoMALA / MALA (got %d, %d)", cfg->explorer, cfg->explorer2);
    }
    if (!swapper && !ising && (cfg->dim < 1 || cfg->dim > 4096))
        return fail(nullptr, "pte_create: dim must be in 1..4096 (got %lld)", (long long)cfg->dim);
    if (swapper && cfg->explorer != PTE_EXPLORER_NONE)
        return fail(nullptr, "pte_create: TestSwapper has no explorer");
    if (!swapper && !ising && cfg->explorer != PTE_EXPLORER_TOY && cfg->explorer != PTE_EXPLORER_SLICE && !grad_based(cfg->explorer))
        return fail(nullptr, "pte_create: explorer %d is not implemented on the device", cfg->explorer);
    if ((cfg->record_flags & PTE_RECORD_TRACES_EXTENDED) && !(cfg->record_flags & PTE_RECORD_TRACES))
        return fail(nullptr, "pte_create: PTE_RECORD_TRACES_EXTENDED needs PTE_RECORD_TRACES");
    if (cfg->record_flags & PTE_RECORD_REFERENCE_REDUCTION) {
        if (!(cfg->record_flags & PTE_RECORD_INDEX_PROCESS))
            return fail(nullptr, "pte_create: PTE_RECORD_REFERENCE_REDUCTION needs PTE_RECORD_INDEX_PROCESS (the replay asks which replica held the lower chain of a pair)");
        if ((double)cfg->max_scans_per_round * (double)(cfg->n_chains + cfg->n_chains_variational) * 16.0 > 64e9)
            return fail(nullptr, "pte_create: the swap log (max_scans_per_round x chains x 2 doubles) would exceed 64 GB");
    }
#ifndef PTE_TEST_KERNELS
    if (cfg->debug_kernel & PTE_KERNEL_TEST_BITS)
        return fail(nullptr, "pte_create: debug_kernel 0x%x carries a fault-injection flag (PTE_KERNEL_TEST_*); those exist in the test build libpte_test.so only", cfg->debug_kernel);
#endif
    {   // debug_kernel: 0 = the default kernel of the explorer; anything else must exist in THIS build (no silent fall-through)
        const int dk = cfg->debug_kernel & ~(PTE_KERNEL_FLAG_BITS | PTE_KERNEL_TEST_BITS);      // (the flag bit chooses the scan loop's form, not the kernel generation)
        const bool slice = cfg->explorer == PTE_EXPLORER_SLICE || cfg->explorer2 == PTE_EXPLORER_SLICE;
        bool ok = dk == 0 || (slice && dk == PTE_KERNEL_SLICE_SEQUENTIAL) || (ising && dk == PTE_KERNEL_ISING_BYTES);
#ifdef PTE_TEST_KERNELS
        ok = ok || (slice && (dk == 2 || dk == 5 || dk == 7 || dk == 8)) || (ising && dk == PTE_KERNEL_ISING_BITS);
#endif
        if (!ok) return fail(nullptr, "pte_create: debug_kernel %d is not available for this explorer in this build of libpte "
                                      "(0 = default, %d = sequential SliceSampler kernel; the other generations live in the test build libpte_test.so)",
                             dk, PTE_KERNEL_SLICE_SEQUENTIAL);
    }
    if ((cfg->record_flags & PTE_RECORD_TRACES) &&
        (double)cfg->max_scans_per_round * (double)((cfg->record_flags & PTE_RECORD_TRACES_EXTENDED) ? cfg->n_chains / cfg->world_size : 1) * (double)(cfg->dim + 1) * 8.0 > 64e9)
        return fail(nullptr, "pte_create: the traces buffer (max_scans_per_round x chains x (dim+1) doubles) would exceed 64 GB");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, "pte_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "pte_create: bad device ordinal %d", cfg->device);

    pte_engine *h = new pte_engine();
    h->cfg = *cfg;
    const int64_t N = h->N = cfg->n_chains + cfg->n_chains_variational;      // Inputs.jl:128
    const int64_t d = h->d = swapper ? 0 : cfg->dim;
    h->world = cfg->world_size; h->rank = cfg->rank;
    const int64_t K = h->K = N / cfg->world_size;
    h->c0 = K * cfg->rank;
    auto bail = [&](int) { g_create_error = h->err; pte_destroy(h); return 1; };
    if (hipSetDevice(cfg->device) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(1); }
    if (hipStreamCreate(&h->stream) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(1); }
    const int64_t B = (d + 63) / 64;
    h->nlu = next_pow2_log(B > 0 ? B : 1);
    // pte_config.debug_kernel (validated above): which kernel generation explores; never read from the environment
    const int dk_kernel = cfg->debug_kernel & ~(PTE_KERNEL_FLAG_BITS | PTE_KERNEL_TEST_BITS);
    h->fused_allowed = (cfg->debug_kernel & PTE_KERNEL_TWO_LAUNCHES) == 0;
    h->fused_wg_allowed = (cfg->debug_kernel & PTE_KERNEL_SCAN_LOOP_ONE_CHAIN) == 0;
    h->test_fault = (cfg->debug_kernel & PTE_KERNEL_TEST_DEAD_CHAIN) ? 1 : ((cfg->debug_kernel & PTE_KERNEL_TEST_LATE_WORKGROUP) ? 2 : 0);
    if (cfg->explorer == PTE_EXPLORER_SLICE || cfg->explorer2 == PTE_EXPLORER_SLICE) h->slice_impl = dk_kernel == 0 ? 8 : dk_kernel;
    if (cfg->explorer == PTE_EXPLORER_ISING_METROPOLIS) h->ising_impl = dk_kernel == PTE_KERNEL_ISING_BITS ? 1 : (dk_kernel == PTE_KERNEL_ISING_BYTES ? 2 : 0);
    EngineDev &e = h->dev;
    e.N = N; e.K = K; e.c0 = h->c0; e.d = d; e.ld = (d + 1) & ~(int64_t)1;
    e.sw = d;
    if (ising) {            // spins bit-packed in HBM (examples/ising.jl:18-22 keeps a BitMatrix): ceil(d/32) words, 8 KiB at L = 256
        const int64_t lw = (d + 31) / 32;
        e.ld = (lw + 1) / 2; e.sw = e.ld;
    }
    e.record_flags = cfg->record_flags; e.target = cfg->target; e.test_swapper_pr = cfg->target_params[0];
    const int64_t dd = d > 0 ? d : 1;
    int rc = 0;
    rc |= dev_alloc(h, &e.x, (size_t)(K * (e.ld > 0 ? e.ld : 1)));
    rc |= dev_alloc(h, &e.rng, (size_t)(2 * K));
    rc |= dev_alloc(h, &e.chain_of_slot, (size_t)K);
    rc |= dev_alloc(h, &h->slot_map[0], (size_t)K);
    rc |= dev_alloc(h, &h->slot_map[1], (size_t)K);
    rc |= dev_alloc(h, &e.replica_id, (size_t)K);
    rc |= dev_alloc(h, &e.stat, (size_t)(2 * K));
    rc |= dev_alloc(h, &e.nbr_stat, 4);
    rc |= dev_alloc(h, &e.bflag, 2);
    rc |= dev_alloc(h, &h->d_payload, (size_t)((e.sw > 0 ? e.sw : 1) + 8));
    rc |= dev_alloc(h, &h->d_napplied, 2);
    rc |= dev_alloc(h, &e.suff, (size_t)K);
    rc |= dev_alloc(h, &h->d_nhp, (size_t)N);
    rc |= dev_alloc(h, &h->d_sd, (size_t)N);
    rc |= dev_alloc(h, &h->d_nprec, (size_t)N);
    rc |= dev_alloc(h, &h->d_beta, (size_t)N);
    rc |= dev_alloc(h, &h->d_target_std, (size_t)dd);
    rc |= dev_alloc(h, &e.suff2, (size_t)K);
    rc |= dev_alloc(h, &e.suff3, (size_t)K);
    rc |= dev_alloc(h, &h->d_vref, (size_t)(5 * (d > 0 ? d : 1)));
    rc |= dev_alloc(h, &h->d_vuse, (size_t)N);
    rc |= dev_alloc(h, &e.am_fac_sum, (size_t)K); rc |= dev_alloc(h, &e.am_fac_n, (size_t)K);
    rc |= dev_alloc(h, &e.am_rev_sum, (size_t)K); rc |= dev_alloc(h, &e.am_rev_n, (size_t)K);
    rc |= dev_alloc(h, &e.swap_sum, (size_t)K);  rc |= dev_alloc(h, &e.swap_n, (size_t)K);
    rc |= dev_alloc(h, &e.lsr_up, (size_t)K);    rc |= dev_alloc(h, &e.lsr_dn, (size_t)K);
    rc |= dev_alloc(h, &e.lsr_n, (size_t)K);
    rc |= dev_alloc(h, &e.rt_state, (size_t)K);   rc |= dev_alloc(h, &e.rt_restarts, (size_t)K);
    rc |= dev_alloc(h, &e.rt_trips, (size_t)K);
    rc |= dev_alloc(h, &e.expl_acc_sum, (size_t)K);   rc |= dev_alloc(h, &e.expl_acc_n, (size_t)K);
    rc |= dev_alloc(h, &e.expl_steps_sum, (size_t)K); rc |= dev_alloc(h, &e.expl_steps_n, (size_t)K);
    rc |= dev_alloc(h, &e.on_mean, (size_t)(2 * (d + 1)));   rc |= dev_alloc(h, &e.on_m2, (size_t)(2 * (d + 1)) + PTE_WAVE_PROFILE_WORDS * (size_t)K);   // (+ 4 words per wave in -DPTE_PROFILE_WAVES builds)
    rc |= dev_alloc(h, &e.eac, (size_t)(5 * K)); rc |= dev_alloc(h, &e.eac_n, (size_t)K);
    rc |= dev_alloc(h, &e.lp_stash, (size_t)K);
    const int64_t trace_rows = (cfg->record_flags & PTE_RECORD_TRACES_EXTENDED) ? K : (cfg->n_chains_variational > 0 ? 2 : 1);   // chains traced per scan
    rc |= dev_alloc(h, &e.traces, (cfg->record_flags & PTE_RECORD_TRACES) ? (size_t)(cfg->max_scans_per_round * trace_rows * (d + 1)) : 1, false);
    rc |= dev_alloc(h, &e.on_n, 2);
    const int64_t ipcap = (cfg->record_flags & PTE_RECORD_INDEX_PROCESS) ? cfg->max_scans_per_round * K : 1;
    rc |= dev_alloc(h, &e.index_process, (size_t)ipcap, false);
    rc |= dev_alloc(h, &e.ip_replica, (size_t)ipcap, false);
    rc |= dev_alloc(h, &e.error, 4);
    rc |= dev_alloc(h, &e.pace, 2);
    e.mw_gk = nullptr;
    if (funnel && uses_grad && d > 512) rc |= dev_alloc(h, &e.mw_gk, (size_t)K * 1024, false);
    e.swap_log = nullptr; e.eac_log = nullptr;
    if (cfg->record_flags & PTE_RECORD_REFERENCE_REDUCTION) rc |= dev_alloc(h, &e.swap_log, (size_t)(cfg->max_scans_per_round * K * 2), false);
    if ((cfg->record_flags & PTE_RECORD_REFERENCE_REDUCTION) && (cfg->record_flags & PTE_RECORD_ENERGY_AC1)) rc |= dev_alloc(h, &e.eac_log, (size_t)(cfg->max_scans_per_round * K * 2), false);
    e.am_log = nullptr; e.am_log_cap = 0;
    if ((cfg->record_flags & PTE_RECORD_REFERENCE_REDUCTION) && (cfg->explorer == PTE_EXPLORER_AUTOMALA || cfg->explorer2 == PTE_EXPLORER_AUTOMALA)) {
        // AutoMALA searches a step size twice per refresh at most (forward, and backward for the reversibility check): am_factors' fits
        const int nref = cfg->am_base_n_refresh * (int)std::ceil(std::pow((double)(d > 0 ? d : 1), cfg->am_exponent_n_refresh));
        e.am_log_cap = 2 * nref;
        rc |= dev_alloc(h, &e.am_log, (size_t)(cfg->max_scans_per_round * K * e.am_log_cap), false);
    }
    if (rc) return bail(1);
    e.nhp = h->d_nhp; e.sd = h->d_sd; e.nprec = h->d_nprec; e.beta = h->d_beta;
    e.ref_nhp = -0.5 * cfg->target_params[0];
    e.ising_beta = cfg->target_params[0];
    h->step_size = cfg->am_step_size;
    // n_refresh = base_n_refresh * ceil(Int, dim^exponent_n_refresh)  (AutoMALA.jl:120)
    h->am_n_refresh = cfg->am_base_n_refresh * (int)std::ceil(std::pow((double)(d > 0 ? d : 1), cfg->am_exponent_n_refresh));
    if (uses_grad && cfg->am_preconditioner != 0) e.record_flags |= PTE_RECORD_ONLINE;   // _transformed_online (GradientBasedSampler.jl:19-25)
    e.slot_of_chain = h->slot_map[0]; e.slot_of_chain_alt = h->slot_map[1]; h->slot_cur = 0;

    // equally_spaced_schedule (reference src/schedules/Schedule.jl:36-44)
    h->betas.resize(N);
    auto equally_spaced = [](int64_t n, int64_t i) { return n == 1 ? 1.0 : ((i == n - 1) ? 1.0 : (double)i / (double)(n - 1)); };
    if (cfg->n_chains_variational > 0) {       // StabilizedPT(inputs): both legs equally spaced, the fixed leg reversed
        const int64_t nv = cfg->n_chains_variational, nf = cfg->n_chains;
        for (int64_t i = 0; i < nv; ++i) h->betas[i] = equally_spaced(nv, i);
        for (int64_t i = 0; i < nf; ++i) h->betas[nv + i] = equally_spaced(nf, nf - 1 - i);
        e.ref2 = N - 1; e.tgt_a = nv - 1; e.tgt_b = nv; e.rt_tgt_a = nf - 1; e.rt_tgt_b = nf;
    } else {
        for (int64_t i = 0; i < N; ++i) h->betas[i] = equally_spaced(N, i);
        e.ref2 = -1; e.tgt_a = e.tgt_b = e.rt_tgt_a = e.rt_tgt_b = N - 1;
    }
    if (upload_ladder(h)) return bail(1);
    if (reset_recorders(h)) return bail(1);
    const double init_sd = swapper ? 1.0 : std::sqrt(cfg->target_params[1]);   // toy_mvn_target.jl:10-11
    hipEvent_t init_a = nullptr, init_b = nullptr;       // k_init's duration: pte_timing_get(kernel = 2), one event pair per engine
    if (!ising) {
        hipEventCreate(&init_a); hipEventCreate(&init_b);
        {   // the events ride on the launch (PTE_LAUNCH1): the kernel's own begin / end
            pte_engine::Ev ev; ev.a = init_a; ev.b = init_b; ev.kernel = 2;
            h->events.push_back(ev); h->ev_open = true; h->ev_ext = true; h->ev_ext_done = false;
            DISPATCH_NLU(h->nlu, k_init, dim3((unsigned)((K + NRM_WPB - 1) / NRM_WPB)), dim3(64 * NRM_WPB), h->stream, e, (uint64_t)cfg->seed, init_sd);
            h->events.pop_back(); h->ev_open = false; h->ev_ext = false;
        }
    }
    else {
        std::vector<int32_t> ch((size_t)K), sl((size_t)K); std::vector<int64_t> rid((size_t)K);
        for (int64_t il = 0; il < K; ++il) { ch[il] = (int32_t)(h->c0 + il); sl[il] = (int32_t)il; rid[il] = h->c0 + il; }
        hipMemcpyAsync(e.chain_of_slot, ch.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, h->stream);
        hipMemcpyAsync(e.slot_of_chain, sl.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, h->stream);
        hipMemcpyAsync(e.replica_id, rid.data(), sizeof(int64_t) * K, hipMemcpyHostToDevice, h->stream);
        hipStreamSynchronize(h->stream);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
        h->err = "k_init launch failed"; return bail(1);
    }
    if (init_a) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, init_a, init_b) == hipSuccess) h->init_ms = ms;
        hipEventDestroy(init_a); hipEventDestroy(init_b);
    }
    if (funnel || ising) {
        // funnel: initialization(::LogDensity, rng, i) = zeros(dim); Ising: falses(L, L) (examples/ising.jl:85) (test/supporting/dimensional-analysis.jl:24): the streams
        // stay untouched; suff2 = funnel(0) = d terms evaluated on the host exactly like the kernels' tree of equal terms
        std::vector<uint64_t> rngs((size_t)(2 * K));
        const uint64_t G = 0x9e3779b97f4a7c15ULL;
        auto mix64h = [](uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
        auto mixg = [](uint64_t z) { z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL; z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL; z = (z ^ (z >> 33)) | 1ULL;
                                     return (__builtin_popcountll(z ^ (z >> 1)) < 24) ? (z ^ 0xaaaaaaaaaaaaaaaaULL) : z; };
        for (int64_t il = 0; il < K; ++il) {
            const uint64_t i = (uint64_t)(h->c0 + il);
            rngs[2 * il] = mix64h(cfg->seed + (2 * i + 1) * G); rngs[2 * il + 1] = mixg(cfg->seed + (2 * i + 2) * G);
        }
        hipMemcpyAsync(e.rng, rngs.data(), sizeof(uint64_t) * 2 * K, hipMemcpyHostToDevice, h->stream);
        hipMemsetAsync(e.x, 0, sizeof(double) * K * e.ld, h->stream);
        hipMemsetAsync(e.suff, 0, sizeof(double) * K, h->stream);
        // funnel(0): terms[0] = -(log2pi)/2 - log 3 ; terms[i] = -(log2pi)/2 - log(exp(0)) ; summed with the fixed tree
        std::vector<double> terms((size_t)d);
        const double LOG2PI = 1.8378770664093453;
        const double sigma = std::exp(0.0 / 2.0), ls = std::log(sigma);
        for (int64_t i = 0; i < d; ++i) terms[i] = -(0.0 * 0.0 + LOG2PI) / 2.0 - (i == 0 ? std::log(3.0) : ls);
        int64_t P = 1; while (P < d) P <<= 1;
        std::vector<double> a((size_t)P, 0.0);
        for (int64_t i = 0; i < d; ++i) a[i] = terms[i];
        for (int64_t len = P; len > 1; len /= 2) for (int64_t i = 0; i < len / 2; ++i) a[i] = a[2 * i] + a[2 * i + 1];
        std::vector<double> s2((size_t)K, a[0]);
        hipMemcpyAsync(e.suff2, s2.data(), sizeof(double) * K, hipMemcpyHostToDevice, h->stream);
        if (ising) {   // all spins -1: every site contributes (-1)(-4) = 4, halved: sum_pair_products = 2 L^2
            std::vector<double> spp((size_t)K, 2.0 * (double)d);
            hipMemcpyAsync(e.suff, spp.data(), sizeof(double) * K, hipMemcpyHostToDevice, h->stream);
            if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "ising init failed"; return bail(1); }
        }
        if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "funnel init failed"; return bail(1); }
    }
    *out = h;
    return 0;
}

int pte_destroy(pte_engine *h) {
    if (!h) return 0;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    time_collect(h);
    pte_comm_destroy(h);
    if (h->ev_pack) hipEventDestroy(h->ev_pack);
    if (h->ev_copied) hipEventDestroy(h->ev_copied);
    for (hipEvent_t ev : h->ev_pool) hipEventDestroy(ev);
    for (void *p : h->allocs) hipFree(p);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int pte_set_schedule(pte_engine *h, const double *betas, int64_t n) {
    if (!h || !betas) return 1;
    PTE_ALIVE(h, "pte_set_schedule");
    if (n != h->N) return fail(h, "pte_set_schedule: expected %lld grid points, got %lld", (long long)h->N, (long long)n);
    // Schedule constructor asserts (reference src/schedules/Schedule.jl:14-27)
    auto check_leg = [&](const double *b, int64_t m, int64_t stride) -> bool {   // grid of one leg, reference -> target
        if (m == 1) return b[0] == 1.0;
        if (b[0] != 0.0 || b[(m - 1) * stride] != 1.0) return false;
        for (int64_t i = 0; i + 1 < m; ++i) if (!(b[i * stride] < b[(i + 1) * stride])) return false;
        return true;
    };
    if (h->cfg.n_chains_variational > 0) {
        const int64_t nv = h->cfg.n_chains_variational, nf = h->cfg.n_chains;
        if (!check_leg(betas, nv, 1) || !check_leg(betas + n - 1, nf, -1)) return fail(h, "Invalid schedule (two legs: each leg must run 0 -> 1)");
    } else if (!check_leg(betas, n, 1)) return fail(h, n == 1 ? "Invalid schedule" : "Invalid schedule: end points must be 0 and 1, strictly increasing");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    h->betas.assign(betas, betas + n);
    return upload_ladder(h);
}

int pte_get_schedule(const pte_engine *h, double *betas) {
    if (!h || !betas) return 1;
    std::memcpy(betas, h->betas.data(), sizeof(double) * h->N);
    return 0;
}

int pte_set_explorer_adaptation(pte_engine *h, double step_size, const double *target_std, int64_t dim) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_set_explorer_adaptation");
    {   // nothing to adapt for SliceSampler / ToyExplorer
        auto gb = [](int k) { return k == PTE_EXPLORER_AUTOMALA || k == PTE_EXPLORER_MALA; };
        if (!gb(h->cfg.explorer) && !gb(h->cfg.explorer2)) return 0;
    }
    if (!(step_size > 0)) return fail(h, "pte_set_explorer_adaptation: step_size must be > 0");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    h->step_size = step_size;
    if (target_std) {
        if (dim != h->d) return fail(h, "pte_set_explorer_adaptation: expected %lld std deviations", (long long)h->d);
        HIP_OK(h, hipMemcpyAsync(h->d_target_std, target_std, sizeof(double) * dim, hipMemcpyHostToDevice, h->stream));
        HIP_OK(h, hipStreamSynchronize(h->stream));
        h->have_target_std = true;
    }
    return 0;
}

int pte_explore(pte_engine *h, int64_t scan) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_explore");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    if (launch_explore(h, scan)) return 1;
    return check_device_error(h);
}

int pte_swap(pte_engine *h, int64_t scan) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_swap");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    if (launch_swap(h, scan)) return 1;
    return check_device_error(h);
}

int pte_run_scans(pte_engine *h, int64_t first_scan, int64_t n_scans) {
    if (!h) return 1;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    PTE_ALIVE(h, "pte_run_scans");
    if (h->world != 1) return run_scans_sharded(h, first_scan, n_scans);
    if (fused_scans_eligible(h, n_scans)) {
        if (h->fused_skip > 0) h->fused_skip -= 1;         // a recent launch found the device shared: not this call (run_scans_fused)
        else return run_scans_fused(h, first_scan, n_scans);
    }
    return run_scans_two_launches(h, first_scan, n_scans);
}

// PTE_RECORD_REFERENCE_REDUCTION: swap_acceptance_pr and log_sum_ratio reduced as the reference reduces them.  There every REPLICA owns a
// GroupBy(chain pair => Mean) and a GroupBy(chain pair => LogSum) and fits them when it holds the lower chain of a swapping pair
// (record_swap_stats!, src/swap/pair_swapper.jl:59-66); at the end of a round the replicas' recorders are merged over the binary tree on the
// replica index -- spacing 1, 2, 4, ...: replica i takes replica i + s (all_reduce_deterministically, src/mpi_utils/Entangler.jl:188-251;
// reduce_recorders!, src/recorders/recorders.jl:88-130).  Mean: fit mu += (1/n)(x - mu), merge mu += (n_b/n)(mu_b - mu) (OnlineStats);
// LogSum: fit / merge by logaddexp (src/recorders/LogSum.jl:1-24; LogExpFunctions: max + log1p(exp(-|x - y|)), exp(-|x - y|) below -37).
// The device keeps chain-keyed sums (one replica's worth of arithmetic per pair: no order to agree on, 1e-12 away); here the same numbers are
// rebuilt from the log {lr of the lower chain's replica, lr of the upper's} per scan and pair, and index_process says whose recorder each fit went to.
static inline double host_logaddexp(double x, double y) {
    const double delta = (x == y) ? 0.0 : std::fabs(x - y);
    const double m = (x > y) ? x : y, nd = -delta;
    return m + ((nd <= -37.0) ? std::exp(nd) : std::log1p(std::exp(nd)));
}
static int reference_reduce(pte_engine *h, Snapshot &s) {
    // A chain-shard replays ITS pairs (those whose lower chain it owns) by itself: the log and index_process rows of a pair live with its lower
    // chain, and the merge tree runs over the GLOBAL replica index whichever shard a replica visited.
    const int64_t K = h->K, N = h->N, c0 = h->c0, T = s.n_scans, pairs = (c0 + K < N) ? K : K - 1;
    if (T == 0) return 0;
    std::vector<double> log((size_t)(T * K * 2));
    HIP_OK(h, hipMemcpy(log.data(), h->dev.swap_log, sizeof(double) * log.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> holder((size_t)(T * K));                         // [scan][local chain] -> replica
    for (int64_t t = 0; t < T; ++t)
        for (int64_t slot = 0; slot < K; ++slot) holder[(size_t)(t * K + (s.ip_chain[(size_t)(t * K + slot)] - c0))] = s.ip_replica[(size_t)(t * K + slot)];
    struct Rec { double mu; int64_t n; double up, dn; };                  // one replica's Mean and two LogSums of ONE pair (their counts move together)
    std::vector<Rec> rec((size_t)N);
    for (int64_t c = 0; c < pairs; ++c) {
        for (auto &r : rec) r = Rec{0.0, 0, -INFINITY, -INFINITY};
        for (int64_t t = 0; t < T; ++t) {
            const double *w = &log[(size_t)((t * K + c) * 2)];
            uint64_t bits; std::memcpy(&bits, w, 8);
            if (bits == ~0ull) continue;                                  // the pair was idle on this scan's graph
            Rec &r = rec[(size_t)holder[(size_t)(t * K + c)]];
            const double ex = std::exp(w[0] + w[1]), alpha = ex < 1.0 ? ex : 1.0;    // swap_acceptance_probability, pair_swapper.jl:88
            r.n += 1;
            r.mu = r.mu + (1.0 / (double)r.n) * (alpha - r.mu);
            r.up = host_logaddexp(r.up, w[0]);
            r.dn = host_logaddexp(r.dn, w[1]);
        }
        for (int64_t sp = 1; sp < N; sp *= 2)
            for (int64_t i = 0; i + sp < N; i += 2 * sp) {
                Rec &a = rec[(size_t)i]; const Rec &b = rec[(size_t)(i + sp)];
                if (b.n == 0) continue;                                   // GroupBy merge: the key is absent on that side
                if (a.n == 0) { a = b; continue; }
                a.n += b.n;
                a.mu = a.mu + ((double)b.n / (double)a.n) * (b.mu - a.mu);
                a.up = host_logaddexp(a.up, b.up);
                a.dn = host_logaddexp(a.dn, b.dn);
            }
        if (rec[0].n != s.swap_n[(size_t)c])
            return fail(h, "PTE_RECORD_REFERENCE_REDUCTION: the log holds %lld swaps of pair %lld, the device counted %lld", (long long)rec[0].n, (long long)(c0 + c), (long long)s.swap_n[(size_t)c]);
        if (rec[0].n > 0) { s.swap_mean[(size_t)c] = rec[0].mu; s.lsr_up[(size_t)c] = rec[0].up; s.lsr_dn[(size_t)c] = rec[0].dn; }
    }
    // am_factors (AutoMALA.jl:277: one fit of 2^exponent per step-size search, keyed by the chain) the same way -- it is what the step size
    // adapts on (AutoMALA.jl:70-79), i.e. what decides whether the NEXT round's states equal the reference's or sit an ulp away
    if (h->dev.am_log) {
        const int cap = h->dev.am_log_cap;
        std::vector<int16_t> al((size_t)(T * K * cap));
        HIP_OK(h, hipMemcpy(al.data(), h->dev.am_log, sizeof(int16_t) * al.size(), hipMemcpyDeviceToHost));
        // reversibility_rate (AutoMALA.jl:294: one fit of `reversed exponent == proposed exponent` per refresh with an MH step) comes out of the same log:
        // a refresh logs its forward search, then -- scan != 1 -- its reversed one, so a scan's row holds n_refresh entries (no MH step) or n_refresh PAIRS.
        // Nothing adapts on it; replayed so that it EQUALS the reference's number too instead of sitting 1e-16 away.
        struct Mn { double mu; int64_t n; };
        std::vector<Mn> mn((size_t)N), rv((size_t)N);
        auto merge = [&](std::vector<Mn> &v) {
            for (int64_t sp = 1; sp < N; sp *= 2)
                for (int64_t i = 0; i + sp < N; i += 2 * sp) {
                    Mn &a = v[(size_t)i]; const Mn &b = v[(size_t)(i + sp)];
                    if (b.n == 0) continue;
                    if (a.n == 0) { a = b; continue; }
                    a.n += b.n;
                    a.mu = a.mu + ((double)b.n / (double)a.n) * (b.mu - a.mu);
                }
        };
        const int nref = h->am_n_refresh;
        for (int64_t c = 0; c < K; ++c) {
            for (auto &r : mn) r = Mn{0.0, 0};
            for (auto &r : rv) r = Mn{0.0, 0};
            bool pairs_ok = true;
            for (int64_t t = 0; t < T; ++t) {
                Mn &r = mn[(size_t)holder[(size_t)(t * K + c)]];
                const int16_t *a = &al[(size_t)((t * K + c) * cap)];
                int m = 0;
                for (int j = 0; j < cap && a[j] != (int16_t)0x7F7F; ++j) {
                    r.n += 1;
                    r.mu = r.mu + (1.0 / (double)r.n) * (std::ldexp(1.0, (int)a[j]) - r.mu);
                    m += 1;
                }
                if (m == 2 * nref) {
                    Mn &q = rv[(size_t)holder[(size_t)(t * K + c)]];
                    for (int j = 0; j < m; j += 2) {
                        q.n += 1;
                        q.mu = q.mu + (1.0 / (double)q.n) * ((a[j] == a[j + 1] ? 1.0 : 0.0) - q.mu);
                    }
                } else if (m != nref && m != 0) pairs_ok = false;           // (a search that failed mid-refresh: the call has reported it; leave the device's sums)
            }
            merge(mn);
            if (mn[0].n != h->fac_n[(size_t)c])
                return fail(h, "PTE_RECORD_REFERENCE_REDUCTION: the log holds %lld step-size searches of chain %lld, the device counted %lld", (long long)mn[0].n, (long long)(c0 + c), (long long)h->fac_n[(size_t)c]);
            if (mn[0].n > 0) h->fac_mean[(size_t)c] = mn[0].mu;
            if (pairs_ok && h->cfg.explorer2 == PTE_EXPLORER_NONE) {
                merge(rv);
                if (rv[0].n == h->rev_n[(size_t)c] && rv[0].n > 0) h->rev_mean[(size_t)c] = rv[0].mu;
            }
        }
    }
    // energy_ac1 (recorder.jl:113: GroupBy(Int, CovMatrix(2)) fitted with (log density before, after) the explore step, src/pt/pigeons.jl:133-143) the same way:
    // OnlineStats' CovMatrix keeps b = mean and A = mean of x x' with a += (1/n)(x - a), merges with n_b / n, value = (A - b b') n / (n - 1), cor = D^-1/2 value D^-1/2.
    // The device's chain-keyed Welford co-moments (pte_get_energy_ac1's `moments`) stay what they are; `cor` -- what energy_ac1s reports -- becomes the reference's number.
    if (h->dev.eac_log) {
        std::vector<double> el((size_t)(T * K * 2));
        HIP_OK(h, hipMemcpy(el.data(), h->dev.eac_log, sizeof(double) * el.size(), hipMemcpyDeviceToHost));
        struct Cv { int64_t n; double b[2], A[3]; };
        std::vector<Cv> cv((size_t)N);
        for (int64_t c = 0; c < K; ++c) {
            for (auto &r : cv) r = Cv{0, {0.0, 0.0}, {0.0, 0.0, 0.0}};
            for (int64_t t = 0; t < T; ++t) {
                const double *w = &el[(size_t)((t * K + c) * 2)];
                uint64_t bits; std::memcpy(&bits, w, 8);
                if (bits == ~0ull) continue;
                Cv &o = cv[(size_t)holder[(size_t)(t * K + c)]];
                o.n += 1;
                const double g = 1.0 / (double)o.n, x0 = w[0], x1 = w[1];
                o.b[0] += g * (x0 - o.b[0]); o.b[1] += g * (x1 - o.b[1]);
                o.A[0] += g * (x0 * x0 - o.A[0]); o.A[1] += g * (x0 * x1 - o.A[1]); o.A[2] += g * (x1 * x1 - o.A[2]);
            }
            for (int64_t sp = 1; sp < N; sp *= 2)
                for (int64_t i = 0; i + sp < N; i += 2 * sp) {
                    Cv &a = cv[(size_t)i]; const Cv &b = cv[(size_t)(i + sp)];
                    if (b.n == 0) continue;
                    if (a.n == 0) { a = b; continue; }
                    a.n += b.n;
                    const double g = (double)b.n / (double)a.n;
                    for (int k = 0; k < 3; ++k) a.A[k] += g * (b.A[k] - a.A[k]);
                    for (int k = 0; k < 2; ++k) a.b[k] += g * (b.b[k] - a.b[k]);
                }
            const Cv &o = cv[0];
            if (o.n != s.eac_n[(size_t)c])
                return fail(h, "PTE_RECORD_REFERENCE_REDUCTION: the log holds %lld explore steps of chain %lld, the device counted %lld", (long long)o.n, (long long)(c0 + c), (long long)s.eac_n[(size_t)c]);
            if (o.n > 1) {
                const double bes = (double)o.n / (double)(o.n - 1);
                const double c00 = (o.A[0] - o.b[0] * o.b[0]) * bes, c01 = (o.A[1] - o.b[0] * o.b[1]) * bes, c11 = (o.A[2] - o.b[1] * o.b[1]) * bes;
                const double v0 = 1.0 / std::sqrt(c00), v1 = 1.0 / std::sqrt(c11);
                s.eac_cor[(size_t)c] = (c01 * v1) * v0;
            }
        }
    }
    // :online / :_transformed_online the same way, when the round's traces are there to replay them from (PTE_RECORD_TRACES: the target
    // chains' [state; log density] per scan are exactly what explore! fits, src/pt/pigeons.jl:116-131): every replica's Mean and Variance
    // (OnlineStats: mu += g (x - mu), s2 += g ((x - mu_new)(x - mu_old) - s2), g = 1/n; merge with g = n_b / n) per coordinate, tree-merged.
    // Without traces the device's Welford sums of the target chains stand (one chain's worth of arithmetic: ~1e-16 relative away).
    const uint32_t f = h->cfg.record_flags;
    const EngineDev &e = h->dev;
    const bool owns_targets = e.tgt_a >= c0 && e.tgt_a < c0 + K && e.tgt_b >= c0 && e.tgt_b < c0 + K;
    if ((f & PTE_RECORD_ONLINE) && (f & PTE_RECORD_TRACES) && s.traces_n == T && owns_targets) {
        const int64_t d = h->d, ntgt = (e.tgt_b != e.tgt_a) ? 2 : 1, rows = (f & PTE_RECORD_TRACES_EXTENDED) ? K : ntgt;
        const int64_t tgt[2] = {e.tgt_a - c0, e.tgt_b - c0};
        struct On { double mu, s2; int64_t n; };
        std::vector<On> on((size_t)N);
        for (int64_t i = 0; i <= d; ++i) {
            for (auto &r : on) r = On{0.0, 0.0, 0};
            for (int64_t t = 0; t < T; ++t)
                for (int64_t w = 0; w < ntgt; ++w) {
                    const int64_t row = (f & PTE_RECORD_TRACES_EXTENDED) ? tgt[w] : w;
                    const double x = s.traces[(size_t)(((t * rows) + row) * (d + 1) + i)];
                    On &r = on[(size_t)holder[(size_t)(t * K + tgt[w])]];
                    const double mu0 = r.mu;
                    r.n += 1;
                    const double g = 1.0 / (double)r.n;
                    r.mu = r.mu + g * (x - r.mu);
                    r.s2 = r.s2 + g * ((x - r.mu) * (x - mu0) - r.s2);
                }
            for (int64_t sp = 1; sp < N; sp *= 2)
                for (int64_t j = 0; j + sp < N; j += 2 * sp) {
                    On &a = on[(size_t)j]; const On &b = on[(size_t)(j + sp)];
                    if (b.n == 0) continue;
                    if (a.n == 0) { a = b; continue; }
                    a.n += b.n;
                    const double g = (double)b.n / (double)a.n, delta = b.mu - a.mu;
                    a.s2 = (a.s2 + g * (b.s2 - a.s2)) + delta * delta * g * (1.0 - g);
                    a.mu = a.mu + g * (b.mu - a.mu);
                }
            if (on[0].n != s.on_n)
                return fail(h, "PTE_RECORD_REFERENCE_REDUCTION: the traces hold %lld target-chain samples, the device counted %lld", (long long)on[0].n, (long long)s.on_n);
            s.on_mean[(size_t)i] = on[0].mu;
            s.on_var[(size_t)i] = on[0].n > 1 ? on[0].s2 * ((double)on[0].n / (double)(on[0].n - 1)) : 1.0;
        }
    }
    return 0;
}

int pte_reduce(pte_engine *h) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_reduce");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t K = h->K, d = h->d;
    EngineDev &e = h->dev;
    Snapshot &s = h->snap;
    std::vector<double> swap_sum(K);
    s.swap_mean.assign(K, 0.0); s.swap_n.assign(K, 0);
    s.lsr_up.assign(K, 0.0); s.lsr_dn.assign(K, 0.0); s.lsr_n.assign(K, 0);
    std::vector<int64_t> rs(K), rr(K);
    std::vector<double> acc_sum(K);
    s.acc_mean.assign(K, 0.0); s.acc_n.assign(K, 0); s.steps_sum.assign(K, 0.0); s.steps_n.assign(K, 0);
    const int64_t dd = d > 0 ? d : 1;
    std::vector<double> m2(dd);
    s.on_mean.assign(2 * (d + 1), 0.0); s.on_var.assign(d + 1, 0.0); m2.assign(2 * (d + 1), 0.0);
    int64_t on_n2[2] = {0, 0};
    s.eac_raw.assign(5 * K, 0.0); s.eac_n.assign(K, 0); s.eac_cor.assign(K, NAN);
#define D2H(dst, src, n) HIP_OK(h, hipMemcpyAsync(dst, src, sizeof(*(dst)) * (n), hipMemcpyDeviceToHost, h->stream))
    D2H(swap_sum.data(), e.swap_sum, K); D2H(s.swap_n.data(), e.swap_n, K);
    D2H(s.lsr_up.data(), e.lsr_up, K);   D2H(s.lsr_dn.data(), e.lsr_dn, K);   D2H(s.lsr_n.data(), e.lsr_n, K);
    D2H(rs.data(), e.rt_restarts, K);     D2H(rr.data(), e.rt_trips, K);
    D2H(acc_sum.data(), e.expl_acc_sum, K); D2H(s.acc_n.data(), e.expl_acc_n, K);
    D2H(s.steps_sum.data(), e.expl_steps_sum, K); D2H(s.steps_n.data(), e.expl_steps_n, K);
    D2H(s.on_mean.data(), e.on_mean, 2 * (d + 1)); D2H(m2.data(), e.on_m2, 2 * (d + 1)); D2H(on_n2, e.on_n, 2);
    D2H(s.eac_raw.data(), e.eac, 5 * K); D2H(s.eac_n.data(), e.eac_n, K);
    const bool ext_traces = (h->cfg.record_flags & PTE_RECORD_TRACES_EXTENDED) != 0;
    s.traces_n = (h->cfg.record_flags & PTE_RECORD_TRACES) && (ext_traces || h->c0 + K == h->N) ? h->scans_in_round : 0;
    const size_t trace_words = (size_t)(s.traces_n * (ext_traces ? K : (h->cfg.n_chains_variational > 0 ? 2 : 1)) * (d + 1));
    s.traces.assign(trace_words, 0.0);
    if (s.traces_n > 0) D2H(s.traces.data(), e.traces, trace_words);
    std::vector<double> fsum(K), rsum(K);
    h->fac_mean.assign(K, 0.0); h->rev_mean.assign(K, 0.0); h->fac_n.assign(K, 0); h->rev_n.assign(K, 0);
    D2H(fsum.data(), e.am_fac_sum, K); D2H(h->fac_n.data(), e.am_fac_n, K);
    D2H(rsum.data(), e.am_rev_sum, K); D2H(h->rev_n.data(), e.am_rev_n, K);
    s.n_scans = h->scans_in_round;
    s.ip_chain.clear(); s.ip_replica.clear();
    if (h->cfg.record_flags & PTE_RECORD_INDEX_PROCESS) {
        s.ip_chain.resize((size_t)(s.n_scans * K)); s.ip_replica.resize((size_t)(s.n_scans * K));
        if (!s.ip_chain.empty()) {
            D2H(s.ip_chain.data(), e.index_process, (size_t)(s.n_scans * K));
            D2H(s.ip_replica.data(), e.ip_replica, (size_t)(s.n_scans * K));
        }
    }
#undef D2H
    HIP_OK(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < K; ++i) s.swap_mean[i] = s.swap_n[i] > 0 ? swap_sum[i] / (double)s.swap_n[i] : 0.0;
    s.restarts = 0; s.trips = 0;
    for (int64_t i = 0; i < K; ++i) { s.restarts += rs[i]; s.trips += rr[i]; }
    for (int64_t i = 0; i < K; ++i) s.acc_mean[i] = s.acc_n[i] > 0 ? acc_sum[i] / (double)s.acc_n[i] : 0.0;
    {   // merge the two target chains' Welford sets (OnlineStats merge of Mean / Variance; one set when there is one leg)
        const int64_t na = on_n2[0], nb = on_n2[1];
        s.on_n = na + nb;
        for (int64_t i = 0; i <= d; ++i) {
            const double ma = s.on_mean[i], mb = s.on_mean[d + 1 + i];
            double mean = ma, M2 = m2[i];
            if (nb > 0) {
                const double delta = mb - ma;
                mean = na > 0 ? ma + delta * ((double)nb / (double)(na + nb)) : mb;
                M2 = m2[i] + m2[d + 1 + i] + (na > 0 ? delta * delta * ((double)na * (double)nb / (double)(na + nb)) : 0.0);
            }
            s.on_mean[i] = mean;
            s.on_var[i] = s.on_n > 1 ? M2 / (double)(s.on_n - 1) : 1.0;
        }
        s.on_mean.resize(d + 1);
    }
    for (int64_t i = 0; i < K; ++i)       // cor(CovMatrix)[1,2]: the Bessel factors cancel
        if (s.eac_n[i] > 1) s.eac_cor[i] = s.eac_raw[5 * i + 3] / std::sqrt(s.eac_raw[5 * i + 2] * s.eac_raw[5 * i + 4]);
    for (int64_t i = 0; i < K; ++i) {
        h->fac_mean[i] = h->fac_n[i] > 0 ? fsum[i] / (double)h->fac_n[i] : 0.0;
        h->rev_mean[i] = h->rev_n[i] > 0 ? rsum[i] / (double)h->rev_n[i] : 0.0;
    }
    if (e.swap_log && reference_reduce(h, s)) return 1;       // (PTE_RECORD_REFERENCE_REDUCTION: replaces the swap recorders, am_factors and, given traces, online)
    return reset_recorders(h);
}

static int64_t local_pairs(const pte_engine *h) { return (h->c0 + h->K < h->N) ? h->K : h->K - 1; }

int pte_shard_info(const pte_engine *h, int64_t *c0, int64_t *K, int64_t *n_pairs) {
    if (!h) return 1;
    if (c0) *c0 = h->c0;
    if (K) *K = h->K;
    if (n_pairs) *n_pairs = local_pairs(h);
    return 0;
}
int pte_get_swap_acceptance(const pte_engine *h, double *mean, int64_t *n) {
    if (!h) return 1;
    for (int64_t i = 0; i < local_pairs(h); ++i) { mean[i] = h->snap.swap_mean[i]; n[i] = h->snap.swap_n[i]; }
    return 0;
}
int pte_get_log_sum_ratio(const pte_engine *h, double *up, int64_t *up_n, double *dn, int64_t *dn_n) {
    if (!h) return 1;
    for (int64_t i = 0; i < local_pairs(h); ++i) {
        up[i] = h->snap.lsr_up[i]; dn[i] = h->snap.lsr_dn[i];
        up_n[i] = h->snap.lsr_n[i]; dn_n[i] = h->snap.lsr_n[i];
    }
    return 0;
}
int pte_get_round_trip(const pte_engine *h, int64_t *restarts, int64_t *trips) {
    if (!h) return 1;
    *restarts = h->snap.restarts; *trips = h->snap.trips;
    return 0;
}
int pte_get_index_process(const pte_engine *h, int64_t *out, int64_t *n_scans) {
    if (!h) return 1;
    if (h->world != 1) return fail(const_cast<pte_engine *>(h), "pte_get_index_process: sharded engines use pte_get_index_process_shard");
    const int64_t T = h->snap.ip_chain.empty() ? 0 : h->snap.n_scans, N = h->N;
    if (n_scans) *n_scans = T;
    if (out && !h->snap.ip_chain.empty())
        for (int64_t t = 0; t < T; ++t)
            for (int64_t sl = 0; sl < N; ++sl)
                out[(size_t)(h->snap.ip_replica[(size_t)(t * N + sl)] * T + t)] = h->snap.ip_chain[(size_t)(t * N + sl)];
    return 0;
}
int pte_get_index_process_shard(const pte_engine *h, int64_t *replica, int64_t *chain, int64_t *n_scans) {
    if (!h) return 1;
    if (n_scans) *n_scans = h->snap.ip_chain.empty() ? 0 : h->snap.n_scans;
    const size_t n = h->snap.ip_chain.size();
    for (size_t i = 0; i < n; ++i) {
        if (replica) replica[i] = h->snap.ip_replica[i];
        if (chain) chain[i] = h->snap.ip_chain[i];
    }
    return 0;
}
int pte_get_explorer_stats(const pte_engine *h, double *am, int64_t *an, double *ss, int64_t *sn) {
    if (!h) return 1;
    for (int64_t i = 0; i < h->K; ++i) {
        am[i] = h->snap.acc_mean[i]; an[i] = h->snap.acc_n[i];
        ss[i] = h->snap.steps_sum[i]; sn[i] = h->snap.steps_n[i];
    }
    return 0;
}
int pte_get_automala_stats(const pte_engine *h, double *fm, int64_t *fn, double *rm, int64_t *rn) {
    if (!h) return 1;
    for (int64_t i = 0; i < h->K; ++i) {
        const bool have = (int64_t)h->fac_mean.size() == h->K;
        fm[i] = have ? h->fac_mean[i] : 0; fn[i] = have ? h->fac_n[i] : 0;
        rm[i] = have ? h->rev_mean[i] : 0; rn[i] = have ? h->rev_n[i] : 0;
    }
    return 0;
}
int pte_get_online(const pte_engine *h, double *mean, double *variance, int64_t *n) {
    if (!h) return 1;
    for (int64_t i = 0; i < h->d; ++i) { mean[i] = h->snap.on_mean[i]; variance[i] = h->snap.on_var[i]; }
    if (n) *n = h->snap.on_n;
    return 0;
}
int pte_get_online_log_density(const pte_engine *h, double *mean, double *variance) {
    if (!h || !mean || !variance) return 1;
    *mean = h->snap.on_mean.size() > (size_t)h->d ? h->snap.on_mean[h->d] : 0.0;
    *variance = h->snap.on_var.size() > (size_t)h->d ? h->snap.on_var[h->d] : 0.0;
    return 0;
}
int pte_get_energy_ac1(const pte_engine *h, double *cor, int64_t *n, double *moments) {
    if (!h) return 1;
    for (int64_t i = 0; i < h->K && (size_t)i < h->snap.eac_n.size(); ++i) {
        if (cor) cor[i] = h->snap.eac_cor[i];
        if (n) n[i] = h->snap.eac_n[i];
        if (moments) for (int k = 0; k < 5; ++k) moments[5 * i + k] = h->snap.eac_raw[5 * i + k];
    }
    return 0;
}
int pte_get_traces(const pte_engine *h, double *out, int64_t *n_scans) {
    if (!h || !n_scans) return 1;
    *n_scans = h->snap.traces_n;
    if (out && !h->snap.traces.empty()) std::memcpy(out, h->snap.traces.data(), sizeof(double) * h->snap.traces.size());
    return 0;
}
int pte_get_replica_ids(const pte_engine *hc, int64_t *out) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h || !out) return 1;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    HIP_OK(h, hipMemcpyAsync(out, h->dev.replica_id, sizeof(int64_t) * h->K, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- two-phase swap of chain-sharded engines ---------------------------------------------------
int pte_swap_begin(pte_engine *h, int64_t scan, double *stats_out, int32_t *active_out) {
    if (!h || !stats_out || !active_out) return 1;
    PTE_ALIVE(h, "pte_swap_begin");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t K = h->K;
    if ((h->cfg.record_flags & PTE_RECORD_INDEX_PROCESS) && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "index_process buffer full (max_scans_per_round = %lld)", (long long)h->cfg.max_scans_per_round);
    const int even = (scan % 2 == 0) ? 1 : 0;
    const unsigned block = 256, grid = (unsigned)((K + block - 1) / block);
    time_begin(h, 1);
    hipLaunchKernelGGL(k_swap_stats, dim3(grid), dim3(block), 0, h->stream, h->dev, even, h->scans_in_round);
    time_end(h);
    HIP_OK(h, hipGetLastError());
    boundary_active(h, even, active_out);
    HIP_OK(h, hipMemcpyAsync(stats_out, h->dev.stat, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipMemcpyAsync(stats_out + 2, h->dev.stat + 2 * (K - 1), 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int pte_swap_finish(pte_engine *h, int64_t scan, const double *nbr_stats, int32_t *accepted_out) {
    if (!h || !nbr_stats || !accepted_out) return 1;
    PTE_ALIVE(h, "pte_swap_finish");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t K = h->K;
    const int even = (scan % 2 == 0) ? 1 : 0;
    HIP_OK(h, hipMemcpyAsync(h->dev.nbr_stat, nbr_stats, 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemsetAsync(h->dev.bflag, 0, 2 * sizeof(int32_t), h->stream));
    const unsigned block = 256, grid = (unsigned)((K + block - 1) / block);
    time_begin(h, 1);
    hipLaunchKernelGGL(k_swap_decide, dim3(grid), dim3(block), 0, h->stream, h->dev, even, h->scans_in_round);
    time_end(h);
    HIP_OK(h, hipGetLastError());
    h->slot_cur ^= 1;                                   // the decide kernel wrote the new chain -> slot map
    h->dev.slot_of_chain = h->slot_map[h->slot_cur];
    h->dev.slot_of_chain_alt = h->slot_map[h->slot_cur ^ 1];
    HIP_OK(h, hipMemcpyAsync(accepted_out, h->dev.bflag, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    h->scans_in_round += 1;
    int rc = check_device_error(h);                     // synchronises
    time_collect(h);
    return rc;
}
int64_t pte_boundary_payload_bytes(const pte_engine *h) { return h ? (int64_t)sizeof(double) * (h->dev.sw + 6) : 0; }
int pte_boundary_export(pte_engine *h, int side, void *dst, int dst_is_device) {
    if (!h || !dst || side < 0 || side > 1) return 1;
    PTE_ALIVE(h, "pte_boundary_export");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    double *buf = dst_is_device ? (double *)dst : h->d_payload;
    hipLaunchKernelGGL(k_boundary_export, dim3(1), dim3(256), 0, h->stream, h->dev, side, buf);
    HIP_OK(h, hipGetLastError());
    if (!dst_is_device)
        HIP_OK(h, hipMemcpyAsync(dst, h->d_payload, (size_t)pte_boundary_payload_bytes(h), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int pte_boundary_import(pte_engine *h, int side, const void *src, int src_is_device) {
    if (!h || !src || side < 0 || side > 1) return 1;
    PTE_ALIVE(h, "pte_boundary_import");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const double *buf = (const double *)src;
    if (!src_is_device) {
        HIP_OK(h, hipMemcpyAsync(h->d_payload, src, (size_t)pte_boundary_payload_bytes(h), hipMemcpyHostToDevice, h->stream));
        buf = h->d_payload;
    }
    hipLaunchKernelGGL(k_boundary_import, dim3(1), dim3(256), 0, h->stream, h->dev, side, buf);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- device-resident, stream-ordered boundary exchange -------------------------------------------
void *pte_get_stream(const pte_engine *h) { return h ? (void *)h->stream : nullptr; }
int64_t pte_shard_message_bytes(const pte_engine *h) { return h ? (int64_t)sizeof(double) * (h->dev.sw + 8) : 0; }
int pte_shard_set_buffers(pte_engine *h, void *send_lo, void *recv_lo, void *send_hi, void *recv_hi) {
    if (!h) return 1;
    h->msg_send[0] = (double *)send_lo; h->msg_recv[0] = (double *)recv_lo;
    h->msg_send[1] = (double *)send_hi; h->msg_recv[1] = (double *)recv_hi;
    return 0;
}
int pte_shard_scan_begin(pte_engine *h, int64_t scan, int32_t *active_out) {
    if (!h || !active_out) return 1;
    PTE_ALIVE(h, "pte_shard_scan_begin");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    if (!h->msg_send[0] || !h->msg_recv[0] || !h->msg_send[1] || !h->msg_recv[1])
        return fail(h, "pte_shard_scan_begin: call pte_shard_set_buffers first");
    if ((h->cfg.record_flags & PTE_RECORD_INDEX_PROCESS) && h->scans_in_round >= h->cfg.max_scans_per_round)
        return fail(h, "index_process buffer full (max_scans_per_round = %lld)", (long long)h->cfg.max_scans_per_round);
    if (launch_explore(h, scan)) return 1;
    const int64_t K = h->K;
    const int even = (scan % 2 == 0) ? 1 : 0;
    const unsigned block = 256, grid = (unsigned)((K + block - 1) / block);
    time_begin(h, 1);
    hipLaunchKernelGGL(k_swap_stats, dim3(grid), dim3(block), 0, h->stream, h->dev, even, h->scans_in_round);
    boundary_active(h, even, active_out);
    if (active_out[0] || active_out[1])
        hipLaunchKernelGGL(k_boundary_pack, dim3(2), dim3(256), 0, h->stream, h->dev, (int)active_out[0], (int)active_out[1],
                           h->msg_send[0], h->msg_send[1]);
    time_end(h);
    HIP_OK(h, hipGetLastError());
    return 0;
}
int pte_shard_scan_finish(pte_engine *h, int64_t scan) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_shard_scan_finish");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t K = h->K;
    const int even = (scan % 2 == 0) ? 1 : 0;
    int32_t active[2];
    boundary_active(h, even, active);
    const unsigned block = 256, grid = (unsigned)((K + block - 1) / block);
    time_begin(h, 1);
    hipLaunchKernelGGL(k_boundary_stats_in, dim3(1), dim3(64), 0, h->stream, h->dev, (int)active[0], (int)active[1],
                       (const double *)h->msg_recv[0], (const double *)h->msg_recv[1]);
    hipLaunchKernelGGL(k_swap_decide, dim3(grid), dim3(block), 0, h->stream, h->dev, even, h->scans_in_round);
    h->slot_cur ^= 1;                                   // the decide kernel wrote the new chain -> slot map
    h->dev.slot_of_chain = h->slot_map[h->slot_cur];
    h->dev.slot_of_chain_alt = h->slot_map[h->slot_cur ^ 1];
    if (active[0] || active[1])
        hipLaunchKernelGGL(k_boundary_apply, dim3(2), dim3(256), 0, h->stream, h->dev, (const double *)h->msg_recv[0],
                           (const double *)h->msg_recv[1], h->d_napplied);
    time_end(h);
    HIP_OK(h, hipGetLastError());
    h->scans_in_round += 1;
    return 0;
}
int pte_shard_sync(pte_engine *h, int64_t *boundary_swaps_out) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_shard_sync");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    int rc = check_device_error(h);                     // synchronises the engine's stream
    time_collect(h);
    if (boundary_swaps_out) {
        HIP_OK(h, hipMemcpyAsync(boundary_swaps_out, h->d_napplied, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(h, hipStreamSynchronize(h->stream));
    }
    return rc;
}

// ---- transport behind the ABI: RCCL point-to-point on the engine's stream / in-process groups ------------
namespace {

#define NCCL_OK(h, api, call)                                                                     \
    do { ncclResult_t r_ = (call);                                                                \
         if (r_ != ncclSuccess) return fail(h, "%s failed: %s", #call, (api)->GetErrorString(r_)); } while (0)

// engine-owned message buffers {send_lo, recv_lo, send_hi, recv_hi}, d + 8 words each, unless the caller installed its own
int ensure_msg_buffers(pte_engine *h) {
    if (h->msg_send[0] && h->msg_recv[0] && h->msg_send[1] && h->msg_recv[1]) return 0;
    const size_t w = (size_t)(h->dev.sw + 8);
    if (!h->own_msg && dev_alloc(h, &h->own_msg, 4 * w)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    h->msg_send[0] = h->own_msg;         h->msg_recv[0] = h->own_msg + w;
    h->msg_send[1] = h->own_msg + 2 * w; h->msg_recv[1] = h->own_msg + 3 * w;
    return 0;
}

// One DEO scan of a chain-sharded engine, everything enqueued on the engine's stream:
//   explore -> k_swap_stats -> k_boundary_pack -> { ncclSend, ncclRecv per active side, one group }
//   -> k_boundary_stats_in -> k_swap_decide -> k_boundary_apply
int run_scans_sharded(pte_engine *h, int64_t first_scan, int64_t n_scans) {
    if (h->comm_kind != 1)
        return fail(h, "pte_run_scans on a chain-sharded engine (world_size %d) needs pte_comm_init "
                       "(or drive the engines of one process with pte_group_run_scans, or the two-phase pte_swap_begin / pte_swap_finish)", h->world);
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(h, "%s", err.c_str());
    if (ensure_msg_buffers(h)) return 1;
    const size_t words = (size_t)(h->dev.sw + 8);
    for (int64_t s = first_scan; s < first_scan + n_scans; ++s) {
        int32_t active[2];
        if (pte_shard_scan_begin(h, s, active)) return 1;
        if (active[0] || active[1]) {
            time_begin(h, 3);                                // the boundary exchange as the stream sees it: pack done .. messages landed
            NCCL_OK(h, api, api->GroupStart());
            for (int sd = 0; sd < 2; ++sd) {
                if (!active[sd]) continue;
                const int peer = sd == 0 ? h->rank - 1 : h->rank + 1;
                NCCL_OK(h, api, api->Send(h->msg_send[sd], words, ncclDouble, peer, h->nccl, h->stream));
                NCCL_OK(h, api, api->Recv(h->msg_recv[sd], words, ncclDouble, peer, h->nccl, h->stream));
            }
            NCCL_OK(h, api, api->GroupEnd());
            time_end(h);
        }
        if (pte_shard_scan_finish(h, s)) return 1;
    }
    return pte_shard_sync(h, nullptr);
}

}  // namespace

int pte_comm_allow_library_override(int32_t allow) { rccl_override_allowed() = allow ? 1 : 0; return 0; }

int pte_comm_library(char *path_out, int64_t capacity, int32_t *version_out) {
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(nullptr, "%s", err.c_str());
    if (path_out && capacity > 0) { std::snprintf(path_out, (size_t)capacity, "%s", api->where.c_str()); }
    if (version_out) { int v = 0; api->GetVersion(&v); *version_out = v; }
    return 0;
}

int pte_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return fail(nullptr, "pte_comm_unique_id: null argument");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(nullptr, "%s", err.c_str());
    static_assert(sizeof(ncclUniqueId) == PTE_COMM_ID_BYTES, "PTE_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int pte_comm_init(pte_engine *h, const uint8_t *id_bytes) {
    if (!h || !id_bytes) return 1;
    if (h->comm_kind != 0) return fail(h, "pte_comm_init: the engine already has a communicator");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(h, "%s", err.c_str());
    HIP_OK(h, hipSetDevice(h->cfg.device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    NCCL_OK(h, api, api->CommInitRank(&h->nccl, h->world, id, h->rank));
    h->comm_kind = 1;
    if (!h->d_coll && dev_alloc(h, &h->d_coll, 16)) return 1;
    if (ensure_msg_buffers(h)) return 1;
    // n_ranks_seen: measured (an all-reduce of 1 over the communicator), not the configured world size
    double one = 1.0;
    if (pte_comm_allreduce(h, &one, 1, 1)) return 1;
    h->n_ranks_seen = (int)std::llround(one);
    if (h->n_ranks_seen != h->world) return fail(h, "pte_comm_init: the communicator spans %d ranks, expected world_size %d", h->n_ranks_seen, h->world);
    return 0;
}

int pte_comm_destroy(pte_engine *h) {
    if (!h) return 1;
    if (h->comm_kind == 1 && h->nccl) {
        std::string err;
        if (RcclApi *api = rccl_api(err)) { hipSetDevice(h->cfg.device); hipStreamSynchronize(h->stream); api->CommDestroy(h->nccl); }
    }
    h->nccl = nullptr; h->comm_kind = 0;
    return 0;
}

int pte_comm_info(pte_engine *h, int32_t *kind, int32_t *n_ranks_seen, int64_t *boundary_swaps) {
    if (!h) return 1;
    if (kind) *kind = h->comm_kind;
    if (n_ranks_seen) *n_ranks_seen = h->comm_kind == 1 ? h->n_ranks_seen : 1;
    if (boundary_swaps) {
        HIP_OK(h, hipSetDevice(h->cfg.device));
        HIP_OK(h, hipMemcpyAsync(boundary_swaps, h->d_napplied, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

int pte_comm_allreduce(pte_engine *h, double *inout, int64_t n, int32_t op) {
    if (!h || !inout || n < 0 || (op != 0 && op != 1)) return 1;
    if (h->world == 1 || n == 0) return 0;
    if (h->comm_kind != 1) return fail(h, "pte_comm_allreduce: call pte_comm_init first");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(h, "%s", err.c_str());
    HIP_OK(h, hipSetDevice(h->cfg.device));
    double *buf = h->d_coll;
    const bool big = n > 16;
    if (big) HIP_OK(h, hipMalloc((void **)&buf, sizeof(double) * n));
    hipError_t e1 = hipMemcpyAsync(buf, inout, sizeof(double) * n, hipMemcpyHostToDevice, h->stream);
    ncclResult_t r = e1 == hipSuccess ? api->AllReduce(buf, buf, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, h->nccl, h->stream) : ncclSuccess;
    if (e1 == hipSuccess && r == ncclSuccess) e1 = hipMemcpyAsync(inout, buf, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    if (big) hipFree(buf);
    if (r != ncclSuccess) return fail(h, "ncclAllReduce failed: %s", api->GetErrorString(r));
    HIP_OK(h, e1); HIP_OK(h, e2);
    return 0;
}

int pte_comm_barrier(pte_engine *h) {
    if (!h) return 1;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    double z = 0.0;
    return pte_comm_allreduce(h, &z, 1, 1);
}

int pte_comm_allgather(pte_engine *h, const void *send, int64_t bytes, void *recv) {
    if (!h || !send || !recv || bytes < 0) return 1;
    if (h->world == 1) { std::memmove(recv, send, (size_t)bytes); return 0; }
    if (h->comm_kind != 1) return fail(h, "pte_comm_allgather: call pte_comm_init first");
    if (bytes == 0) return 0;
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(h, "%s", err.c_str());
    HIP_OK(h, hipSetDevice(h->cfg.device));
    unsigned char *buf = nullptr;
    HIP_OK(h, hipMalloc((void **)&buf, (size_t)bytes * (size_t)(h->world + 1)));
    unsigned char *sb = buf + (size_t)bytes * (size_t)h->world;
    hipError_t e1 = hipMemcpyAsync(sb, send, (size_t)bytes, hipMemcpyHostToDevice, h->stream);
    ncclResult_t r = e1 == hipSuccess ? api->AllGather(sb, buf, (size_t)bytes, ncclUint8, h->nccl, h->stream) : ncclSuccess;
    if (e1 == hipSuccess && r == ncclSuccess) e1 = hipMemcpyAsync(recv, buf, (size_t)bytes * (size_t)h->world, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    hipFree(buf);
    if (r != ncclSuccess) return fail(h, "ncclAllGather failed: %s", api->GetErrorString(r));
    HIP_OK(h, e1); HIP_OK(h, e2);
    return 0;
}

// G engines of one process (engines[g] = rank g): the same kernels, the messages move by stream-ordered
// device-to-device copies.  Cross-stream order per scan:
//   pack(g)   waits for copied(g-1), copied(g+1) of the previous scan   (they read g's send buffers)
//   copies(g) wait for pack(g-1) / pack(g+1) of this scan               (they read the neighbours' send buffers)
int pte_group_run_scans(pte_engine *const *hs, int32_t G, int64_t first_scan, int64_t n_scans) {
    if (!hs || G < 1 || !hs[0]) return fail(nullptr, "pte_group_run_scans: null argument");
    pte_engine *h0 = hs[0];
    for (int g = 0; g < G; ++g) {
        pte_engine *h = hs[g];
        if (!h) return fail(h0, "pte_group_run_scans: engine %d is null", g);
        if (h->world != G || h->rank != g || h->N != h0->N || h->d != h0->d)
            return fail(h0, "pte_group_run_scans: engines[%d] must be rank %d of a world of %d over the same chains (got rank %d of %d)", g, g, G, h->rank, h->world);
        if (h->comm_kind != 0) return fail(h0, "pte_group_run_scans: engine %d has an RCCL communicator; use pte_run_scans", g);
        HIP_OK(h, hipSetDevice(h->cfg.device));
        if (ensure_msg_buffers(h)) { h0->err = h->err; return 1; }
        if (!h->ev_pack) HIP_OK(h, hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming));
        if (!h->ev_copied) HIP_OK(h, hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    }
    const size_t bytes = sizeof(double) * (size_t)(h0->dev.sw + 8);
    std::vector<int32_t> act((size_t)(2 * G));
    for (int64_t s = first_scan; s < first_scan + n_scans; ++s) {
        for (int g = 0; g < G; ++g) {
            pte_engine *h = hs[g];
            HIP_OK(h, hipSetDevice(h->cfg.device));
            if (g > 0) HIP_OK(h, hipStreamWaitEvent(h->stream, hs[g - 1]->ev_copied, 0));
            if (g + 1 < G) HIP_OK(h, hipStreamWaitEvent(h->stream, hs[g + 1]->ev_copied, 0));
            if (pte_shard_scan_begin(h, s, &act[2 * g])) { if (h != h0) h0->err = h->err; return 1; }
            HIP_OK(h, hipEventRecord(h->ev_pack, h->stream));
        }
        for (int g = 0; g < G; ++g) {
            pte_engine *h = hs[g];
            HIP_OK(h, hipSetDevice(h->cfg.device));
            if (act[2 * g]) {                     // my recv_lo <- lower neighbour's send_hi
                HIP_OK(h, hipStreamWaitEvent(h->stream, hs[g - 1]->ev_pack, 0));
                HIP_OK(h, hipMemcpyAsync(h->msg_recv[0], hs[g - 1]->msg_send[1], bytes, hipMemcpyDefault, h->stream));
            }
            if (act[2 * g + 1]) {                 // my recv_hi <- upper neighbour's send_lo
                HIP_OK(h, hipStreamWaitEvent(h->stream, hs[g + 1]->ev_pack, 0));
                HIP_OK(h, hipMemcpyAsync(h->msg_recv[1], hs[g + 1]->msg_send[0], bytes, hipMemcpyDefault, h->stream));
            }
            HIP_OK(h, hipEventRecord(h->ev_copied, h->stream));
            if (pte_shard_scan_finish(h, s)) { if (h != h0) h0->err = h->err; return 1; }
        }
    }
    int rc = 0;
    for (int g = 0; g < G; ++g)
        if (pte_shard_sync(hs[g], nullptr)) { if (hs[g] != h0) h0->err = hs[g]->err; rc = 1; }
    return rc;
}

const char *pte_kernel_name(const pte_engine *h) {
    if (!h) return "";
    switch (h->cfg.explorer) {
    case PTE_EXPLORER_TOY: return "k_explore_toy";
    case PTE_EXPLORER_SLICE:
        // SliceSampler on the interpolated path: the one-wave Langevin-family kernel in its slice mode (no momentum, gradient or trial copies:
        // its sixteen-block instantiation for d > 512 holds 255 VGPRs + 7 AGPRs and does not touch scratch)
        if (h->cfg.target == PTE_TARGET_FUNNEL) return "k_explore_automala";
        switch (h->slice_impl) {
        case 1: return "k_explore_slice"; case 2: return "k_explore_slice2"; case 5: return "k_explore_slice5";
        case 7: return "k_explore_slice7";
        default:
            if (h->K > PTE_S8_TWIN_FROM) return "k_explore_slice8_lds10k";
            return (h->cfg.slice_p > PTE_S8_BD && h->cfg.slice_p <= 20 && h->cfg.slice_max_iter >= PTE_S8_BS) ? "k_explore_slice8" : "k_explore_slice8_generic";
        }
    case PTE_EXPLORER_AUTOMALA: case PTE_EXPLORER_MALA:
        // d > 512: four waves per replica (pte_automala_mw.hpp, round 6); the one-wave kernel with sixteen blocks per lane -- 250-300 spilled VGPRs,
        // "unoptimised" in rounds 1-5 -- survives in the test build as its A/B reference
        if (h->d > 512) return (h->cfg.debug_kernel & PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE) ? "k_explore_automala [test build: one wave, sixteen blocks per lane]" : "k_explore_langevin_mw";
        return "k_explore_automala";
    case PTE_EXPLORER_ISING_METROPOLIS: {
        const int64_t L = (int64_t)std::llround(std::sqrt((double)h->d));
        return (L % 32 == 0 && h->ising_impl == 0) ? "k_explore_ising_spec" : (L % 32 == 0 && h->ising_impl == 1) ? "k_explore_ising_bits" : "k_explore_ising";
    }
    default: return "";
    }
}

// The form pte_run_scans takes on this engine: "" = two launches per scan (explore, swap); otherwise the name of the ONE kernel that runs
// all the scans of a call (k_scans_*).  *resident_limit: workgroups of that kernel the device holds at once (0 when not available);
// *timed_launches / *timed_scans: fused launches and the scans inside them since the last pte_timing_reset (pte_timing_get(kernel = 4) holds
// their durations).
const char *pte_scan_loop_name(const pte_engine *hc) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h) return "";
    hipSetDevice(h->cfg.device);
    if (!fused_scans_eligible(h, 1)) return "";
    if (fused_kind(h) == 2) return h->d > 512 ? "k_scans_langevin_mw" : h->fused_wg > 1 ? "k_scans_automala_wg" : "k_scans_automala";
    return fused_slice_variant(h) == 0 ? "k_scans_slice8" : "k_scans_slice8_generic";
}
int pte_scan_loop_info(const pte_engine *hc, int64_t *resident_limit, int64_t *timed_launches, int64_t *timed_scans) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h) return 1;
    hipSetDevice(h->cfg.device);
    (void)fused_scans_eligible(h, 1);
    hipStreamSynchronize(h->stream);
    time_collect(h);
    if (resident_limit) *resident_limit = h->fused_limit < 0 ? 0 : h->fused_limit;
    if (timed_launches) *timed_launches = h->t_n[4];
    if (timed_scans) *timed_scans = h->t_scans4;
    return 0;
}

int pte_scan_loop_stats(const pte_engine *h, int64_t *fused_calls, int64_t *gate_aborts, int32_t *poisoned) {
    if (!h) return 1;
    if (fused_calls) *fused_calls = h->fused_calls;
    if (gate_aborts) *gate_aborts = h->gate_aborts;
    if (poisoned) *poisoned = h->poisoned ? 1 : 0;
    return 0;
}

namespace {
int refresh_funnel_stats(pte_engine *h) {
    const int E = h->d <= 64 ? 1 : h->d <= 128 ? 2 : h->d <= 256 ? 4 : h->d <= 512 ? 8 : 16;
    const double log3 = std::log(3.0);
    const unsigned N = (unsigned)h->K;
    langevin_refresh_funnel_stats(E, N, h->stream, h->dev, log3);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}
}  // namespace

// update_reference! + update_path_variational (src/variational/variational.jl:28-41, GaussianReference.jl:24-31): from now on
// the chains with uses[c] != 0 run InterpolatingPath(GaussianReference(mean, std), target).  mean = std = NULL deactivates.
int pte_set_variational_reference(pte_engine *h, const double *mean, const double *std_dev, int64_t dim, const int32_t *uses) {
    if (!h) return 1;
    PTE_ALIVE(h, "pte_set_variational_reference");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    EngineDev &e = h->dev;
    if (!mean || !std_dev || !uses) { e.v_use = nullptr; return 0; }
    if (h->cfg.target != PTE_TARGET_FUNNEL) return fail(h, "pte_set_variational_reference: only the interpolated (funnel) path has a replaceable reference");
    if (h->world != 1) return fail(h, "pte_set_variational_reference: single engine only");
    if (h->cfg.explorer == PTE_EXPLORER_SLICE || h->cfg.explorer2 == PTE_EXPLORER_SLICE)
        return fail(h, "pte_set_variational_reference: a GaussianReference under SliceSampler is not implemented on the device; use the reference CPU path");
    if (dim != h->d) return fail(h, "pte_set_variational_reference: expected %lld coordinates", (long long)h->d);
    const int64_t d = h->d;
    std::vector<double> buf((size_t)(5 * d));
    for (int64_t i = 0; i < d; ++i) {
        const double s = std_dev[i], s2 = s * s;
        if (!(s > 0.0) || !std::isfinite(s) || !std::isfinite(mean[i])) return fail(h, "pte_set_variational_reference: bad mean / standard deviation at coordinate %lld", (long long)i);
        buf[i] = mean[i]; buf[d + i] = s;
        buf[2 * d + i] = -0.5 * std::log(2.0 * M_PI * s2);        // gaussian_logdensity, GaussianReference.jl:46
        buf[3 * d + i] = 1.0 / (2.0 * s2);
        buf[4 * d + i] = -1.0 / s2;                                // logdensity_and_gradient(::BufferedAD{GaussianReference}), :71
    }
    HIP_OK(h, hipMemcpyAsync(h->d_vref, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->d_vuse, uses, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    e.v_mean = h->d_vref; e.v_std = h->d_vref + d; e.v_c0 = h->d_vref + 2 * d; e.v_i2 = h->d_vref + 3 * d; e.v_gf = h->d_vref + 4 * d;
    e.v_use = h->d_vuse;
    return refresh_funnel_stats(h);                                // suff3 of the current states
}

int pte_get_state(const pte_engine *hc, double *state, int64_t *chain, uint64_t *rng) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h) return 1;
    PTE_ALIVE(h, "pte_get_state");
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t N = h->K, d = h->d;
    const bool ising = h->cfg.target == PTE_TARGET_ISING;
    std::vector<uint32_t> packed;
    if (state && d > 0 && ising) {      // the Replica.state contract of the ABI stays 0.0 / 1.0 per site; the device row is bit-packed
        packed.resize((size_t)(N * h->dev.ld * 2));
        HIP_OK(h, hipMemcpyAsync(packed.data(), h->dev.x, sizeof(double) * N * h->dev.ld, hipMemcpyDeviceToHost, h->stream));
    } else if (state && d > 0)
        HIP_OK(h, hipMemcpy2DAsync(state, sizeof(double) * d, h->dev.x, sizeof(double) * h->dev.ld,
                                   sizeof(double) * d, N, hipMemcpyDeviceToHost, h->stream));
    std::vector<int32_t> ch(N);
    if (chain) HIP_OK(h, hipMemcpyAsync(ch.data(), h->dev.chain_of_slot, sizeof(int32_t) * N, hipMemcpyDeviceToHost, h->stream));
    if (rng) HIP_OK(h, hipMemcpyAsync(rng, h->dev.rng, sizeof(uint64_t) * 2 * N, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    if (chain) for (int64_t i = 0; i < N; ++i) chain[i] = ch[i];
    if (!packed.empty())
        for (int64_t r = 0; r < N; ++r) {
            const uint32_t *w = packed.data() + (size_t)(r * h->dev.ld * 2);
            for (int64_t sI = 0; sI < d; ++sI) state[r * d + sI] = (double)((w[sI >> 5] >> (sI & 31)) & 1u);
        }
    return 0;
}

int pte_set_state(pte_engine *h, const double *state, const int64_t *chain, const uint64_t *rng) {
    if (!h) return 1;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int64_t N = h->K, d = h->d;
    if (h->poisoned) {
        // the one way back after a failure inside the one-launch scan loop: every field of every replica replaced, the round's recorders
        // discarded, the hand-shake flags re-based to the epochs handed out so far ("every chain has published everything up to now")
        if ((!state && d > 0) || !chain || !rng) return fail(h, "pte_set_state: a poisoned engine needs state, chain and rng of every replica");
        HIP_OK(h, hipStreamSynchronize(h->stream));
        HIP_OK(h, hipMemsetAsync(h->dev.error, 0, 4 * sizeof(int32_t), h->stream));
        if (h->hs_flag) {
            std::vector<unsigned long long> fl((size_t)N, h->hs_epoch);
            HIP_OK(h, hipMemcpyAsync(h->hs_flag, fl.data(), sizeof(unsigned long long) * N, hipMemcpyHostToDevice, h->stream));
            HIP_OK(h, hipStreamSynchronize(h->stream));
        }
        if (reset_recorders(h)) return 1;
        h->poisoned = false;
    }
    std::vector<uint32_t> packed;
    if (state && d > 0 && h->cfg.target == PTE_TARGET_ISING) {
        packed.assign((size_t)(N * h->dev.ld * 2), 0u);
        for (int64_t r = 0; r < N; ++r) {
            uint32_t *w = packed.data() + (size_t)(r * h->dev.ld * 2);
            for (int64_t sI = 0; sI < d; ++sI) if (state[r * d + sI] != 0.0) w[sI >> 5] |= 1u << (sI & 31);
        }
        HIP_OK(h, hipMemcpyAsync(h->dev.x, packed.data(), sizeof(double) * N * h->dev.ld, hipMemcpyHostToDevice, h->stream));
    } else if (state && d > 0)
        HIP_OK(h, hipMemcpy2DAsync(h->dev.x, sizeof(double) * h->dev.ld, state, sizeof(double) * d,
                                   sizeof(double) * d, N, hipMemcpyHostToDevice, h->stream));
    std::vector<int32_t> ch(N), inv(N, -1);
    if (chain) {
        for (int64_t i = 0; i < N; ++i) {
            const int64_t cl = chain[i] - h->c0;
            if (cl < 0 || cl >= N || inv[cl] != -1) return fail(h, "pte_set_state: chain is not a permutation of the local chains");
            ch[i] = (int32_t)chain[i]; inv[cl] = (int32_t)i;
        }
        HIP_OK(h, hipMemcpyAsync(h->dev.chain_of_slot, ch.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice, h->stream));
        HIP_OK(h, hipMemcpyAsync(h->dev.slot_of_chain, inv.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice, h->stream));
    }
    if (rng) HIP_OK(h, hipMemcpyAsync(h->dev.rng, rng, sizeof(uint64_t) * 2 * N, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    if (state && d > 0 && h->cfg.target == PTE_TARGET_ISING) {   // sum_pair_products of every slot (examples/ising.jl:27-35)
        const int64_t L = (int64_t)std::llround(std::sqrt((double)d));
        std::vector<double> spp((size_t)N);
        for (int64_t r = 0; r < N; ++r) {
            const double *m = state + r * d;
            long long sum = 0;
            auto sg = [&](int64_t i, int64_t j) { return m[((i + L) % L) * L + ((j + L) % L)] != 0.0 ? 1 : -1; };
            for (int64_t i = 0; i < L; ++i) for (int64_t j = 0; j < L; ++j)
                sum += sg(i, j) * (sg(i - 1, j) + sg(i + 1, j) + sg(i, j - 1) + sg(i, j + 1));
            spp[r] = (double)(sum / 2);
        }
        HIP_OK(h, hipMemcpy(h->dev.suff, spp.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    } else if (state && d > 0) {   // refresh the swap statistic of every slot
        std::vector<double> suff(N);
        std::vector<double> row(d);
        double *tmp = nullptr;
        HIP_OK(h, hipMalloc((void **)&tmp, sizeof(double) * N * d));
        HIP_OK(h, hipMemcpy(tmp, state, sizeof(double) * N * d, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_test_sqr_norm, dim3((unsigned)N), dim3(64), 0, h->stream, tmp, N, d, h->nlu, h->dev.suff);
        hipError_t e1 = hipStreamSynchronize(h->stream);
        hipFree(tmp);
        HIP_OK(h, e1);
        if (h->cfg.target == PTE_TARGET_FUNNEL && refresh_funnel_stats(h)) return 1;   // + the target (and variational) log densities
    }
    return 0;
}

int pte_timing_reset(pte_engine *h, int enable) {
    if (!h) return 1;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    time_collect(h);
    h->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    for (int k = 0; k < 5; ++k) { h->t_ms[k] = 0.0; h->t_n[k] = 0; h->t_samples[k].clear(); }
    h->t_scans4 = 0;
    return 0;
}
int pte_timing_get_samples(const pte_engine *hc, int kernel, double *out_ms, int64_t capacity, int64_t *n_out) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h || kernel < 0 || kernel > 4 || kernel == 2 || !n_out) return 1;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    time_collect(h);
    const int64_t n = (int64_t)h->t_samples[kernel].size();
    *n_out = n;
    if (out_ms) for (int64_t i = 0; i < n && i < capacity; ++i) out_ms[i] = h->t_samples[kernel][(size_t)i];
    return 0;
}
int pte_timing_get(const pte_engine *hc, int kernel, double *total_ms, int64_t *launches) {
    pte_engine *h = const_cast<pte_engine *>(hc);
    if (!h || kernel < 0 || kernel > 4) return 1;
    if (kernel == 2) {                                  // k_init: timed once, at pte_create
        if (total_ms) *total_ms = h->init_ms < 0 ? 0.0 : h->init_ms;
        if (launches) *launches = h->init_ms < 0 ? 0 : 1;
        return 0;
    }
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    time_collect(h);
    if (total_ms) *total_ms = h->t_ms[kernel];
    if (launches) *launches = h->t_n[kernel];
    return 0;
}

// include/pte_rng_policy.h: one word per device, read by every engine of this process on that device
int pte_set_rng_policy(int32_t device, uint32_t policy) {
    if (policy & ~PTE_RNG_POLICY_VALID_MASK) return fail(nullptr, "pte_set_rng_policy: invalid policy 0x%x", policy);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "pte_set_rng_policy: no HIP device %d", device);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_rng_policy), &policy, sizeof policy);
    if (e == hipSuccess) e = (hipError_t)langevin_set_rng_policy(policy);       // the second translation unit's copy of the word
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : fail(nullptr, "pte_set_rng_policy: %s", hipGetErrorString(e));
}
int pte_get_rng_policy(int32_t device, uint32_t *policy) {
    if (!policy) return 1;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "pte_get_rng_policy: no HIP device %d", device);
    hipError_t e = hipMemcpyFromSymbol(policy, HIP_SYMBOL(g_rng_policy), sizeof *policy);
    return e == hipSuccess ? 0 : fail(nullptr, "pte_get_rng_policy: %s", hipGetErrorString(e));
}

int pte_test_rng_fill(int32_t device, uint64_t *sg, int32_t kind, int64_t n, double *out) {
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "pte_test_rng_fill: no HIP device");
    uint64_t *d_sg = nullptr; double *d_out = nullptr;
    if (hipMalloc((void **)&d_sg, 16) != hipSuccess || hipMalloc((void **)&d_out, sizeof(double) * (n ? n : 1)) != hipSuccess)
        return fail(nullptr, "pte_test_rng_fill: hipMalloc failed");
    hipMemcpy(d_sg, sg, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test_rng, dim3(1), dim3(64), 0, 0, d_sg, (int)kind, n, d_out);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(sg, d_sg, 16, hipMemcpyDeviceToHost);
    hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
    hipFree(d_sg); hipFree(d_out);
    return e == hipSuccess ? 0 : fail(nullptr, "pte_test_rng_fill: %s", hipGetErrorString(e));
}

int pte_test_quotient(int32_t device, const double *a, const double *b, int64_t n, double *out, int32_t *took_division) {
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "pte_test_quotient: no HIP device");
    if (!a || !b || !out || !took_division || n < 1) return fail(nullptr, "pte_test_quotient: bad argument");
    double *da = nullptr, *db = nullptr, *dout = nullptr; int32_t *dt = nullptr;
    if (hipMalloc((void **)&da, sizeof(double) * n) != hipSuccess || hipMalloc((void **)&db, sizeof(double) * n) != hipSuccess ||
        hipMalloc((void **)&dout, sizeof(double) * n) != hipSuccess || hipMalloc((void **)&dt, sizeof(int32_t) * n) != hipSuccess)
        return fail(nullptr, "pte_test_quotient: hipMalloc failed");
    hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test_quotient, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, n, dout, dt);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost); hipMemcpy(took_division, dt, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout); hipFree(dt);
    return e == hipSuccess ? 0 : fail(nullptr, "pte_test_quotient: %s", hipGetErrorString(e));
}

int pte_test_sqr_norm(int32_t device, const double *x, int64_t rows, int64_t d, double *out) {
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "pte_test_sqr_norm: no HIP device");
    if (d < 1 || d > 4096) return fail(nullptr, "pte_test_sqr_norm: d must be in 1..4096");
    double *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&dx, sizeof(double) * rows * d) != hipSuccess || hipMalloc((void **)&dout, sizeof(double) * rows) != hipSuccess)
        return fail(nullptr, "pte_test_sqr_norm: hipMalloc failed");
    hipMemcpy(dx, x, sizeof(double) * rows * d, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test_sqr_norm, dim3((unsigned)rows), dim3(64), 0, 0, dx, rows, d, next_pow2_log((d + 63) / 64), dout);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out, dout, sizeof(double) * rows, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dout);
    return e == hipSuccess ? 0 : fail(nullptr, "pte_test_sqr_norm: %s", hipGetErrorString(e));
}

}  // extern "C"

#if defined(PTE_PROFILE_WAVES) || defined(PTE_PROFILE_AM)
// debug builds only: out[4K] = per local chain {start, end on the 100 MHz clock, HW_ID, XCC_ID} of the last explore launch
// (PTE_PROFILE_AM: out[12K], see pte_automala.hpp)
extern "C" int pte_debug_wave_profile(pte_engine *h, double *out) {
    if (!h || !out) return 1;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(out, h->dev.on_m2 + 2 * (h->d + 1), sizeof(double) * PTE_WAVE_PROFILE_WORDS * (size_t)h->K, hipMemcpyDeviceToHost));
    return 0;
}
#endif
