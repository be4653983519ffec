#!/usr/bin/env python3
"""bench.py -- replica-steps/sec of the explore-then-swap scan loop on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
A "step" is one scan = explore! over all chains + one DEO communicate! (reference
src/pt/pigeons.jl:49-52), the unit the reference's own stopwatch brackets.  Workload at N=1 is the
configuration BASELINE.json's metric is quoted on: toy_mvn_target(1024), n_chains=1024,
SliceSampler(w=10, p=20, n_passes=3), seed=1, synthetic (states drawn from split RNG streams).
Replica states are resident in HBM before the timed region starts.

N > 1: one process per GPU.  Started as plain `python bench.py --gpus N` this process only LAUNCHES: it touches no
GPU, spawns N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relays rank 0's JSON
line and exits non-zero if any child fails.  Started under `python -m torch.distributed.run --nproc-per-node N` the
environment is already there and the process is a rank.  The path shards by CHAIN: rank g owns chains
[g*K, (g+1)*K); only the boundary pair of neighbouring ranks is exchanged, by RCCL send/recv that libpte itself
enqueues on the engine's stream (pte_comm_init / pte_run_scans) -- there is no data-path collective and no
torch.distributed call inside the timed loop; torch.distributed (backend nccl = RCCL) carries the 128-byte
communicator id, the barriers around the timed region and the MAX over ranks.
  --scaling weak   (default) 1024 chains per GPU, d = 1024: the metric's shape on every GPU, ladder of N*1024 chains
  --scaling strong BASELINE configs[3]: toy_mvn_target(4096), n_chains = 8192 in total, 8192/N per GPU
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0    # ... and the measured float4-copy rate the guide quotes (SURVEY.md 8(d) prices against both)
NRM_OCCUPANCY = 5              # waves per SIMD of k_explore_toy / k_init (96 VGPRs, 31 KB of LDS per 4-wave workgroup: profiles/r05_kernel_resources.txt)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (weak) / in total (strong)")
    ap.add_argument("--explorer", default="slice", choices=["slice", "toy"])
    ap.add_argument("--prepare", type=int, default=64,
                    help="untimed preparation of the synthetic input before --warmup: this many scans from the initial states, reduce + schedule "
                         "adaptation, this many scans again (the states of a run in progress: a schedule change is followed by ~20 scans "
                         "that cost 2 %% more, tools/bench_equilibration.py).  0 = the order of rounds 1-5: warm-up, then the adaptation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed hbm_kernels / extra_configs blocks")
    ap.add_argument("--same-device", action="store_true",
                    help="TEST ONLY: every rank uses HIP device 0 (the control plane then runs over gloo, and the data path needs an "
                         "RCCL stand-in that accepts two ranks on one device: $PTE_RCCL_LIB=tests/fakerccl/libfakerccl.so); "
                         "the line says so and is not a measurement")
    ap.add_argument("--timeout-s", type=float, default=1500.0,
                    help="watchdog: a rank (or the launcher) that is still running after this many seconds reports it and exits "
                         "non-zero instead of sitting in a collective for ever")
    ap.add_argument("--round-trip-rounds", type=int, default=15,
                    help="untimed leg after the timed region: rounds 1..R of the reference's round loop with schedule "
                         "adaptation; the round-trip rate is read off the last round (2^R scans).  0 = skip")
    return ap.parse_args()


def launch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks.  Nothing here may touch the GPU (a process that
    has initialised HIP must not fork / exec ranks), so neither torch nor libpte is imported in this process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus),
                LOCAL_WORLD_SIZE=str(args.gpus))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    base.setdefault("OMP_NUM_THREADS", "1")
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    # wait for all; a rank that dies would leave its peers blocked in a collective, so the first failure ends the others
    rcs = [None] * len(procs)
    t_start = time.time()
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if time.time() - t_start > args.timeout_s + 60.0:        # (the ranks' own watchdogs fire first)
            sys.stderr.write("bench.py: ranks still running after %.0f s, ending them\n" % (time.time() - t_start))
            for i, p in enumerate(procs):
                if rcs[i] is None and p.poll() is None:
                    p.kill()                                 # exactly the PIDs started above
            rcs = [p.wait() or 124 for p in procs]
            break
        if any(rc not in (None, 0) for rc in rcs):
            time.sleep(2.0)
            for i, p in enumerate(procs):
                if rcs[i] is None and p.poll() is None:
                    p.kill()                                 # exactly the PIDs started above
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    out0.seek(0)
    text = out0.read()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if any(rcs) or not lines:
        sys.stderr.write("bench.py: ranks exited with %s\n%s\n" % (rcs, text))
        return 1
    print(lines[-1])
    return 0


def physical_cores():
    """(physical cores, logical CPUs) available to this process."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        allowed = os.sched_getaffinity(0)
        cores = set()
        cpu, phys, core = None, 0, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                cpu = int(ln.split(":")[1])
            elif ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id"):
                core = int(ln.split(":")[1])
            elif not ln.strip() and cpu is not None:
                if cpu in allowed:
                    cores.add((phys, core if core is not None else cpu))
                cpu, phys, core = None, 0, None
        return (len(cores) or logical), logical
    except Exception:
        return logical, logical


def cpu_baseline(d, target_seconds=12.0):
    """Restated CPU baseline (NOT Pigeons.jl): the oracle's SliceSampler with the full O(d) log density per evaluation as in
    SliceSampler.jl, OpenMP static schedule over replicas (mirrors @threads, reference src/pt/pigeons.jl:82-85), one thread
    pinned to every physical core (OMP_PLACES=cores, set in main() before any OpenMP runtime starts), compiled for this
    machine (-O3 -march=native, -ffp-contract=off) -- on a bounded sample of the same workload."""
    import oracle as O
    phys, logical = physical_cores()
    try:
        path, flags = O.build_native(), "-O3 -march=native"
    except Exception as exc:                          # no compiler on the box: the portable build, and say so
        path, flags = None, "-O2 (native build failed: %r)" % (exc,)
    chains = max(phys * 4, 64)                          # 4 replicas per core: static schedule balanced
    pt = O.OraclePT(lib_path=path, n_chains=chains, dim=d, explorer=O.EXPLORER_SLICE, n_threads=phys, record_index_process=0)
    pt.begin_round()
    t0 = time.perf_counter()
    pt.run_scans(1)                                     # untimed for the rate: spins up the team, first touch; sizes the sample
    probe = time.perf_counter() - t0
    scans = max(1, min(64, int(target_seconds / max(probe, 1e-3))))
    t0 = time.perf_counter()
    pt.run_scans(scans)
    dt = time.perf_counter() - t0
    return {
        "value": chains * scans / dt, "unit": "replica-steps/s", "cores": phys, "kind": "port",
        "physical_cores": phys, "logical_cpus": logical, "threads": phys, "pinning": "OMP_PLACES=cores OMP_PROC_BIND=close",
        "compiler_flags": flags + " -ffp-contract=off -fopenmp",
        "per_core": chains * scans / dt / phys,
        "sample": "toy_mvn_target(%d), %d chains x %d scans, SliceSampler(w=10,p=20,n_passes=3), oracle (full O(d) log density "
                  "per evaluation as in SliceSampler.jl), OpenMP static over replicas; restated CPU baseline, not Pigeons.jl"
                  % (d, chains, scans),
        "seconds": dt,
    }


def hbm_kernels(P):
    """The kernels the HBM roofline applies to, untimed for the headline: ToyExplorer at N = 8192, d = 4096 (256 MiB of
    state).  Durations are HIP events carried by the launch itself (hipExtLaunchKernelGGL: the kernel's own begin and end, what rocprofv3's kernel trace reports) in THIS run (pte_timing_*; k_init is timed at pte_create: six constructions, the mean of the five fastest reported -- the slowest first-touches fresh memory --, their minimum beside it);
    bytes are algorithmic: k_explore_toy / k_init write 8 d + 32 B per replica, k_swap moves 96 B per replica."""
    N, d = 8192, 4096
    init_all = []
    for _ in range(6):                                   # k_init runs once per pte_create: six constructions (the first one cold, fresh allocations first-touched)
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=8, record=[P.round_trip, P.log_sum_ratio], show_report=False))
        e = pt.replicas
        init_all.append(e.timing(2)[0])
    # one of the six constructions first-touches freshly mapped memory (usually the first, sometimes a later one when the allocator hands out a
    # new region: 204 us against 77-80 in one run): the slowest one is dropped, the other five are AVERAGED -- all six stay in the line
    init_warm = sorted(init_all)[:-1]
    init_ms = sum(init_warm) / len(init_warm)
    e.run_scans(1, 4)
    e.timing_reset(True)
    e.run_scans(1, 16)
    out = {"workload": "toy_mvn_target(%d), n_chains=%d, ToyExplorer (i.i.d. refresh of every chain)" % (d, N),
           "source": "HIP events in this run", "hbm_achievable_GBps": HBM_ACHIEVABLE_GBS, "hbm_peak_GBps": HBM_PEAK_GBS}
    for name, ms, nbytes in (("k_explore_toy", e.timing(0)[0] / max(e.timing(0)[1], 1), (8 * d + 32) * N),
                             ("k_init", init_ms, (8 * d + 32) * N),
                             ("k_swap", e.timing(1)[0] / max(e.timing(1)[1], 1), 96 * N)):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[name] = {"bytes_per_launch": nbytes, "avg_launch_us": ms * 1e3, "GBps": gbs,
                     "frac_of_6.29TBps": gbs / HBM_ACHIEVABLE_GBS, "frac_of_8TBps": gbs / HBM_PEAK_GBS}
    # the two floors of k_explore_toy: HBM -- the bytes at the achievable rate -- and ISSUE -- the model fitted
    # to tools/ubench/normals_dev.hip, T(k) = 27 + 5 k us for k rows sharing a SIMD: a row costs its SIMD 5 us of issue slots (8 rows: 40 us, at the
    # HBM floor) and 27 us is one wave's dependency chain, paid once per GENERATION of co-resident waves (occupancy_waves_per_simd of them at a time)
    rows_per_simd = N / 1024.0
    out["k_explore_toy"]["floors"] = {"hbm_us_at_6.29TBps": (8 * d + 32) * N / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6, "hbm_us_at_8TBps": (8 * d + 32) * N / (HBM_PEAK_GBS * 1e9) * 1e6,
                                      "rows_per_simd": rows_per_simd, "issue_us_per_row": 5.0, "issue_us": 5.0 * rows_per_simd,
                                      "dependency_chain_us_per_generation": 27.0, "occupancy_waves_per_simd": NRM_OCCUPANCY,
                                      "model": "T(k) = 27 + 5 k us for k rows of d = 4096 sharing a SIMD (DESIGN.md 4.1, tools/ubench/normals_dev.hip)"}
    out["k_init"]["launch_us_of_6_constructions"] = [m * 1e3 for m in init_all]      # avg_launch_us is the MEAN of the five fastest (the slowest first-touches fresh memory) ...
    out["k_init"]["min_launch_us"] = min(init_all) * 1e3                              # ... the minimum is reported separately, not as the average
    out["k_init"]["GBps_at_min_launch"] = (8 * d + 32) * N / (min(init_all) * 1e-3) / 1e9
    e.timing_reset(False)
    return out


def invariance_check(P, d, total_chains, explorer_name, rank, world, local_rank, dist):
    """The reference's parallelism invariance (docs/src/distributed.md:37-58) on THIS node and transport, before anything is timed: a short
    seeded run (rounds 1..3 = 14 scans with schedule adaptation) of the ladder cut over the `world` ranks against the same run on ONE
    engine (rank 0 holds it), index process / swap recorders / schedule / final states bit for bit.  At most 1024 chains of the ladder's
    dimension, so that the check costs seconds.  Returns True / False on every rank."""
    import numpy as np, torch
    from pigeons_amd.pt import next_round, run_one_round, adapt
    n = min(total_chains, 128 * world)
    n -= n % world
    expl = P.SliceSampler() if explorer_name == "slice" else P.ToyExplorer()
    mk = lambda: P.Inputs(target=P.toy_mvn_target(d), n_chains=n, n_rounds=3, explorer=expl, seed=7,
                          record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False, device=local_rank)
    def agree(flag_ok):
        """MIN over ranks of a local ok flag: every rank learns of a failure anywhere BEFORE it enters the next step's collectives"""
        flag = torch.tensor([1 if flag_ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    # (1) the sharded run, every rank in lock step; a rank that fails says so at the next agreement point and all stop together
    #     (libpte's RCCL exchanges inside run_one_round need every rank: nobody may step alone)
    pt, sharded, err = None, [], None
    try:
        pt = P.PT(mk(), rank=rank, world=world)
    except Exception as exc:
        err = exc
    ok = agree(err is None)
    for _ in range(3):
        if not ok:
            break
        try:
            next_round(pt); red = run_one_round(pt); adapt(pt, red)
            sharded.append((red.index_process.copy(), red.round_trip, red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(),
                            np.array(pt.shared.tempering.schedule.grids).copy()))
        except Exception as exc:
            err = exc
        ok = agree(err is None)
    states = None
    if ok:
        try:
            states = pt.shards.states()                 # (an all-gather inside libpte: still in lock step)
        except Exception as exc:
            err = exc
        ok = agree(err is None)
    if pt is not None:
        try:
            pt.replicas.comm_destroy()
        except Exception as exc:
            err = err or exc
    # (2) rank 0's single-engine run and the comparisons: only AFTER every collective of the sharded run is done, so an exception here
    #     cannot leave the peers waiting inside a collective (they go straight to the final agreement)
    if ok and rank == 0:
        try:
            one = P.PT(mk())
            for r in range(3):
                next_round(one); ra = run_one_round(one); adapt(one, ra)
                ip, rt, sw, ls, gr = sharded[r]
                ok = ok and np.array_equal(ra.index_process, ip) and ra.round_trip == rt and np.array_equal(ra.swap_acceptance_pr[0], sw) \
                    and np.array_equal(ra.log_sum_ratio[0], ls) and np.array_equal(one.shared.tempering.schedule.grids, gr)
            xa, ca, ga = one.replicas.states()
            ok = ok and np.array_equal(states[0], xa) and np.array_equal(states[1], ca) and np.array_equal(states[2], ga)
        except Exception as exc:
            err = exc
            ok = False
    if err is not None:
        sys.stderr.write("bench.py rank %d: invariance check failed to run: %r\n" % (rank, err))
    return agree(ok)


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X datasheet, FP64 vector: 256 CUs x 128 flop / clk x 2.4 GHz; a wave64 FP64 op occupies its SIMD's VALU for 4 cycles


def extra_config_list(P):
    """The other BASELINE configs at their per-GPU shapes + the 1-GPU anchor of the strong-scaling clause:
    (key, description, Inputs factory, preparation rounds, timed scans, algorithmic HBM bytes per replica and scan).
    Bytes: explore 16 d + 32 (state in / out + rng) and swap 96; Ising: the 8 KiB bit-packed lattice in / out + 128."""
    rec = [P.round_trip, P.log_sum_ratio]
    mvn = lambda d: 16 * d + 32 + 96
    return [
        ("C1", "C1 toy_mvn_target(2), n_chains=10, SliceSampler (the reference's quickstart: launch bound)", lambda: P.Inputs(target=P.toy_mvn_target(2), n_chains=10, explorer=P.SliceSampler(), record=rec, n_rounds=8, show_report=False), 5, 256, mvn(2)),
        ("C2", "C2 toy_mvn_target(1024), n_chains=256, SliceSampler", lambda: P.Inputs(target=P.toy_mvn_target(1024), n_chains=256, explorer=P.SliceSampler(), record=rec, n_rounds=8, show_report=False), 3, 16, mvn(1024)),
        ("C3", "C3 funnel d=128, n_chains=1024, AutoMALA", lambda: P.Inputs(target=P.Funnel(128), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 128), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False), 6, 128, mvn(128)),
        ("C4_shard", "C4 shard: toy_mvn_target(4096), 1024 of 8192 chains, SliceSampler", lambda: P.Inputs(target=P.toy_mvn_target(4096), n_chains=1024, explorer=P.SliceSampler(), record=rec, n_rounds=8, show_report=False), 2, 8, mvn(4096)),
        ("C4_one_gpu", "C4 on ONE GPU (strong-scaling anchor): toy_mvn_target(4096), n_chains=8192, SliceSampler", lambda: P.Inputs(target=P.toy_mvn_target(4096), n_chains=8192, explorer=P.SliceSampler(), record=rec, n_rounds=8, show_report=False), 1, 4, mvn(4096)),
        ("C5_shard", "C5 shard: Ising 256x256, 512 of 4096 chains, IsingMetropolis(3 sweeps)", lambda: P.Inputs(target=P.IsingLogPotential(1.0, 256), n_chains=512, record=rec, n_rounds=8, show_report=False), 2, 8, 2 * 8192 + 128),
    ]


def run_extra_config(P, cfg, static=None, repeats=1):
    """One entry of extra_configs: prepared as a run in progress -- rounds 1 .. r of the algorithm itself (2, 4, ... 2^r scans, reduce +
    adaptation after each: schedule; AutoMALA: step size and preconditioner) -- then `scans` scans timed by the wall clock around pte_run_scans
    (ms_per_scan, no HIP events in the stream) and the same scans once more with HIP events on every kernel launch (hipExtLaunchKernelGGL:
    the kernels' own begin / end): the kernel time per scan behind `roofline`.  `static`: this config's entry of profiles/r06_configs.json
    (rocprofv3 PMC passes of tools/prof_configs6.py -- the same preparation, the same scans: counters cannot be read inside an unprofiled run)."""
    import torch
    from pigeons_amd.pt import reduce_recorders, adapt
    key, name, mk, rounds, scans, bytes_per_replica = cfg
    inp = mk()
    pt = P.PT(inp)
    e = pt.replicas
    for r in range(1, rounds + 1):
        e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
    torch.cuda.synchronize()
    t = time.perf_counter(); e.run_scans(1, scans); torch.cuda.synchronize(); dt = time.perf_counter() - t
    fused = e.scan_loop_name()
    k_ms = []
    for _ in range(max(repeats, 1)):
        e.timing_reset(True)
        e.run_scans(1, scans)
        if fused:
            k_ms.append((e.timing(4)[0] / scans, 0.0))
        else:
            k_ms.append((e.timing(0)[0] / scans, e.timing(1)[0] / scans))
        e.timing_reset(False)
    ex_ms = sum(a for a, _ in k_ms) / len(k_ms); sw_ms = sum(b for _, b in k_ms) / len(k_ms)
    kernel_ms = ex_ms + sw_ms
    alg = bytes_per_replica * inp.n_chains
    gbs = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    roof = {"bound": "hbm", "kernel": fused or e.kernel_name(), "source": "HIP events on the kernel launches of this run (%d scans%s)" % (scans, ", one launch" if fused else ", explore + swap per scan"),
            "kernel_ms_per_scan": kernel_ms, "explore_kernel_ms_per_scan": None if fused else ex_ms, "swap_kernel_ms_per_scan": None if fused else sw_ms,
            "algorithmic_bytes_per_replica_scan": bytes_per_replica, "algorithmic_bytes_per_scan": alg,
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS, "frac_of_6.29TBps": gbs / HBM_ACHIEVABLE_GBS}
    if static:
        # executed FP64 flop/s = flops per scan the PMC pass counted at this shape / the kernel time per scan measured HERE
        fl = static.get("fp64_flops_per_scan")
        roof.update({"limited_by": static.get("limited_by"), "static_source": static.get("source"),
                     "valu_issue_frac": static.get("valu_issue_frac"), "cycles_per_instruction": static.get("cycles_per_instruction"),
                     "sq_wait_any_frac": static.get("sq_wait_any_frac"),
                     "fp64_flops_executed_per_scan": fl,
                     "fp64_executed_TFLOPs": (fl / (kernel_ms * 1e-3) / 1e12) if (fl and kernel_ms > 0) else None,
                     "fp64_vector_peak_TFLOPs": FP64_VECTOR_PEAK_TFLOPS,
                     "fp64_frac_of_vector_peak": (fl / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if (fl and kernel_ms > 0) else None,
                     "traffic": static.get("traffic_bytes_per_scan"), "traffic_over_algorithmic": (static.get("traffic_bytes_per_scan") / alg) if static.get("traffic_bytes_per_scan") else None})
    out = {"config": name, "key": key, "kernel": e.kernel_name(), "scan_loop": fused or "two launches per scan", "ms_per_scan": dt / scans * 1e3, "replica_steps_per_s": inp.n_chains * scans / dt,
           "preparation": "rounds 1..%d of the algorithm (%d scans, adapted after each round); then %d timed scans" % (rounds, 2 ** (rounds + 1) - 2, scans),
           "chains_per_gpu": inp.n_chains, "waves_per_simd": inp.n_chains / 1024.0, "roofline": roof}
    del pt, e
    return out


def extra_configs(P):
    """ms / scan (explore + swap, wall clock around pte_run_scans, states resident) of the other BASELINE configs at their per-GPU
    shapes, and the 1-GPU anchor of the strong-scaling clause -- untimed for the headline, so that every config has a driver-visible
    number and a `roofline` object of its own.  A handful of scans each, prepared as a run in progress (run_extra_config).  What a fresh
    engine costs is another regime, and C3's cost moves with the adapted step size and preconditioner (tools/diag_regimes.py: from zeros with the
    untuned step size 0.23-0.24 ms per scan, rounds 3-8 0.19-0.23, adapted every 16 scans 0.17-0.20): the line times round 7 (128 scans) after rounds 1-6."""
    try:
        static = json.load(open(os.path.join(ROOT, "profiles", "r06_configs.json")))
    except Exception:
        static = {}
    return [run_extra_config(P, cfg, static.get(cfg[0])) for cfg in extra_config_list(P)]


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")

    def _watchdog():
        sys.stderr.write("bench.py rank %d: still running after %.0f s (a peer died or a collective hangs); giving up\n" % (rank, args.timeout_s))
        sys.stderr.flush()
        os._exit(124)
    import threading
    wd = threading.Timer(args.timeout_s, _watchdog)
    wd.daemon = True
    wd.start()
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # the CPU baseline leg pins one OpenMP thread per physical core; libgomp reads this when it is first mapped
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        os.environ.setdefault("OMP_PLACES", "cores")
        os.environ.setdefault("OMP_PROC_BIND", "close")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = "gloo" if args.same_device else os.environ.get("PTE_BENCH_BACKEND", "nccl")   # nccl == RCCL on ROCm (control plane only)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import numpy as np
    import pigeons_amd as P
    from pigeons_amd.pt import reduce_recorders, adapt, next_round, run_one_round
    from pigeons_amd import engine as _engine
    if args.same_device:
        _engine.comm_allow_library_override(True)        # the test-only RCCL stand-in ($PTE_RCCL_LIB); a real run never opts in, and a stale variable then fails loudly

    K, W = args.steps, args.warmup
    if args.scaling == "strong":                # BASELINE configs[3]: d = 4096, 8192 chains in total
        d = args.dim or 4096
        total_chains = args.chains or 8192
        if total_chains % world:
            raise SystemExit("bench.py: %d chains do not divide over %d GPUs" % (total_chains, world))
        n_chains = total_chains // world
    else:                                       # the metric's shape on every GPU
        d = args.dim or 1024
        n_chains = args.chains or 1024
        total_chains = n_chains * world
    explorer = P.SliceSampler() if args.explorer == "slice" else P.ToyExplorer()
    rt_rounds = max(args.round_trip_rounds, 0)
    inputs = P.Inputs(target=P.toy_mvn_target(d), n_chains=total_chains, n_rounds=max(rt_rounds, 1), explorer=explorer, seed=1,     # (n_rounds only sizes buffers that this run does not record)
                      record=[P.round_trip, P.log_sum_ratio], show_report=False, device=local_rank)
    transport = "single GPU"
    if world > 1:
        # RcclShard: the communicator id goes over torch.distributed once, the data path is RCCL inside libpte.  If that cannot be
        # set up on ANY rank (no RCCL to map, communicator creation refused) every rank falls back -- together, agreed by an
        # all-reduce, and said in the line -- to the host-driven exchange over torch.distributed point-to-point.
        err = ""
        try:
            pt = P.PT(inputs, rank=rank, world=world)
        except Exception as exc:
            err = "%s: %s" % (type(exc).__name__, exc)
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            transport = "RCCL send/recv enqueued by libpte (pte_comm_init / pte_run_scans)"
        else:
            sys.stderr.write("bench.py rank %d: RCCL transport inside libpte unavailable (%s); falling back to the host-driven exchange\n" % (rank, err or "another rank failed"))
            pt = P.PT(inputs, rank=rank, world=world, transport="host", dist_device=torch.device("cuda", local_rank))
            transport = "FALLBACK: host-driven two-phase exchange over torch.distributed (libpte RCCL transport failed: %s)" % (err or "on another rank")
        runner = pt.shards
    else:
        pt = P.PT(inputs)
        runner = pt.replicas
    eng = pt.replicas

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    transport_library = None
    invariant = None
    if world > 1 and "FALLBACK" not in transport:
        path, ver = _engine.comm_library()               # the file ncclSend / ncclRecv really come from, and what it says it is
        transport_library = {"path": path, "nccl_version": ver}
        invariant = invariance_check(P, d, total_chains, args.explorer, rank, world, local_rank, dist)

    # warmup: W scans (first RCCL transfers open their connections here: at least two even scans per boundary whatever --warmup says),
    # then one reduce + schedule adaptation
    W_run = max(W, 4) if world > 1 else W
    prep = max(args.prepare, 0)
    if prep:                            # the synthetic input = the replicas of a run in progress (untimed, declared in config.preparation): the reduce +
        runner.run_scans(1, prep)       # adaptation every round ends with happens HERE, and the states settle under the adapted ladder, so
        adapt(pt, reduce_recorders(pt)) # that the K timed scans are scans of the steady loop and not the ~20-scan transient after a schedule
        runner.run_scans(1, prep)       # change (+0.3 ms in total at the metric shape whatever ran before: tools/bench_equilibration.py)
    eng.timing_reset(True)              # the swap kernels' duration is taken during the warmup scans ...
    runner.run_scans(1, W_run)
    sw_ms, sw_n = eng.timing(1)
    red = reduce_recorders(pt)          # (empties the recorders: lp_evals below counts the passes from here on)
    if not prep:
        adapt(pt, red)

    eng.timing_reset(2)                 # ... the timed region carries HIP events around the dominant (explore) kernel only
    sync()
    t0 = time.perf_counter()
    runner.run_scans(1, K)              # exactly K explore+swap scans, synchronous at return
    sync()
    dt_local = time.perf_counter() - t0
    dt, per_rank_ms = dt_local, [dt_local / K * 1e3]
    if dist is not None:
        t = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        per_rank_ms = [float(x.item()) / K * 1e3 for x in ts]
        dt = max(float(x.item()) for x in ts)
    # the dominant kernel of the timed region: the explore kernel (one launch per scan) -- or, where pte_run_scans runs as ONE kernel
    # (pte_scan_loop_name: explore + pairwise swap hand-shakes for all K scans), that kernel: one launch holding K scans
    scan_loop = eng.scan_loop_name() if world == 1 else ""
    tkind = 4 if scan_loop else 0
    ex_ms, ex_n = eng.timing(tkind)
    samples = np.sort(eng.timing_samples(tkind))
    scans_timed = eng.scan_loop_info()[2] if scan_loop else ex_n
    eng.timing_reset(False)
    # the same K scans once more WITHOUT HIP events in the stream (an event pair costs stream time per launch): the cross-check of
    # what the instrumentation costs the timed region above; `value` stays the instrumented, contract-timed pass
    sync()
    t0 = time.perf_counter()
    runner.run_scans(1, K)
    sync()
    dt_noev = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt_noev], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_noev = float(t.item())
    # the driver's --steps 20 makes the timed region ~17 ms: when it is shorter than 100 ms the same loop runs once more for 256
    # scans, bracketed the same way, and the line carries both (`value` stays the requested-K pass)
    long_run = None
    if dt < 0.1 and K < 256:
        K2 = 256
        sync()
        t0 = time.perf_counter()
        runner.run_scans(1, K2)
        sync()
        dt2 = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        long_run = {"steps": K2, "ms_per_step": dt2 / K2 * 1e3, "value": total_chains * K2 / dt2, "hip_events": False,
                    "note": "the requested --steps make a timed region under 100 ms; the same scan loop for 256 scans, same barriers"}
    scans_recorded = 2 * K + (256 if long_run else 0)
    # the boundary exchange as the engine's stream sees it (HIP events around ncclGroupStart .. ncclGroupEnd, one sample per even scan):
    # the first thing to read on a real multi-GPU node -- the 8-32 KiB message against the 0.8 / 3 ms kernel
    boundary_exchange = None
    if world > 1 and "RCCL" in transport and "FALLBACK" not in transport:
        eng.timing_reset(True)
        runner.run_scans(1, 16)
        bs = np.sort(eng.timing_samples(3))
        eng.timing_reset(False)
        scans_recorded += 16
        if len(bs):
            boundary_exchange = {"samples": int(len(bs)), "us_min_median_max": [float(bs[0]) * 1e3, float(bs[len(bs) // 2]) * 1e3, float(bs[-1]) * 1e3],
                                 "message_bytes_per_side": int(8 * (d + 8)),
                                 "note": "HIP events on the engine's stream around the grouped ncclSend / ncclRecv of the ranks' boundary replicas (rank 0's view)"}
    red = reduce_recorders(pt)
    ss_sum, ss_n = red.explorer_n_steps
    boundary = [int(getattr(runner, "n_boundary_swaps", 0))]
    ranks_seen = int(getattr(runner, "n_ranks_seen", 1))
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, boundary[0])
        boundary = [int(b) for b in gathered]

    # round-trip half of the metric (src/recorders/RoundTripRecorder.jl:23): untimed, the reference's own
    # round loop -- rounds 1..R with schedule adaptation after each -- the rate is n_round_trips(last round) / 2^R
    rt = None
    if rt_rounds > 0:          # keep the untimed leg near 90 s whatever the workload: rounds 1..R run 2^(R+1) - 2 scans (same R on every rank: dt is the max over ranks)
        import math
        rt_rounds = max(1, min(rt_rounds, int(math.floor(math.log2(90.0 / max(dt / K, 1e-6)))) - 1))
    if rt_rounds > 0 and args.explorer == "slice":
        t1 = time.perf_counter()
        pt.shared.iterators.round = 0
        pt.inputs.n_rounds = rt_rounds
        last = None
        while next_round(pt):
            last = run_one_round(pt)
            adapt(pt, last)
        restarts, trips = last.round_trip
        n_scans = 2 ** rt_rounds
        rt = {"rounds": rt_rounds, "scans_in_last_round": n_scans, "n_round_trips": int(trips), "n_tempered_restarts": int(restarts),
              "round_trip_rate": trips / n_scans, "global_barrier": float(P.global_barrier(pt)),
              "seconds": time.perf_counter() - t1,
              "note": "untimed leg after the timed region; last round of an R-round run with schedule adaptation, as the reference reports it"}

    value = total_chains * K / dt
    # algorithmic HBM bytes of the dominant kernel per launch:
    #   explore (slice / iid): state read + write + rng r/w = 16 d + 32 B per replica; composite explore+swap: 24 d + 128
    bytes_per_replica = 16 * d + 32 if args.explorer == "slice" else 8 * d + 32
    scans_per_launch = max(scans_timed, 1) // max(ex_n, 1) if scan_loop else 1
    if scan_loop:
        bytes_per_replica += 96                     # the fused kernel also does the swap: 96 B per replica and scan (it reads the statistic, never the state)
    alg_bytes = bytes_per_replica * n_chains * scans_per_launch
    ex_avg_ms = ex_ms / max(ex_n, 1)
    kernel_name = scan_loop or eng.kernel_name()   # reported by the library (pte_scan_loop_name / pte_kernel_name), not guessed
    explore_kernel_name = eng.kernel_name()
    traffic = None
    traffic_source = None
    issue = None
    try:   # HBM bytes per launch and the SQ instruction counters: STATIC, from the committed rocprofv3 PMC passes of this kernel at
           # this workload (PMC passes cannot run inside an unprofiled bench; the line names the file they come from)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(kernel_name)
        if tj and d == 1024 and n_chains == 1024:
            traffic = (tj["fetch_bytes"] + tj["write_bytes"]) * (scans_per_launch if tj.get("per_scan") else 1)      # per launch, like `achieved`
            traffic_source = "static: " + tj.get("source", "profiles/traffic.json")
            issue = tj.get("issue")                   # what actually bounds the kernel: instruction issue of one wave per replica
            if issue:
                # issue roofline of ONE wave per SIMD: every instruction of a lone wave costs >= 4.44 cycles (tools/ubench/issue_floor.hip,
                # profiles/r03_issue_floor.txt); frac = instructions x floor / wave cycles = the share of the wave's life that is issue
                floor = 4.44
                issue = dict(issue, source="static: " + tj.get("source", "profiles/traffic.json"), floor_cycles_per_instruction=floor,
                             frac_of_issue_floor=issue["instructions_per_wave"] * floor / issue["wave_cycles"])
    except Exception:
        traffic = None
    achieved = alg_bytes / (ex_avg_ms * 1e-3) / 1e9 if ex_avg_ms > 0 else 0.0
    lp_evals = float(np.sum(ss_sum) / max(scans_recorded * (total_chains - 1), 1)) + 2.0 * 3 * d if args.explorer == "slice" else 0.0   # (the recorders cover every pass since the adaptation)
    composite_bytes = (24 * d + 128) * value            # B/s over the whole job
    out = {
        "metric": "replica-steps/sec (explore+swap), toy_mvn d=%d, n_chains=%d; round-trip rate" % (d, total_chains),
        "value": value, "unit": "replica-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "ms_per_step_without_hip_events": dt_noev / K * 1e3, "long_run": long_run,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "toy_mvn_target(%d), n_chains=%d per GPU x %d GPU, %s, seed=1, DEO swaps every scan"
                               % (d, n_chains, world, "SliceSampler(w=10,p=20,n_passes=3)" if args.explorer == "slice" else "ToyExplorer"),
                   "sharding": ("chains sharded over %d GPUs, boundary replicas only; transport: %s" % (world, transport)) if world > 1 else "single GPU",
                   "n_ranks_seen": ranks_seen, "boundary_swaps_per_rank": boundary, "ms_per_step_per_rank": per_rank_ms,
                   "chains_per_gpu": n_chains, "waves_per_simd": n_chains / 1024.0,      # one wave per replica, 1024 SIMDs per GPU
                   "preparation": ("untimed, before --warmup: %d scans from the initial states, reduce + schedule adaptation, %d scans under the adapted "
                                   "schedule; then %d warm-up scans, then the %d timed scans (no adaptation in between)" % (prep, prep, W_run, K)) if prep else
                                  "none: %d warm-up scans from the initial states, reduce + schedule adaptation, then the %d timed scans" % (W_run, K),
                   "transport_library": transport_library, "boundary_exchange": boundary_exchange, "parallelism_invariant": invariant,
                   "env_overrides": {k: os.environ[k] for k in ("PTE_RCCL_LIB", "PTE_BENCH_BACKEND") if os.environ.get(k)},      # ($PTE_LIB is not read any more: the bench runs the in-tree library)
                   **({"same_device_test_run": "every rank on HIP device 0 with an RCCL stand-in ($PTE_RCCL_LIB=%s): exercises the "
                       "multi-rank code path, NOT a measurement" % os.environ.get("PTE_RCCL_LIB", "")} if args.same_device else {})},
        "round_trip_rate": rt["round_trip_rate"] if rt else None, "n_round_trips": rt["n_round_trips"] if rt else None,
        "n_tempered_restarts": rt["n_tempered_restarts"] if rt else None, "round_trip": rt,
        "lp_evals_per_replica_step": lp_evals,
        "roofline": {"bound": "hbm", "limited_by": "instruction_issue" if args.explorer == "slice" else "hbm", "kernel": kernel_name,
                     # the contract's roofline: bound = "hbm" (no MFMA on this path), achieved = algorithmic bytes / launch duration, frac =
                     # achieved / 8 TB/s -- well under 1 % for the SliceSampler kernel BY CONSTRUCTION: what limits it is the
                     # instruction issue of ONE wave per replica (limited_by, frac_of_issue_floor)
                     "frac_of_issue_floor": (issue or {}).get("frac_of_issue_floor"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": ex_avg_ms, "launches": ex_n,
                     "scans_per_launch": scans_per_launch, "avg_launch_ms_per_scan": ex_avg_ms / scans_per_launch, "explore_kernel": explore_kernel_name,
                     "scan_loop": ("one launch per pte_run_scans: workgroup c holds chain c, pairwise release / acquire swap hand-shakes (no launch "
                                   "boundary, no grid barrier per scan)") if scan_loop else "two launches per scan (explore, swap)",
                     "launch_ms_min_median_max": [float(samples[0]), float(samples[len(samples) // 2]), float(samples[-1])] if len(samples) else None,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "swap_kernel_avg_launch_ms": (sw_ms / sw_n) if sw_n else None,      # (None: the swap is inside the fused scan loop)
                     "composite_explore_plus_swap": {
                         "bytes_per_replica_step": 24 * d + 128, "achieved_GBps": composite_bytes / 1e9,
                         "frac_of_6.29TBps": composite_bytes / 1e9 / HBM_ACHIEVABLE_GBS / world,
                         "frac_of_8TBps": composite_bytes / 1e9 / HBM_PEAK_GBS / world},
                     "fp64_reference_equivalent_flops": 2.0 * d * lp_evals * value,
                     "instruction_issue": issue,
                     "note": "SliceSampler is bound by the instruction issue of ONE wave per replica walking a sequential "
                             "decision chain (3*d coordinate updates, ~6.5 draws each), not by HBM; see DESIGN.md sec. 5"},
    }
    if rank == 0 and world == 1 and not args.no_extra:
        del pt, runner, eng                              # (frees the metric engine's HBM before the 256 MiB one)
        if prep:
            # the order of rounds 1-4 next to the prepared one, in the same line -- a fresh engine, W warm-up scans from the initial
            # states, reduce + schedule adaptation, then the K timed scans (they then hold the ~20-scan transient after a schedule change)
            pt0 = P.PT(inputs); e0 = pt0.replicas
            e0.run_scans(1, W); adapt(pt0, reduce_recorders(pt0))
            torch.cuda.synchronize(); t0 = time.perf_counter(); e0.run_scans(1, K); torch.cuda.synchronize(); dt0 = time.perf_counter() - t0
            out["value_unprepared"] = {"value": total_chains * K / dt0, "ms_per_step": dt0 / K * 1e3,
                                       "preparation": "none (--prepare 0, the order of rounds 1-4): %d warm-up scans from the initial states, reduce + schedule adaptation, then the %d timed scans" % (W, K)}
            del pt0, e0
        out["hbm_kernels"] = hbm_kernels(P)
        out["extra_configs"] = extra_configs(P)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.explorer == "slice":
            out["cpu_baseline"] = cpu_baseline(d)
            cb = out["cpu_baseline"]
            if cb and cb.get("value"):
                # BASELINE.md publishes no number for this metric, so `vs_baseline` stays null (the contract).  The ratio to the restated CPU
                # path timed in this run is a separate, fully labelled object: ONE GPU against the `cores` host cores this box grants the
                # process (the driver's box: 1) running the oracle's O(d)-per-evaluation port -- a reported baseline, neither a speed-up over
                # Pigeons.jl on a realistic host nor a target
                out["vs_restated_cpu_port"] = {"ratio": value / cb["value"], "gpus": world, "cores": cb.get("cores"), "kind": cb.get("kind"),
                                               "per_core_replica_steps_per_s": cb.get("per_core"),
                                               "note": "value / cpu_baseline.value; cpu_baseline is the restated CPU path (kind = port) on %s host core(s), "
                                                       "not Pigeons.jl and not a published number" % cb.get("cores")}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    wd.cancel()
    if dist is not None:
        # orderly teardown, every rank at the same point: the engine's communicator first, then torch's; nothing in it can
        # change the result any more, so a teardown hiccup must not turn a measured run into a failed one
        try:
            dist.barrier()
            if hasattr(eng, "comm_destroy"):
                eng.comm_destroy()
            dist.barrier()
            dist.destroy_process_group()
        except Exception as exc:
            sys.stderr.write("bench.py rank %d: teardown: %r\n" % (rank, exc))
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
