#!/usr/bin/env python3
"""bench.py -- replica-steps/sec of the explore-then-swap scan loop on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one scan = explore! over all chains + one DEO communicate! (reference
src/pt/pigeons.jl:49-52), the unit the reference's own stopwatch brackets.  Workload at N=1 is the
configuration BASELINE.json's metric is quoted on: toy_mvn_target(1024), n_chains=1024,
SliceSampler(w=10, p=20, n_passes=3), seed=1, synthetic (states drawn from split RNG streams).
Replica states are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(d, cores, sample_chains=1024, sample_scans=2):
    """Restated CPU baseline (NOT Pigeons.jl): the oracle's full-recompute SliceSampler, OpenMP
    static schedule over replicas (mirrors @threads, reference src/pt/pigeons.jl:82-85), on a
    bounded sample of the same workload."""
    import oracle as O
    pt = O.OraclePT(n_chains=sample_chains, dim=d, explorer=O.EXPLORER_SLICE, n_threads=cores,
                    record_index_process=0)
    pt.begin_round()
    pt.run_scans(1)                       # untimed: spins up the OpenMP team, first-touch of the replica buffers
    t0 = time.perf_counter()
    pt.run_scans(sample_scans)
    dt = time.perf_counter() - t0
    return {
        "value": sample_chains * sample_scans / dt, "unit": "replica-steps/s", "cores": cores, "kind": "port",
        "sample": "toy_mvn_target(%d), %d chains x %d scans, SliceSampler, oracle (full O(d) log-density "
                  "per evaluation as in SliceSampler.jl), OpenMP over replicas; restated CPU baseline, "
                  "not Pigeons.jl" % (d, sample_chains, sample_scans),
        "seconds": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU (weak scaling)")
    ap.add_argument("--explorer", default="slice", choices=["slice", "toy"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("PTE_BENCH_BACKEND", "nccl")          # nccl == RCCL on ROCm
        if os.environ.get("PTE_BENCH_SINGLE_DEVICE") == "1":           # smoke-testing N ranks on a 1-GPU box (gloo only)
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import numpy as np
    import pigeons_amd as P

    d, K, W = args.dim, args.steps, args.warmup
    n_chains = args.chains                      # chains per GPU; the ladder has n_chains * world chains
    explorer = P.SliceSampler() if args.explorer == "slice" else P.ToyExplorer()
    # The path shards by chain: rank g owns chains [g*n_chains, (g+1)*n_chains); only the
    # boundary pair of neighbouring ranks is exchanged (RCCL send/recv), no data-path collective.
    inputs = P.Inputs(target=P.toy_mvn_target(d), n_chains=n_chains * world, n_rounds=30, explorer=explorer, seed=1,
                      record=[P.round_trip, P.log_sum_ratio], show_report=False, device=local_rank)
    if world > 1:
        pt = P.PT(inputs, rank=rank, world=world, dist_device=torch.device("cuda", local_rank))
        runner = pt.shards
    else:
        pt = P.PT(inputs)
        runner = pt.replicas
    eng = pt.replicas

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    from pigeons_amd.pt import reduce_recorders, adapt
    # warmup: W scans, then one reduce + schedule adaptation (as at a round boundary)
    eng.timing_reset(True)              # the swap kernel's duration is taken during the warmup scans ...
    runner.run_scans(1, W)
    sw_ms, sw_n = eng.timing(1)
    adapt(pt, reduce_recorders(pt))

    eng.timing_reset(2)                 # ... the timed region carries HIP events around the dominant (explore) kernel only
    sync()
    t0 = time.perf_counter()
    runner.run_scans(1, K)              # exactly K explore+swap scans, synchronous at return
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ex_ms, ex_n = eng.timing(0)
    eng.timing_reset(False)

    # round-trip rate over the timed scans (RoundTripRecorder semantics: FSM reset at the reduce above)
    red = reduce_recorders(pt)
    restarts, trips = red.round_trip
    ss_sum, ss_n = red.explorer_n_steps

    total_chains = n_chains * world
    value = total_chains * K / dt
    # algorithmic HBM bytes of the dominant kernel per launch:
    #   explore (slice / iid): state read + write + rng r/w = 16 d + 32 B per replica
    bytes_per_replica = 16 * d + 32 if args.explorer == "slice" else 8 * d + 32
    alg_bytes = bytes_per_replica * n_chains
    ex_avg_ms = ex_ms / max(ex_n, 1)
    impl = os.environ.get("PTE_SLICE_IMPL", "8")
    kernel_name = {"slice": "k_explore_slice" + ("" if impl == "1" else impl), "toy": "k_explore_toy"}[args.explorer]
    traffic = None
    issue = None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC passes of this kernel at this workload
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(kernel_name)
        if tj and d == 1024 and n_chains == 1024:
            traffic = tj["fetch_bytes"] + tj["write_bytes"]
            issue = tj.get("issue")                   # what actually bounds the kernel: instruction issue of one wave per replica
    except Exception:
        traffic = None
    achieved = alg_bytes / (ex_avg_ms * 1e-3) / 1e9 if ex_avg_ms > 0 else 0.0
    out = {
        "metric": "replica-steps/sec (explore+swap), toy_mvn d=%d, n_chains=%d; round-trip rate" % (d, total_chains),
        "value": value, "unit": "replica-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "toy_mvn_target(%d), n_chains=%d per GPU x %d GPU, %s, seed=1, DEO swaps every scan"
                               % (d, n_chains, world, "SliceSampler(w=10,p=20,n_passes=3)" if args.explorer == "slice" else "ToyExplorer"),
                   "sharding": ("chains sharded over %d GPUs, boundary replicas exchanged by RCCL send/recv" % world) if world > 1 else "single GPU",
                   "boundary_swaps_rank0": getattr(runner, "n_boundary_swaps", 0)},
        "round_trip_rate": trips / K, "n_round_trips": trips, "n_tempered_restarts": restarts,
        "lp_evals_per_replica_step": float(np.sum(ss_sum) / max(K * (total_chains - 1), 1)) + 2.0 * 3 * d if args.explorer == "slice" else 0.0,
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "avg_launch_ms": ex_avg_ms, "launches": ex_n,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "swap_kernel_avg_launch_ms": sw_ms / max(sw_n, 1),
                     "instruction_issue": issue,
                     "note": "SliceSampler is bound by the instruction issue of ONE wave per replica walking a sequential "
                             "decision chain (3*d coordinate updates, ~6.5 draws each), not by HBM; see DESIGN.md sec. 5"},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.explorer == "slice":
            out["cpu_baseline"] = cpu_baseline(d, os.cpu_count() or 1)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
