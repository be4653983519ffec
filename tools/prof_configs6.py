"""Round 6: the program rocprofv3 runs for the per-config profiles (tools/prof_configs6.sh): every entry of bench.py's extra_configs -- C1, C2,
C3, the C4 shard, C4 on one GPU, the C5 shard -- prepared EXACTLY as bench.py prepares it (bench.extra_config_list / the same rounds of the
algorithm), then REPS pte_run_scans calls of the config's timed scan count.  Prints one JSON line per config saying which kernel dispatches are
the timed ones (tools/r06_configs_summary.py picks the LAST reps (fused: launches) or reps x scans (explore + swap per scan) dispatches of that
kernel out of the rocprofv3 databases).  No HIP events in the stream here: the profiler times the kernels itself."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import bench                                    # noqa: E402
import pigeons_amd as P                         # noqa: E402
from pigeons_amd.pt import reduce_recorders, adapt   # noqa: E402

REPS = int(os.environ.get("PC6_REPS", "3"))
ONLY = [k for k in os.environ.get("PC6_ONLY", "").split(",") if k]

for key, name, mk, rounds, scans, bytes_per_replica in bench.extra_config_list(P):
    if ONLY and key not in ONLY:
        continue
    inp = mk()
    pt = P.PT(inp)
    e = pt.replicas
    for r in range(1, rounds + 1):
        e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
    t = time.perf_counter()
    for _ in range(REPS):
        e.run_scans(1, scans)
    dt = time.perf_counter() - t
    fused = e.scan_loop_name()
    print("PC6 " + json.dumps({"key": key, "config": name, "n_chains": inp.n_chains, "rounds": rounds, "scans": scans, "reps": REPS,
                               "scan_loop": fused, "explore_kernel": e.kernel_name(), "bytes_per_replica_scan": bytes_per_replica,
                               "wall_ms_per_scan": dt / (REPS * scans) * 1e3,
                               "timed_dispatches": REPS if fused else REPS * scans}), flush=True)
    del pt, e
