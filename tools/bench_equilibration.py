"""Where do the 0.4 ms come from that separate the driver's 20-scan timed region (0.73 ms / scan) from a 256-scan one (0.715)?  ms / scan over
20 scans of the metric shape after (a) 5 warm-up scans + reduce + adaptation (what `bench.py --steps 20 --warmup 5` timed through round 5),
(b) R rounds of the PT algorithm first (is it the states' transient?  no), (c) reduce + adaptation first and the 5 warm-up scans LAST, so that
no host work stands between the warm-up and the timed region (is it the GPU coming back from idle?).  Usage: python tools/bench_equilibration.py [N d]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np, torch
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024


def timed(e, k):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, k); torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


def region(rounds, order, reps=3, idle_ms=0.0):
    out = []
    for _ in range(reps):
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=12, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
        e = pt.replicas
        for r in range(1, rounds + 1):
            e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
        if order == "warmup, adapt":
            e.run_scans(1, 5); adapt(pt, reduce_recorders(pt))
        else:
            e.run_scans(1, 5); adapt(pt, reduce_recorders(pt)); e.run_scans(1, 5)
        if idle_ms:
            time.sleep(idle_ms / 1e3)
        a = timed(e, 20); b = timed(e, 20); c = timed(e, 256)
        out.append((a, b, c))
    return out


for R, order, idle in ((0, "warmup, adapt", 0), (7, "warmup, adapt", 0), (0, "adapt, warmup", 0), (0, "adapt, warmup", 5.0), (0, "adapt, warmup", 50.0)):
    res = region(R, order, idle_ms=idle)
    print("%d rounds first, order %-14s idle %4.0f ms:  20 scans %s | 20 more %s | 256 %s  ms/scan" %
          (R, order, idle, " ".join("%.4f" % a for a, _, _ in res), " ".join("%.4f" % b for _, b, _ in res), " ".join("%.4f" % c for _, _, c in res)), flush=True)
