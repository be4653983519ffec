"""Fixed cost of one pte_run_scans call at the metric shape: wall time of run_scans(1, n) for n = 1 ... 128, least-squares a + b n
(fused scan loop, and the launch-per-scan loop with BV_TWO=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np, torch, pigeons_amd as P
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt
for two in (False, True):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, n_rounds=30, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]),
              debug_kernel=_lib.KERNEL_TWO_LAUNCHES if two else 0)
    e = pt.replicas
    e.run_scans(1, 16); adapt(pt, reduce_recorders(pt)); e.run_scans(1, 16)
    ns, ts = [1, 2, 4, 8, 16, 20, 32, 64, 128], []
    for n in ns:
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, n); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        ts.append(best * 1e6)
    A = np.vstack([np.ones(len(ns)), ns]).T
    a, b = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
    print("%-16s us per call: %s  -> %.0f us + %.1f us per scan" % (e.scan_loop_name() or "two launches", dict(zip(ns, [round(t) for t in ts])), a, b))
    for mode in (2, False, 2, False):
        e.timing_reset(mode)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
        print("   HIP events %-5s run_scans(1, 20): %s us" % (bool(mode), [round(t) for t in ts]))
    e.timing_reset(False)
