"""C3 (funnel d = 128, 1024 chains, AutoMALA): ms / scan of the three forms of pte_run_scans (k_scans_automala_wg: four chains per workgroup, LDS hand-shakes /
k_scans_automala: one chain per workgroup / launch per scan), interleaved.  BV_N = chains, BV_MVN=d: the MVN path instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import torch, pigeons_amd as P
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt
def mk(flags):
    tgt = dict(target=P.toy_mvn_target(int(os.environ["BV_MVN"]))) if os.environ.get("BV_MVN") else dict(target=P.Funnel(128), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 128))
    pt = P.PT(P.Inputs(n_chains=int(os.environ.get("BV_N", "1024")), n_rounds=30,
                       explorer=P.AutoMALA(), show_report=False, record=[P.round_trip, P.log_sum_ratio], **tgt), debug_kernel=flags)
    e = pt.replicas
    for _ in range(3):
        e.run_scans(1, 16); adapt(pt, reduce_recorders(pt))
    return pt, e
for rep in range(3):
    for flags in (0, _lib.KERNEL_SCAN_LOOP_ONE_CHAIN, _lib.KERNEL_TWO_LAUNCHES):
        pt, e = mk(flags)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(2, 64); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 64 * 1e3)
        print("C3 %-22s %.4f ms/scan" % (e.scan_loop_name() or "two launches", best))
