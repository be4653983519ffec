"""ms / scan of toy_mvn(1024) AutoMALA against the number of chains: where does k_explore_langevin_mw stop being resident in one generation"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import pigeons_amd as P
_variant.apply()
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt
for N in (64, 256, 512, 640, 768, 896, 1024, 1280, 2048):
    out = []
    for flags in (0, _lib.KERNEL_TWO_LAUNCHES):           # one launch per call where the engine is eligible (N <= 1024) | explore + swap launches per scan
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, explorer=P.AutoMALA(), record=[P.round_trip, P.log_sum_ratio], n_rounds=8, show_report=False), debug_kernel=flags); e = pt.replicas
        for r in range(1, 5):
            e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
        e.run_scans(1, 2)
        t = time.perf_counter(); e.run_scans(1, 32); dt = time.perf_counter() - t
        out.append("%-22s %7.3f" % (e.scan_loop_name() or "(two launches)", dt / 32 * 1e3))
        del pt, e
    print("N=%-5d ms/scan: %s | %s" % (N, out[0], out[1]), flush=True)
