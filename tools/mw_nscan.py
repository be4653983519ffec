"""ms / scan of toy_mvn(1024) AutoMALA against the number of chains: where does k_explore_langevin_mw stop being resident in one generation"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import pigeons_amd as P
_variant.apply()
from pigeons_amd.pt import reduce_recorders, adapt
for N in (256, 512, 640, 768, 896, 1024, 1280, 2048):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, explorer=P.AutoMALA(), record=[P.round_trip, P.log_sum_ratio], n_rounds=8, show_report=False)); e = pt.replicas
    for r in range(1, 5):
        e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
    e.run_scans(1, 2)
    t = time.perf_counter(); e.run_scans(1, 8); dt = time.perf_counter() - t
    print("N=%-5d %8.3f ms/scan" % (N, dt / 8 * 1e3), flush=True)
    del pt, e
