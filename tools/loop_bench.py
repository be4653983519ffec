"""Scan time of G loopback chain-shards on ONE GPU (host-driven boundary exchange) vs the single engine."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import pigeons_amd as P
N, d = 1024, 1024
for G in (1, 2, 4, 8):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False), n_shards=G)
    eng = pt.shards if pt.shards is not None else pt.replicas
    eng.run_scans(1, 4)
    t = time.time(); eng.run_scans(5, 32); eng.reduce()
    dt = time.time() - t
    print("G=%d  %.3f ms/scan" % (G, dt / 32 * 1e3))
