"""ToyExplorer state-write rate as a function of the number of chains (d = 4096): launch / tail effects vs steady state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import pigeons_amd as P
d = int(os.environ.get("BT_D", "4096"))
for N in (1024, 2048, 4096, 8192, 16384, 32768):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, record=[P.log_sum_ratio], n_rounds=20, show_report=False))
    e = pt.replicas
    e.run_scans(1, 8)
    e.timing_reset(True)
    e.run_scans(1, 32)
    ms, n = e.timing(0)
    print("N=%6d d=%d  explore kernel %.4f ms  %7.1f GB/s of state writes (HIP events around the kernel)" % (N, d, ms / n, N * d * 8 / (ms / n) / 1e6), flush=True)
    del pt, e
