"""ToyExplorer (the i.i.d. refresh every kernel runs at the reference chain) at full occupancy: GB/s of state writes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import pigeons_amd as P
for N, d in ((8192, 4096), (8192, 1024), (1024, 1024))[:1 if os.environ.get('PTE_BENCH_TOY_ONLY_FIRST') else 3]:
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, record=[P.log_sum_ratio], n_rounds=20, show_report=False))
    e = pt.replicas
    e.run_scans(1, 8)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); e.run_scans(1, 64); best = min(best, (time.perf_counter() - t) / 64)
    print("%-28s N=%5d d=%5d  %.4f ms/scan  %7.1f GB/s of state writes (incl. the swap kernel's time)" % (os.path.basename(os.environ.get("PTE_LIB", "libpte.so")), N, d, best * 1e3, N * d * 8 / best / 1e9), flush=True)
