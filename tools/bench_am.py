"""ms/scan of the C3 shape (funnel d = 128, 1024 chains, AutoMALA) with the library named by PTE_LIB (A/B of tuning builds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import torch, pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
d, N = int(os.environ.get("BV_D", "128")), int(os.environ.get("BV_N", "1024"))
pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, n_rounds=20, explorer=P.AutoMALA(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
for r in range(1, 5):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 16); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 16 * 1e3)
print("funnel(%d) N=%d AutoMALA %-32s %.4f ms/scan" % (d, N, os.path.basename(os.environ.get("PTE_LIB", "default")), best))
