"""ms/scan of the C5 shard shape (Ising 256x256, 512 chains, IsingMetropolis) with the library named by PTE_LIB (A/B of tuning builds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import torch, pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
N, L = int(os.environ.get("BV_N", "512")), int(os.environ.get("BV_L", "256"))
pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, L), n_chains=N, n_rounds=20, show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
e.run_scans(1, 4); adapt(pt, reduce_recorders(pt))
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 8); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 8 * 1e3)
print("Ising %dx%d N=%d %-36s %.4f ms/scan" % (L, L, N, os.path.basename(os.environ.get("PTE_LIB", "default")), best))
