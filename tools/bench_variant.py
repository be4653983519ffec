"""ms/scan of the metric workload with the library named by PTE_LIB (A/B of tuning builds); events off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import torch, pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=int(os.environ.get("BV_N", "1024")), n_rounds=30, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
e.run_scans(1, 8); adapt(pt, reduce_recorders(pt))
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 32); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 32 * 1e3)
print("N=%s %-40s %.4f ms/scan" % (os.environ.get("BV_N", "1024"), os.path.basename(os.environ.get("PTE_LIB", "default")), best))
