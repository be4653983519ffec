"""ms/scan of the metric workload (BV_N chains, BV_D dimensions; BV_TWO=1: the launch-per-scan loop, BV_ONE=1: the one-kernel loop with one chain per workgroup) with the library named by PTE_LIB (A/B of tuning builds); events off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import torch, pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
from pigeons_amd import _lib
D = int(os.environ.get("BV_D", "1024"))
pt = P.PT(P.Inputs(target=P.toy_mvn_target(D), n_chains=int(os.environ.get("BV_N", "1024")), n_rounds=30, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]),
          debug_kernel=(_lib.KERNEL_TWO_LAUNCHES if os.environ.get("BV_TWO") else 0) | (_lib.KERNEL_SCAN_LOOP_ONE_CHAIN if os.environ.get("BV_ONE") else 0))
e = pt.replicas
e.run_scans(1, 32); adapt(pt, reduce_recorders(pt)); e.run_scans(1, 32)
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 64); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 64 * 1e3)
print("N=%s d=%d %-32s %-24s %.4f ms/scan" % (os.environ.get("BV_N", "1024"), D, os.path.basename(os.environ.get("PTE_LIB", "default")), e.scan_loop_name() or "two launches", best))
