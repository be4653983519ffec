"""Debug-only (-DPTE_S8_DUO_STATS build named by PTE_LIB): the kernel prints rounds and coordinates per round of a few chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import pigeons_amd as P
N = int(os.environ.get("DS_N", "256"))
pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False, record=[P.log_sum_ratio]))
e = pt.replicas
print(e.kernel_name(), flush=True)
e.run_scans(1, 1)
