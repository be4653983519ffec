import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import oracle as O, pigeons_amd as P
import test_gpu_parity as TP
from test_gpu_parity import _check_round
TP.RTOL = float(os.environ.get("LP_RTOL", "1e-6"))        # the north star's tolerance for floating-point recorders (the suite asserts 1e-9 over rounds 1-2;
                                                           # log_sum_ratio of round 3 differs by 7e-9 in 2 % of the pairs: ocml vs glibc exp / log1p under cancellation)
from pigeons_amd import _lib
N, d, R = 1024, 1024, int(os.environ.get("LP_ROUNDS", "4"))
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=R, explorer=P.SliceSampler(), seed=11, record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online], show_report=False),
          debug_kernel=_lib.KERNEL_TWO_LAUNCHES if os.environ.get("LP_TWO") else 0, reference_reduction=bool(os.environ.get("LP_REFRED")))
ref = O.OraclePT(n_chains=N, dim=d, seed=11, record_online=1, explorer=O.EXPLORER_SLICE, n_threads=max(1, len(os.sched_getaffinity(0))))
print(pt.replicas.scan_loop_name() or "two launches per scan", "rtol", TP.RTOL)
t = time.time()
if os.environ.get("LP_REFRED"):              # LP_REFRED=1: PTE_RECORD_REFERENCE_REDUCTION -- swap recorders, schedule, stepping stone EQUAL to the oracle's, not close
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_reference_reduction import _exact_round
    _check_round = lambda P, pt, ref: (_exact_round(P, pt, ref), np.testing.assert_array_equal(pt.replicas.states()[0], ref.states()[0]))
for r in range(R):
    _check_round(P, pt, ref)
    print("round", r + 1, "ok (fused scan loop vs oracle: index process, chains, RNG counters, recorders, schedule, states)", round(time.time() - t), "s", flush=True)
