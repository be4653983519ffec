import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import torch, pigeons_amd as P
pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, n_rounds=30, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
e.run_scans(1, 8)
for timing in (False, True, False, True):
    e.timing_reset(timing)
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, 64); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ex = e.timing(0) if timing else (0, 0)
    print("timing", timing, "ms/scan %.4f" % (dt / 64 * 1e3), "kernel avg %.4f" % (ex[0] / max(ex[1], 1)))
