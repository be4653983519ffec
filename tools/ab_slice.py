"""Dev check of a tuning build (PTE_LIB=build_variants/...): the default SliceSampler kernel against the plain sequential kernel
(debug_kernel = 1) of the SAME library at d = 1024 / 4096 (what PTE_DEV_FEW_NLU builds hold), several seeds and parameter sets, bit for bit;
then ms / scan of the metric workload and of the C4 shard."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import numpy as np, torch
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt

def run(impl, N, d, seed, rounds, w, p):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, seed=seed, explorer=P.SliceSampler(w=w, p=p),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), debug_kernel=impl)
    out = []
    for _ in range(rounds):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        out.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.explorer_n_steps[0].copy()))
    return out, pt.replicas.states()

bad = n = 0
if not os.environ.get("AB_NO_CHECK"):
    for (N, d), (w, p), seed in itertools.product([(96, 1024), (16, 4096)], [(10.0, 20), (1.0, 20), (0.2, 4), (100.0, 20), (10.0, 2)], range(1, 4)):
        a, sa = run(1, N, d, seed, 4, w, p)
        b, sb = run(0, N, d, seed, 4, w, p)
        ok = all(np.array_equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb)) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", N, d, w, p, seed, flush=True)
    print("%s: %d configurations against the sequential kernel, %d mismatches" % (os.path.basename(os.environ.get("PTE_LIB", "default")), n, bad), flush=True)
for N, d, scans in ((1024, 1024, 32), (1024, 4096, 8), (256, 1024, 32)):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=30, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
    e = pt.replicas
    e.run_scans(1, 8); adapt(pt, reduce_recorders(pt))
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, scans); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / scans * 1e3)
    print("%-40s N=%d d=%d  %.4f ms/scan" % (os.path.basename(os.environ.get("PTE_LIB", "default")), N, d, best), flush=True)
