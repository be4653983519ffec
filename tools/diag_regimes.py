"""ms / scan of C5 (Ising 256 x 256, 512 chains) and C3 (funnel d = 128, 1024 chains, AutoMALA) along a run: by chunk of scans, before and after
schedule / explorer adaptations -- which regime does a "handful of scans" measure?  Usage: python tools/diag_regimes.py [ising|funnel|slice]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np, torch
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt

def timed(e, k):
    torch.cuda.synchronize(); t = time.perf_counter(); e.run_scans(1, k); torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3

which = sys.argv[1] if len(sys.argv) > 1 else "ising"
rec = [P.round_trip, P.log_sum_ratio]
if which == "ising":
    mk = lambda: P.PT(P.Inputs(target=P.IsingLogPotential(1.0, 256), n_chains=512, record=rec, n_rounds=12, show_report=False))
    chunk = 4
elif which == "slice":
    mk = lambda: P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, explorer=P.SliceSampler(), record=rec, n_rounds=12, show_report=False))
    chunk = 8
else:
    mk = lambda: P.PT(P.Inputs(target=P.Funnel(128), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 128), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=12, show_report=False))
    chunk = 16
pt = mk(); e = pt.replicas
print(which, e.kernel_name(), e.scan_loop_name() or "two launches")
print("no adaptation:", " ".join("%.3f" % timed(e, chunk) for _ in range(6)))
for k in range(8):
    red = reduce_recorders(pt); adapt(pt, red)
    sw = np.asarray(red.swap_acceptance_pr[0])
    print("after adaptation %d (swap acceptance min %.3f mean %.3f):" % (k + 1, sw.min(), sw.mean()), " ".join("%.3f" % timed(e, chunk) for _ in range(4)), flush=True)
pt = mk(); e = pt.replicas
print("rounds of the algorithm (2^r scans, adapted after each):")
for r in range(1, 9 if which != "ising" else 6):
    t = timed(e, 2 ** r); adapt(pt, reduce_recorders(pt))
    print("  round %d: %.3f ms/scan" % (r, t), flush=True)
