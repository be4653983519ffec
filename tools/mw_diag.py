"""how many leapfrogs a refresh makes, and what a scan costs a LONE workgroup (N small) -- k_explore_langevin_mw"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import numpy as np
import pigeons_amd as P
_variant.apply()
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt
rec = [P.round_trip, P.log_sum_ratio]
for N in (8, 256, 1024):
  for name, mk in (("mvn1024", lambda: P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)),
                 ("funnel1024", lambda: P.Inputs(target=P.Funnel(1024), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 1024), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False))):
    for label, flags in (("mw", 0), ("one", _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)):
        if os.environ.get("PTE_LIB") and label == "one": continue
        pt = P.PT(mk(), debug_kernel=flags); e = pt.replicas
        for r in range(1, 5):
            e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
        e.run_scans(1, 2)
        e.reduce()
        t = time.perf_counter(); e.run_scans(1, 16); dt = time.perf_counter() - t
        e.reduce()
        am, an, ss, sn = e.explorer_stats()
        nref = 3 * int(np.ceil(1024 ** 0.35))
        print("%-10s N=%-5d %-4s %8.3f ms/scan; per chain-scan: searches %.1f, leapfrogs %.1f (n_refresh %d); step size %.4g" % (name, N, label, dt / 16 * 1e3, sn[1:].mean() / 16, ss[1:].mean() / 16, nref, pt.shared.explorer.step_size if hasattr(pt.shared, "explorer") else float("nan")), flush=True)
        del pt, e
