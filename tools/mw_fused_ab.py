"""Round 6: k_scans_langevin_mw (one launch per pte_run_scans) against explore + swap launches per scan, bit for bit: states, rng, chains, every recorder
after several rounds.  PTE_LIB=<development build> python tools/mw_fused_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import numpy as np
import pigeons_amd as P
_variant.apply()
from pigeons_amd import _lib
rec = [P.round_trip, P.index_process, P.log_sum_ratio]
def mk(kind, d, n):
    if kind == "mvn": return P.Inputs(target=P.toy_mvn_target(d), n_chains=n, explorer=P.AutoMALA(), record=rec, n_rounds=4, seed=3, show_report=False)
    if kind == "mala": return P.Inputs(target=P.toy_mvn_target(d), n_chains=n, explorer=P.MALA(step_size=0.05), record=rec, n_rounds=4, seed=3, show_report=False)
    return P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=n, explorer=P.AutoMALA(), record=rec, n_rounds=4, seed=3, show_report=False)
bad = 0
for kind, d, n in (("mvn", 1024, 64), ("mvn", 600, 33), ("mala", 1024, 16), ("funnel", 1024, 48), ("funnel", 700, 10), ("mvn", 1024, 1024), ("funnel", 1024, 1024)):
    a = P.PT(mk(kind, d, n)); b = P.PT(mk(kind, d, n), debug_kernel=_lib.KERNEL_TWO_LAUNCHES)
    names = (a.replicas.scan_loop_name(), b.replicas.scan_loop_name())
    ok = True
    for r in range(4 if n < 1000 else 3):
        P.next_round(a); ra = P.run_one_round(a); P.adapt(a, ra)
        P.next_round(b); rb = P.run_one_round(b); P.adapt(b, rb)
        for f in ("index_process", "swap_acceptance_pr", "log_sum_ratio", "explorer_n_steps", "explorer_acceptance_pr", "am_factors"):
            va, vb = getattr(ra, f, None), getattr(rb, f, None)
            if va is None: continue
            for xa, xb in zip(va if isinstance(va, tuple) else (va,), vb if isinstance(vb, tuple) else (vb,)):
                if not np.array_equal(np.asarray(xa), np.asarray(xb), equal_nan=True): ok = False; print("  differs:", f, "round", r + 1)
    sa, sb = a.replicas.states(), b.replicas.states()
    ok = ok and all(np.array_equal(u, v) for u, v in zip(sa, sb))
    st = a.replicas.scan_loop_stats()
    print("%-7s d=%-5d N=%-5d %s | %s   %s   %s" % (kind, d, n, names[0], names[1] or "(two launches)", "equal" if ok else "DIFFERENT", st), flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
