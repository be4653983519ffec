"""Round 6: k_scans_langevin_mw over long launches (up to round 12 = 4096 scans in ONE launch) against explore + swap launches per scan: every round's index process,
swap acceptance and step-size factors, and the final states, bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import pigeons_amd as P
from pigeons_amd import _lib
rec = [P.round_trip, P.index_process, P.log_sum_ratio]
def mk(N, d, flags, rounds):
    return P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=rounds, seed=5, show_report=False), debug_kernel=flags)
for N, d, rounds in ((1024, 1024, 9), (300, 600, 11), (64, 777, 12)):
    a = mk(N, d, 0, rounds); b = mk(N, d, _lib.KERNEL_TWO_LAUNCHES, rounds)
    t0 = time.time(); ok = True
    for r in range(rounds):
        P.next_round(a); ra = P.run_one_round(a); P.adapt(a, ra)
        P.next_round(b); rb = P.run_one_round(b); P.adapt(b, rb)
        ok = ok and np.array_equal(ra.index_process, rb.index_process) and np.array_equal(ra.swap_acceptance_pr[0], rb.swap_acceptance_pr[0]) and np.array_equal(ra.am_factors[0], rb.am_factors[0])
    sa, sb = a.replicas.states(), b.replicas.states()
    ok = ok and all(np.array_equal(u, v) for u, v in zip(sa, sb))
    print(N, d, rounds, a.replicas.scan_loop_name(), "equal" if ok else "DIFFERENT", a.replicas.scan_loop_stats(), "%.1f s" % (time.time() - t0), flush=True)
