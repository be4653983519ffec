"""Round 6: ms / scan of AutoMALA / MALA at 512 < d <= 1024, N = 1024 -- k_explore_langevin_mw (four waves per replica) against the one-wave
kernel it replaces (test build, PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE) -- prepared as runs in progress (rounds 1..4 of the algorithm).  BM_ONLY=mw|mw2|one (comma-separated)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import pigeons_amd as P
_variant.apply()          # PTE_LIB=<path>: a development build (tools/build_variant_mw.sh)
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt

rec = [P.round_trip, P.log_sum_ratio]
def cfgs():
    yield "toy_mvn(1024) AutoMALA", lambda: P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)
    yield "funnel(1024)  AutoMALA", lambda: P.Inputs(target=P.Funnel(1024), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 1024), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)
    yield "toy_mvn(600)  AutoMALA", lambda: P.Inputs(target=P.toy_mvn_target(600), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)
    yield "toy_mvn(1024) MALA", lambda: P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, explorer=P.MALA(step_size=0.05), record=rec, n_rounds=8, show_report=False)

only = os.environ.get("BM_ONLY", "")
for name, mk in cfgs():
    for label, flags in (("mw", 0), ("mw2", _lib.KERNEL_TWO_LAUNCHES), ("one", _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)):     # mw: one launch per call where eligible; mw2: explore + swap launches per scan
        if only and label not in only.split(","):
            continue
        pt = P.PT(mk(), debug_kernel=flags)
        e = pt.replicas
        for r in range(1, 5):
            e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
        e.run_scans(1, 2)
        ns = int(os.environ.get("BM_SCANS", "16"))
        per_call = int(os.environ.get("BM_PER_CALL", "0"))          # > 0: that many scans per pte_run_scans call
        t = time.perf_counter()
        if per_call:
            for k in range(ns // per_call): e.run_scans(1 + k * per_call, per_call)
        else: e.run_scans(1, ns)
        dt = time.perf_counter() - t
        print("%-24s %-4s %-62s %8.3f ms/scan" % (name, label, e.scan_loop_name() or e.kernel_name(), dt / ns * 1e3), flush=True)
        del pt, e
