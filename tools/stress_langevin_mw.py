"""Round 6 stress: k_explore_langevin_mw (AutoMALA / MALA / Compose / two legs / GaussianReference at 512 < d <= 1024) against the oracle over seeds and
shapes (integers exact, floats 1e-6), and -- test build -- bit for bit against the one-wave kernel.  STRESS_NSEEDS (default 3)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import pigeons_amd as P
from pigeons_amd import _lib
import oracle as O

def inputs(kind, d, nf, nv, seed, rounds):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    common = dict(n_chains=nf, n_chains_variational=nv, n_rounds=rounds, seed=seed, record=rec, show_report=False)
    ocommon = dict(n_chains=nf, n_chains_variational=nv, dim=d, seed=seed, am_preconditioner=2)
    if kind == "automala_mvn":
        return P.Inputs(target=P.toy_mvn_target(d), explorer=P.AutoMALA(), **common), dict(ocommon, explorer=O.EXPLORER_AUTOMALA)
    if kind == "mala_mvn":
        return P.Inputs(target=P.toy_mvn_target(d), explorer=P.MALA(step_size=0.05), **common), dict(ocommon, explorer=O.EXPLORER_MALA, am_step_size=0.05)
    if kind == "compose_mvn":
        return P.Inputs(target=P.toy_mvn_target(d), explorer=P.Compose(P.AutoMALA(), P.SliceSampler()), **common), dict(ocommon, explorer=O.EXPLORER_AUTOMALA, explorer2=O.EXPLORER_SLICE)
    if kind == "automala_funnel":
        return P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), explorer=P.AutoMALA(), **common), dict(ocommon, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1 / 9.)
    return P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), explorer=P.AutoMALA(), variational=P.GaussianReference(first_tuning_round=2), **common), \
           dict(ocommon, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1 / 9., variational_first_tuning_round=2)

def case(kind, d, nf, nv, seed, rounds=4):
    inp, okw = inputs(kind, d, nf, nv, seed, rounds)
    pt = P.PT(inp); ref = O.OraclePT(**okw)
    inp2, _ = inputs(kind, d, nf, nv, seed, rounds)
    one = P.PT(inp2, debug_kernel=_lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)
    assert pt.replicas.kernel_name().endswith("k_explore_langevin_mw") or kind == "compose_mvn", pt.replicas.kernel_name()
    for _ in range(rounds):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        P.next_round(one); red1 = P.run_one_round(one); P.adapt(one, red1)
        ref.run_round()
        if not np.array_equal(red.index_process, ref.index_process()): return "index_process vs oracle"
        if not np.allclose(red.swap_acceptance_pr[0], ref.swap_pr()[0], rtol=1e-6, atol=1e-300): return "swap_pr vs oracle"
        if not np.allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-6): return "schedule vs oracle"
        for a, b in ((red.index_process, red1.index_process), (red.swap_acceptance_pr[0], red1.swap_acceptance_pr[0]), (red.log_sum_ratio[0], red1.log_sum_ratio[0]),
                     (red.explorer_n_steps[0], red1.explorer_n_steps[0]), (red.am_factors[0], red1.am_factors[0])):
            if not np.array_equal(a, b, equal_nan=True): return "recorders vs the one-wave kernel"
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states(); x1, c1, r1 = one.replicas.states()
    if not (np.array_equal(chain, cr) and np.array_equal(rng, rr)): return "rng/chain vs oracle"
    if not np.allclose(x, xr, rtol=1e-6, atol=1e-9): return "state vs oracle"
    if not (np.array_equal(x, x1) and np.array_equal(chain, c1) and np.array_equal(rng, r1)): return "state vs the one-wave kernel"
    return None

bad = 0; n = 0
shapes = [(513, 4, 0), (576, 3, 3), (600, 5, 0), (640, 3, 2), (767, 4, 0), (768, 3, 0), (769, 3, 3), (900, 4, 0), (1000, 3, 2), (1023, 3, 0), (1024, 6, 0), (1024, 3, 3)]
for kind, (d, nf, nv), seed in itertools.product(["automala_mvn", "mala_mvn", "compose_mvn", "automala_funnel", "variational_funnel"], shapes, range(1, 1 + int(os.environ.get('STRESS_NSEEDS', '3')))):
    if nv and kind in ("automala_mvn", "mala_mvn", "compose_mvn"):
        continue                                            # (two legs on the MVN path: one shape is enough below)
    r = case(kind, d, nf, nv, seed)
    n += 1
    if r:
        bad += 1; print("MISMATCH", kind, d, nf, nv, seed, r, flush=True)
print("stress: %d configurations, %d mismatches" % (n, bad))
