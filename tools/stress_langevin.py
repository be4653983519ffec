"""One-off stress: AutoMALA / MALA / Compose / two legs / GaussianReference on the device against the oracle over seeds
and shapes (integers exact, floats 1e-6)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import pigeons_amd as P
import oracle as O

def case(kind, d, nf, nv, seed, rounds=5):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    common = dict(n_chains=nf, n_chains_variational=nv, n_rounds=rounds, seed=seed, record=rec, show_report=False)
    ocommon = dict(n_chains=nf, n_chains_variational=nv, dim=d, seed=seed, am_preconditioner=2)
    if kind == "automala_mvn":
        inp = P.Inputs(target=P.toy_mvn_target(d), explorer=P.AutoMALA(), **common); okw = dict(explorer=O.EXPLORER_AUTOMALA)
    elif kind == "mala_mvn":
        inp = P.Inputs(target=P.toy_mvn_target(d), explorer=P.MALA(step_size=0.15), **common); okw = dict(explorer=O.EXPLORER_MALA, am_step_size=0.15)
    elif kind == "compose_mvn":
        inp = P.Inputs(target=P.toy_mvn_target(d), explorer=P.Compose(P.AutoMALA(), P.SliceSampler()), **common)
        okw = dict(explorer=O.EXPLORER_AUTOMALA, explorer2=O.EXPLORER_SLICE)
    elif kind == "automala_funnel":
        inp = P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), explorer=P.AutoMALA(), **common)
        okw = dict(explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1 / 9.)
    elif kind == "variational_funnel":
        inp = P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), explorer=P.AutoMALA(),
                       variational=P.GaussianReference(first_tuning_round=2), **common)
        okw = dict(explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1 / 9., variational_first_tuning_round=2)
    pt = P.PT(inp); ref = O.OraclePT(**ocommon, **okw)
    for _ in range(rounds):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        ref.run_round()
        if not np.array_equal(red.index_process, ref.index_process()): return "index_process"
        if not np.allclose(red.swap_acceptance_pr[0], ref.swap_pr()[0], rtol=1e-6, atol=1e-300): return "swap_pr"
        if not np.allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-6): return "schedule"
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    if not (np.array_equal(chain, cr) and np.array_equal(rng, rr)): return "rng/chain"
    if not np.allclose(x, xr, rtol=1e-6, atol=1e-9): return "state"
    return None

bad = 0; n = 0
for kind, (d, nf, nv), seed in itertools.product(["automala_mvn", "mala_mvn", "compose_mvn", "automala_funnel", "variational_funnel"],
                                                 [(3, 6, 0), (40, 5, 4), (130, 4, 0), (300, 3, 3)] + ([(64, 5, 0), (128, 4, 2), (256, 3, 0), (512, 3, 2), (1024, 3, 0)] if os.environ.get('STRESS_WHOLE_BLOCKS', '1') != '0' else []), range(1, 1 + int(os.environ.get('STRESS_NSEEDS', '4')))):
    r = case(kind, d, nf, nv, seed)
    n += 1
    if r:
        bad += 1; print("MISMATCH", kind, d, nf, nv, seed, r, flush=True)
print("stress: %d configurations, %d mismatches" % (n, bad))
