"""Per-round ms / scan along the reference's own round structure (2, 4, ... 2^R scans, adaptation after each) for every explorer / path the device
serves: a regime in which a kernel falls off a cliff (the Ising kernels at beta < 1e-6 did, through round 4: x55 for one round) shows as a round that
costs several times the others.  Usage: python tools/soak_regimes.py [R]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np, torch
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders

R = int(sys.argv[1]) if len(sys.argv) > 1 else 9
rec = [P.round_trip, P.log_sum_ratio]
ref9 = lambda d: P.ScaledPrecisionNormalLogPotential(1 / 9., d)
cfgs = [
    ("SliceSampler, MVN d=256, 256 chains", dict(target=P.toy_mvn_target(256), n_chains=256, explorer=P.SliceSampler())),
    ("SliceSampler, MVN d=2, 10 chains", dict(target=P.toy_mvn_target(2), n_chains=10, explorer=P.SliceSampler())),
    ("ToyExplorer, MVN d=1024, 512 chains", dict(target=P.toy_mvn_target(1024), n_chains=512, explorer=P.ToyExplorer())),
    ("AutoMALA, MVN d=64, 256 chains", dict(target=P.toy_mvn_target(64), n_chains=256, explorer=P.AutoMALA())),
    ("AutoMALA, funnel d=16, 256 chains", dict(target=P.Funnel(16), reference=ref9(16), n_chains=256, explorer=P.AutoMALA())),
    ("AutoMALA, funnel d=128, 64 chains", dict(target=P.Funnel(128), reference=ref9(128), n_chains=64, explorer=P.AutoMALA())),
    ("MALA, MVN d=64, 128 chains", dict(target=P.toy_mvn_target(64), n_chains=128, explorer=P.MALA())),
    ("SliceSampler, funnel d=16, 128 chains", dict(target=P.Funnel(16), reference=ref9(16), n_chains=128, explorer=P.SliceSampler())),
    ("Compose(Slice, AutoMALA), MVN d=32, 64 chains", dict(target=P.toy_mvn_target(32), n_chains=64, explorer=P.Compose(P.SliceSampler(), P.AutoMALA()))),
    ("IsingMetropolis, 64 x 64, 128 chains", dict(target=P.IsingLogPotential(1.0, 64), n_chains=128)),
    ("IsingMetropolis, 24 x 24 (byte kernel), 64 chains", dict(target=P.IsingLogPotential(0.6, 24), n_chains=64)),
    ("two legs, SliceSampler, MVN d=64, 64 + 64 chains", dict(target=P.toy_mvn_target(64), n_chains=64, n_chains_variational=64, variational=None, explorer=P.SliceSampler())),
    ("GaussianReference, AutoMALA, funnel d=8, 32 + 32 chains", dict(target=P.Funnel(8), reference=ref9(8), n_chains=32, n_chains_variational=32, variational=P.GaussianReference(first_tuning_round=3), explorer=P.AutoMALA())),
]
bad = 0
for name, kw in cfgs:
    pt = P.PT(P.Inputs(n_rounds=R, record=rec + ([P.online] if kw.get("variational") is not None else []), show_report=False, **kw))
    ms = []
    for r in range(1, R + 1):
        assert P.next_round(pt)
        torch.cuda.synchronize(); t = time.perf_counter()
        pt.replicas.run_scans(1, 2 ** r)                      # (the scan loop alone: the reduction and the adaptation are per-round host work)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t) / 2 ** r * 1e3)
        pt.shared.iterators.scan = 0
        P.adapt(pt, reduce_recorders(pt))
    med = float(np.median(ms[2:]))
    worst = max(ms[2:]) / med
    flag = "  <-- CLIFF?" if worst > 3 else ""
    bad += worst > 3
    print("%-58s %-22s ms/scan by round: %s   max/median (rounds 3..) %.2f%s" % (name, pt.replicas.scan_loop_name() or "two launches", " ".join("%.3f" % m for m in ms), worst, flag), flush=True)
print("configurations with a round above 3x the median:", bad)
