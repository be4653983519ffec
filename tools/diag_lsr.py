import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import oracle as O, pigeons_amd as P
N, d, R = 1024, 1024, 3
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=R, explorer=P.SliceSampler(), seed=11, record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online], show_report=False))
ref = O.OraclePT(n_chains=N, dim=d, seed=11, record_online=1, explorer=O.EXPLORER_SLICE, n_threads=max(1, len(os.sched_getaffinity(0))))
for r in range(R):
    P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red); ref.run_round()
    up, un, dn, dnn = red.log_sum_ratio; upr, unr, dnr, dnnr = ref.log_sum_ratio()
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    rel = np.abs(up - upr) / np.abs(upr)
    bad = np.where(rel > 1e-9)[0]
    print("round", r + 1, "ints equal:", np.array_equal(red.index_process, ref.index_process()), np.array_equal(rng, rr), "| max rel diff up %.2e dn %.2e" % (rel.max(), (np.abs(dn - dnr) / np.abs(dnr)).max()),
          "| states max rel %.2e, exactly equal fraction %.5f" % (np.max(np.abs(x - xr) / np.maximum(np.abs(xr), 1e-300)), np.mean(x == xr)), "| bad pairs", bad[:12], len(bad))
    if len(bad):
        for c in bad[:5]:
            print("   pair", c, "dev %.15g ref %.15g diff %.3e" % (up[c], upr[c], up[c] - upr[c]), "schedule beta", pt.shared.tempering.schedule.grids[c])
    m, n = red.swap_acceptance_pr; mr, nr = ref.swap_pr()
    print("   swap pr max abs diff %.3e; schedule max abs diff %.3e" % (np.abs(m - mr).max(), np.abs(np.array(pt.shared.tempering.schedule.grids) - np.array(ref.schedule())).max()))
