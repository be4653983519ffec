"""One-off stress: the default SliceSampler kernel against the plain sequential kernel (pte_config.debug_kernel = 1) on many
seeds / shapes / parameters -- both on the GPU, so large sizes are cheap.  Everything must be bit-identical."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np
import pigeons_amd as P

def run(impl, N, d, seed, rounds, w, p):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, seed=seed, explorer=P.SliceSampler(w=w, p=p),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1], show_report=False), debug_kernel=impl)
    out = []
    for _ in range(rounds):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        out.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.explorer_n_steps[0].copy(), red.energy_ac1[2].copy()))
    return out, pt.replicas.states()

bad = 0; n = 0
shapes = [(64, 300), (33, 64), (16, 1000), (128, 129), (8, 4096), (50, 7)]
if os.environ.get("STRESS_MANY", "1") != "0":      # more than 2048 replicas: k_explore_slice8_lds10k at the tree depths it is quoted at (<4,9>, <6,9>)
    shapes += [(2304, 1024), (2100, 4096)]
params = [(10.0, 20), (1.0, 20), (0.2, 4), (100.0, 20)]
seed0 = int(os.environ.get("STRESS_SEED0", "1")); nseeds = int(os.environ.get("STRESS_NSEEDS", "6")); extra = int(os.environ.get("STRESS_EXTRA_ROUNDS", "0"))
for (N, d), (w, p), seed in itertools.product(shapes, params, range(seed0, seed0 + nseeds)):
    rounds = (2 if N > 2048 else 4 if d >= 1000 else 5) + extra
    if N > 2048 and seed >= seed0 + 2:          # the big shapes: two seeds per parameter set (16 runs of the sequential kernel at 2304 x 1024 / 2100 x 4096)
        continue
    a, sa = run(1, N, d, seed, rounds, w, p)
    b, sb = run(int(os.environ.get("STRESS_IMPL", "0")), N, d, seed, rounds, w, p)      # 0 = the default kernel; 2 / 5 / 7: test build
    ok = all(np.array_equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb)) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", N, d, w, p, seed, flush=True)
print("stress: %d configurations, %d mismatches" % (n, bad))
