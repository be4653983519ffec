"""One-off stress: lane-speculative Ising kernel against the byte-lattice scalar kernel over seeds / sizes / betas."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np
import pigeons_amd as P

def run(impl, L, N, beta, seed, rounds, n_steps):
    from pigeons_amd import _lib
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(beta, L), n_chains=N, n_rounds=rounds, seed=seed, explorer=P.IsingMetropolis(n_steps=n_steps),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1], show_report=False),
              debug_kernel=_lib.KERNEL_ISING_BYTES if impl == "bytes" else 0)
    out = []
    for _ in range(rounds):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        out.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.energy_ac1[2].copy()))
    return out, pt.replicas.states()

bad = 0; n = 0
for (L, N), beta, seed in itertools.product([(32, 12), (64, 9), (96, 5), (256, 4)], [0.1, 0.44, 1.0, 1e-7], range(1, 6)):
    rounds = 3 if L >= 96 else 5
    a, sa = run("bytes", L, N, beta, seed, rounds, 1 + seed % 3)
    b, sb = run("spec", L, N, beta, seed, rounds, 1 + seed % 3)
    ok = all(np.array_equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb)) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
    n += 1
    if not ok:
        bad += 1; print("MISMATCH", L, N, beta, seed, flush=True)
print("stress: %d configurations, %d mismatches" % (n, bad))
