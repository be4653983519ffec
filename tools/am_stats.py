"""C3 (funnel d=128, N=1024, AutoMALA): leapfrog counts and time per refresh, after a few adaptation rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np
import pigeons_amd as P
d, N = 128, 1024
pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, explorer=P.AutoMALA(),
                   record=[P.round_trip, P.log_sum_ratio], n_rounds=8, show_report=False))
for r in range(7):
    P.next_round(pt); t = time.perf_counter(); red = P.run_one_round(pt); dt = time.perf_counter() - t; P.adapt(pt, red)
    ss, sn = red.explorer_n_steps
    fm, fn = red.am_factors
    print("round %d: %.3f ms/scan  step_size %.4g  leapfrog-evals per auto_step_size call: mean %.2f max-chain %.2f; n calls/scan/replica %.1f"
          % (r + 1, dt / 2 ** (r + 1) * 1e3, pt.shared.explorer.step_size, ss[1:].sum() / sn[1:].sum(), (ss[1:] / sn[1:]).max(), sn[1:].mean() / 2 ** (r + 1)))
