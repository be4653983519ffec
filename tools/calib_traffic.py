"""Known-byte-count runs in THIS engine's access pattern (8 B per lane, one 512-B row segment per wave instruction),
to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md (HBM section) asks:
  k_test_sqr_norm reads  N*d*8 bytes (and writes N*8);
  k_explore_toy   writes N*d*8 bytes of fresh states (reads ~nothing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import numpy as np
import pigeons_amd as P
from pigeons_amd.engine import test_sqr_norm
N, d = 8192, 4096
x = np.random.default_rng(0).standard_normal((N, d))
for _ in range(3):
    test_sqr_norm(x)
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=4, record=[P.log_sum_ratio], show_report=False))
pt.replicas.run_scans(1, 4)
print("known bytes: k_test_sqr_norm reads %d ; k_explore_toy writes %d" % (N * d * 8, N * d * 8))
