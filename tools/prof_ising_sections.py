"""Debug-only (build_variants/libpte_v_isprof.so = tools/build_variant.sh isprof -DPTE_PROFILE_WAVES -DPTE_PROFILE_ISING_SECTIONS): shader-clock
cycles per 16-site chunk of k_explore_ising_spec at the C5 shard shape, split into the vector pass, the chase and the rest of the sweep loop
(word loop, refills, LDS).  The stamps themselves cost (s_memtime + its wait): read the split, not the total."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
from pigeons_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libpte_v_isprof.so")
import numpy as np
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
N, Lsz = 512, 256
pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, Lsz), n_chains=N, n_rounds=10, show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
for r in range(1, 4):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
e.run_scans(9, 1)
out = np.zeros(4 * N)
assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
o = out.reshape(N, 4)[1:]                      # chain 0 = the reference (refresh only)
chunks = 3 * Lsz * Lsz / 16
print("per 16-site chunk, mean over %d chains: loop %.0f cycles = vector pass %.0f + chase %.0f + rest %.0f" %
      (N - 1, o[:, 0].mean() / chunks, o[:, 1].mean() / chunks, o[:, 2].mean() / chunks, (o[:, 0] - o[:, 1] - o[:, 2]).mean() / chunks))
betas = e.schedule()
for k in range(0, 10, 3):
    sl = slice(max(1, k * N // 10), (k + 1) * N // 10)
    q = out.reshape(N, 4)[sl]
    print("  chains %3d-%3d beta %.3f-%.3f: loop %.0f pass %.0f chase %.0f" % (sl.start, sl.stop - 1, betas[sl.start], betas[sl.stop - 1], q[:, 0].mean() / chunks, q[:, 1].mean() / chunks, q[:, 2].mean() / chunks))
