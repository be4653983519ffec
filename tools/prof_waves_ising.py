"""Debug-only (-DPTE_PROFILE_WAVES build, build_variants/libpte_waves.so): per-wave duration of k_explore_ising_spec at the C5 shard shape
(Ising 256 x 256, 512 chains): is the launch as long as its mean wave or as its slowest one, and which chains are slow?"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
from pigeons_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libpte_waves.so")
import numpy as np
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
N, Lsz = int(os.environ.get("PW_N", "512")), int(os.environ.get("PW_L", "256"))
pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, Lsz), n_chains=N, n_rounds=10, show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
for r in range(1, 4):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
durs = []
for scan in range(5):
    e.run_scans(2 + scan, 1)
    out = np.zeros(4 * N)
    assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    o = out.reshape(N, 4)
    dur = (o[:, 1] - o[:, 0]) / 100.0
    span = (o[:, 1].max() - o[:, 0].min()) / 100.0
    durs.append(dur)
    print("scan %d: launch span %.0f us; wave duration mean %.0f  median %.0f  p90 %.0f  max %.0f us (chain %d); mean/span %.3f" %
          (scan, span, dur[1:].mean(), np.median(dur[1:]), np.percentile(dur[1:], 90), dur.max(), int(dur.argmax()), dur[1:].mean() / span))
dur = np.mean(durs[1:], axis=0)
betas = e.schedule()
print("mean duration by chain decile (chain 0 = reference: Bernoulli refresh):")
for k in range(10):
    sl = slice(max(1, k * N // 10), (k + 1) * N // 10)
    print("  chains %3d-%3d  beta %.3f-%.3f: mean %.0f us  max %.0f" % (sl.start, sl.stop - 1, betas[sl.start], betas[sl.stop - 1], dur[sl].mean(), dur[sl].max()))
