"""Debug-only (-DPTE_PROFILE_AM build, build_variants/libpte_amprof.so): shader-clock time per section of AutoMALA's refresh loop at
BASELINE configs[2] (funnel(128), N = 1024), averaged over the waves.  Answers where a scan's ~700 k cycles go: momentum draw,
gradient at the start, the two step-size searches, the proposal leapfrog, accept / reject."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
lib = os.path.join(ROOT, "build_variants", "libpte_amprof.so")
from pigeons_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
N, d = int(os.environ.get("PW_N", "1024")), int(os.environ.get("PW_D", "128"))
pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, n_rounds=10, explorer=P.AutoMALA(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
for r in range(1, 5):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
names = ["momentum draw", "grad at start + kinetic", "2 rand + 2 log", "step-size search (forward)", "proposal leapfrog", "step-size search (reversed)", "accept / reject", "loop head"]
acc = []
for scan in range(6):
    e.run_scans(2 + scan, 1)                                            # (scan 1 of a round skips the MH step)
    out = np.zeros(12 * N)
    assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    acc.append(out.reshape(N, 12)[1:])                                  # (chain 0 = reference: i.i.d. refresh, no stamps)
o = np.mean(acc[1:], axis=0)
tot = o[:, :8].sum(axis=1)
n_refresh = o[0, 11]
leaps = o[:, 9]                                                         # sum of (1 + n_steps) over the 2 * n_refresh searches
print("funnel(%d) N = %d AutoMALA: n_refresh = %d; per wave and scan: %.0f shader-clock ticks in the loop = %.1f us on the 100 MHz clock"
      % (d, N, n_refresh, tot.mean(), o[:, 8].mean() / 100.0))
print("ticks per us: %.1f" % (tot.mean() / (o[:, 8].mean() / 100.0)))
for k in range(8):
    print("  %-30s %9.0f ticks  %5.1f %%   per refresh %7.0f" % (names[k], o[:, k].mean(), 100.0 * o[:, k].mean() / tot.mean(), o[:, k].mean() / n_refresh))
print("trial leapfrogs per scan (both searches): %.1f  -> %.0f ticks per trial leapfrog; the proposal leapfrog %.0f" %
      (leaps.mean(), (o[:, 3] + o[:, 5]).mean() / leaps.mean(), o[:, 4].mean() / n_refresh))
print("gradient evaluations per scan: %.1f" % (leaps.mean() + 2 * n_refresh))
rt = np.mean([a[:, 8] for a in acc[1:]], axis=0) / 100.0
print("wave duration (us): mean %.1f  median %.1f  p90 %.1f  max %.1f; by chain decile: %s" % (rt.mean(), np.median(rt), np.percentile(rt, 90), rt.max(),
      " ".join("%.0f" % rt[k * (N - 1) // 10:(k + 1) * (N - 1) // 10].mean() for k in range(10))))
print("trial leapfrogs by chain decile: " + " ".join("%.1f" % leaps[k * (N - 1) // 10:(k + 1) * (N - 1) // 10].mean() for k in range(10)))
