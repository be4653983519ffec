"""Debug-only (-DPTE_PROFILE_AM development build, tools/build_variant_mw.sh <name> -DPTE_PROFILE_AM with PROF=1): shader-clock time per section of
k_explore_langevin_mw's refresh loop as wave 0 of each replica sees it.  PTE_LIB=build_variants/libpte_mw_<name>.so python tools/prof_mw.py"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")]
import _variant
import numpy as np
import pigeons_amd as P
_variant.apply()
from pigeons_amd import _lib
from pigeons_amd.pt import reduce_recorders, adapt
N, d, path = int(os.environ.get("PW_N", "1024")), int(os.environ.get("PW_D", "1024")), os.environ.get("PW_PATH", "mvn")
rec = [P.round_trip, P.log_sum_ratio]
inp = P.Inputs(target=P.toy_mvn_target(d), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False) if path == "mvn" else \
      P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)
pt = P.PT(inp); e = pt.replicas
for r in range(1, 5):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
names = ["momentum (16 blocks, handed on) + start store", "grad at start / |p|^2 + g_s to LDS", "2 rand + 2 log", "step-size search (forward)", "kept trial from LDS",
         "step-size search (reversed)", "accept / reject"]
acc = []
for scan in range(6):
    e.run_scans(2 + scan, 1)
    out = np.zeros(12 * N)
    assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    acc.append(out.reshape(N, 12)[1:])
o = np.mean(acc[1:], axis=0)
tot = o[:, :7].sum(axis=1); nref = int(acc[-1][0, 11]) % 4096; leaps = o[:, 9]
print("%s(%d) N = %d k_explore_langevin_mw: n_refresh %d; per replica and scan %.0f ticks in the loop = %.1f us (100 MHz clock): %.1f ticks per us" % (path, d, N, nref, tot.mean(), o[:, 8].mean() / 100.0, tot.mean() / (o[:, 8].mean() / 100.0)))
for k in range(7):
    print("  %-48s %9.0f ticks  %5.1f %%   per refresh %7.0f" % (names[k], o[:, k].mean(), 100.0 * o[:, k].mean() / tot.mean(), o[:, k].mean() / nref))
print("  of which: waiting at exchanges / barriers (wave 0) %.0f ticks (%.1f %%)" % (o[:, 10].mean(), 100 * o[:, 10].mean() / tot.mean()))
st = acc[-1][:, 7] / 100.0; en = st + acc[-1][:, 8] / 100.0; t0 = st.min()
print("last scan, us after the first workgroup's start: starts p10 / p50 / p90 / max %.0f %.0f %.0f %.0f; ends p10 / p50 / p90 / max %.0f %.0f %.0f %.0f; a replica's loop: mean %.0f us" % (
      *np.percentile(st - t0, [10, 50, 90, 100]), *np.percentile(en - t0, [10, 50, 90, 100]), (en - st).mean()))
print("trial leapfrogs per scan %.1f -> %.0f ticks per leapfrog of the two searches" % (leaps.mean(), (o[:, 3] + o[:, 5]).mean() / leaps.mean()))
