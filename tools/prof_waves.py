"""Debug-only (-DPTE_PROFILE_WAVES build, build_variants/libpte_waves.so): per-wave start / end / placement of k_explore_slice8 at the
metric configuration.  Answers: is the launch as long as its MEAN wave or as its SLOWEST one, and what makes a wave slow -- its
chain (work per replica-step grows with beta) or where it runs (XCD / CU / SIMD)?"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
lib = os.path.join(ROOT, "build_variants", "libpte_waves.so")
from pigeons_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
N, d = int(os.environ.get("PW_N", "1024")), int(os.environ.get("PW_D", "1024"))
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]))
e = pt.replicas
e.run_scans(1, 8); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
durs = []
for scan in range(6):
    e.run_scans(1, 1)
    out = np.zeros(4 * N)
    assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    o = out.reshape(N, 4)
    t0, t1 = o[:, 0], o[:, 1]
    dur = (t1 - t0) / 100.0                      # us
    span = (t1.max() - t0.min()) / 100.0
    durs.append(dur)
    hw = o[:, 2].astype(np.int64); xcc = o[:, 3].astype(np.int64) & 15
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
    print("scan %d: launch span %.1f us; wave duration mean %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us; mean/span %.3f; start spread %.1f us"
          % (scan, span, dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.mean() / span, (t0.max() - t0.min()) / 100.0))
dur = np.mean(durs[1:], axis=0)
print("per-chain mean duration over 5 scans, by chain decile (chain 0 = reference, i.i.d. refresh):")
for k in range(10):
    sl = slice(k * N // 10, (k + 1) * N // 10)
    print("  chains %4d-%4d: mean %.1f us  max %.1f" % (sl.start, sl.stop - 1, dur[sl].mean(), dur[sl].max()))
red = reduce_recorders(pt)
steps = red.explorer_n_steps[0] / np.maximum(red.explorer_n_steps[1], 1)
print("corr(duration, chain index) = %.3f; corr(duration, mean steps) = %.3f" % (np.corrcoef(dur[1:], np.arange(1, N))[0, 1], np.corrcoef(dur[1:], steps[1:])[0, 1]))
# placement of the LAST scan
print("last scan by XCD: " + "  ".join("x%d %.1f/%.1f" % (x, durs[-1][xcc == x].mean(), durs[-1][xcc == x].max()) for x in range(8)))
print("last scan by SIMD: " + "  ".join("s%d %.1f/%.1f" % (s, durs[-1][simd == s].mean(), durs[-1][simd == s].max()) for s in range(4)))
key = (xcc * 8 + se) * 16 + cu
cnt = np.bincount(key, minlength=1024)
print("waves per CU histogram:", np.bincount(cnt[cnt > 0]))
slow = np.argsort(durs[-1])[-10:]
print("10 slowest waves of the last scan: " + ", ".join("chain %d (x%d se%d cu%d s%d) %.1f" % (c, xcc[c], se[c], cu[c], simd[c], durs[-1][c]) for c in slow))
