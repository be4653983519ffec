"""Debug-only (PROF=1 development build, as tools/prof_mw.py): who is the slowest replica of a k_explore_langevin_mw launch?  Per chain: trial leapfrogs,
start / end on the 100 MHz clock, the compute unit it ran on.  PTE_LIB=build_variants/libpte_mw_<name>.so PW_PATH=funnel python tools/mw_pace.py"""
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
N, d, path = int(os.environ.get("PW_N", "1024")), int(os.environ.get("PW_D", "1024")), os.environ.get("PW_PATH", "funnel")
rec = [P.round_trip, P.log_sum_ratio]
inp = P.Inputs(target=P.toy_mvn_target(d), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False) if path == "mvn" else \
      P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, explorer=P.AutoMALA(), record=rec, n_rounds=8, show_report=False)
pt = P.PT(inp); e = pt.replicas
for r in range(1, 5):
    e.run_scans(1, 2 ** r); adapt(pt, reduce_recorders(pt))
L = _lib.load()
L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
runs = []
for scan in range(6):
    e.run_scans(2 + scan, 1)
    out = np.zeros(12 * N)
    assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    runs.append(out.reshape(N, 12).copy())
print("%s(%d) N = %d" % (path, d, N))
for k, o in enumerate(runs[1:]):
    o = o[1:]                                             # chain 0 draws iid
    chain = np.arange(1, N)
    st = o[:, 7] / 100.0; dur = o[:, 8] / 100.0; en = st + dur; t0 = st.min(); leaps = o[:, 9]
    cu = (o[:, 11] // 4096).astype(int)
    print("scan %d: launch %.0f us; ends p10 %.0f p50 %.0f p90 %.0f max %.0f; leapfrogs mean %.1f min %.0f max %.0f; corr(duration, leapfrogs) %.3f; %d compute units" % (
        k, (en - t0).max(), *np.percentile(en - t0, [10, 50, 90, 100]), leaps.mean(), leaps.min(), leaps.max(), np.corrcoef(dur, leaps)[0, 1], len(set(cu))))
    if k == len(runs) - 2:
        print("  chains      leapfrogs (mean)   end (mean / max) us")
        for b in range(0, N, 64):
            m = (chain >= b) & (chain < b + 64)
            print("  %4d-%4d   %7.1f            %6.0f / %6.0f" % (b, b + 63, leaps[m].mean(), (en - t0)[m].mean(), (en - t0)[m].max()))
        # per compute unit: the chains it hosted, the sum of their leapfrogs, the last end
        per = {}
        for i in range(len(chain)):
            per.setdefault(cu[i], []).append(i)
        sums = np.array([leaps[v].sum() for v in per.values()]); last = np.array([(en - t0)[v].max() for v in per.values()]); cnt = np.array([len(v) for v in per.values()])
        print("  per compute unit: replicas min %d max %d; sum of leapfrogs p10 %.0f p50 %.0f max %.0f; last end p10 %.0f p50 %.0f max %.0f; corr(sum, last end) %.3f" % (
            cnt.min(), cnt.max(), *np.percentile(sums, [10, 50, 100]), *np.percentile(last, [10, 50, 100]), np.corrcoef(sums, last)[0, 1]))
        worst = sorted(per.items(), key=lambda kv: -(en - t0)[kv[1]].max())[:4]
        for key, v in worst:
            print("  cu %04x: chains %s leapfrogs %s ends %s" % (key, [int(chain[i]) for i in v], [int(leaps[i]) for i in v], [int((en - t0)[i]) for i in v]))
        slow = np.argsort(-(en - t0))[:12]
        print("  slowest replicas: chain/leapfrogs/end:", " ".join("%d/%d/%d" % (chain[i], leaps[i], (en - t0)[i]) for i in slow))
