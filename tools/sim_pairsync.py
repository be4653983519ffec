"""What would ONE launch per pte_run_scans buy the SliceSampler scan -- priced on recorded per-wave durations before anything is built
(VERDICT r04 next-round item 6).

A scan is two launches today: the explore kernel is as long as its SLOWEST wave (one wave per replica, one per SIMD at N = 1024), then the
swap kernel and the launch gaps.  A persistent kernel can replace the launch boundary by
  (b) a device-scope barrier between the explore and swap phases (the verdict's proposal): the scan still waits for the slowest of N waves,
      only the ~18 us of launch / swap-kernel / gap time shrink to two barriers;
  (c) PAIRWISE hand-shakes: workgroup c always holds chain c, and the DEO swap of the pair (c, c +- 1) needs nothing but the two partners'
      swap statistics -- each wave publishes {log-ratio, uniform, slot} with a release store and waits for ITS PARTNER only.  The scan of chain c
      then starts when c's own previous swap is done: f_c(t) = max(f_c(t-1) + T_c(t), f_p(t-1) + T_p(t)) + delta, p = DEO partner at scan t.
      Waves only wait for neighbours, so random fluctuations average out along the ladder; what remains is the slowest CHAIN's mean.

This script (GPU box, -DPTE_PROFILE_WAVES build) records T_c(t) -- per-wave start / end stamps of the explore kernel on the 100 MHz clock -- for
S scans and replays the three schemes on them.  Usage: python tools/sim_pairsync.py [N d scans]   (needs build_variants/libpte_v_waves.so:
tools/build_variant.sh waves -DPTE_PROFILE_WAVES)"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(N, d, scans):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
    from pigeons_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libpte_v_waves.so")
    import pigeons_amd as P
    from pigeons_amd.pt import reduce_recorders, adapt
    if os.environ.get("PS_ISING"):               # PS_ISING=1: the C5 shape (d = lattice side)
        pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, d), n_chains=N, n_rounds=10, show_report=False, record=[P.round_trip, P.log_sum_ratio]), debug_kernel=_lib.KERNEL_TWO_LAUNCHES)
    else:
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]),
                  debug_kernel=_lib.KERNEL_TWO_LAUNCHES)       # (the stamps are the per-scan kernel's)
    e = pt.replicas
    e.run_scans(1, 8); adapt(pt, reduce_recorders(pt))
    e.run_scans(1, 8); adapt(pt, reduce_recorders(pt))
    L = _lib.load()
    L.pte_debug_wave_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    T = np.zeros((scans, N)); span = np.zeros(scans)
    for s in range(scans):
        e.run_scans(s + 1, 1)                        # scan numbers 1, 2, ...: odd / even graphs alternate as in a round
        out = np.zeros(4 * N)
        assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        o = out.reshape(N, 4)
        ok = o[:, 0] > 0                              # (the reference chain's wave returns before the stamps: its i.i.d. refresh takes a few us)
        T[s] = np.where(ok, (o[:, 1] - o[:, 0]) / 100.0, 5.0)      # us
        span[s] = (o[ok, 1].max() - o[ok, 0].min()) / 100.0
    return T, span


def partner(N, scan, c):
    even = scan % 2 == 0
    chain_even = (c + 1) % 2 == 0
    p = (c + 1) + (1 if chain_even == even else -1)
    return c if p == 0 or p == N + 1 else p - 1


def replay(T, span, launch_overhead_us=18.0, barrier_us=4.0, delta_us=2.0):
    S, N = T.shape
    cur = float(np.sum(span) + S * launch_overhead_us)
    bar = float(np.sum(span) + S * 2 * barrier_us)
    f = np.zeros(N)
    for s in range(S):
        e = f + T[s]
        g = e.copy()
        for c in range(N):
            p = partner(N, s + 1, c)
            g[c] = max(e[c], e[p]) + delta_us
        f = g
    pair = float(f.max())
    return cur / S, bar / S, pair / S


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    scans = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cache = os.path.join(ROOT, "gpurun_out", "r05_wave_durations_%sN%d_d%d.npz" % ("ising_" if os.environ.get("PS_ISING") else "", N, d))
    if os.path.exists(cache) and os.environ.get("PAIRSYNC_REPLAY"):
        z = np.load(cache); T, span = z["T"], z["span"]
    else:
        T, span = record(N, d, scans)
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        np.savez_compressed(cache, T=T, span=span)
    S = T.shape[0]
    print("%s, %d chains: %d scans recorded (PTE_PROFILE_WAVES build)" % (("Ising %d x %d, IsingMetropolis(3)" % (d, d)) if os.environ.get("PS_ISING") else "toy_mvn_target(%d), SliceSampler" % d, N, S))
    print("  per scan: launch span mean %.1f us; wave duration mean %.1f, max-over-waves mean %.1f us (mean wave = %.3f of the span)"
          % (span.mean(), T.mean(), T.max(axis=1).mean(), T.mean() / span.mean()))
    per_chain = T.mean(axis=0)
    print("  per chain (mean over scans): min %.1f  median %.1f  p99 %.1f  max %.1f us (chain %d); per-scan sd of one chain's duration %.1f us"
          % (per_chain.min(), np.median(per_chain), np.percentile(per_chain, 99), per_chain.max(), int(per_chain.argmax()), (T - per_chain).std()))
    for ov, b, dl in ((18.0, 4.0, 2.0), (18.0, 2.0, 1.0), (18.0, 6.0, 4.0)):
        cur, bar, pair = replay(T, span, ov, b, dl)
        print("  launch overhead %.0f us, barrier %.0f us, hand-shake %.0f us:  two launches %.1f us/scan | grid barriers %.1f (x%.3f) | pairwise %.1f (x%.3f)"
              % (ov, b, dl, cur, bar, cur / bar, pair, cur / pair))


if __name__ == "__main__":
    main()
