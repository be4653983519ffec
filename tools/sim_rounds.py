"""Monte-Carlo of the speculative round of k_explore_slice8 (pte_slice7.hpp window tables, budgets 3 doublings / 9 proposals): how many
coordinates a round retires with the 64 hypotheses of one wave, and with 128 hypotheses over two waves (levels 5-7 on the second wave,
windows placed on the empirical distribution of the start offsets) -- the upper bound of what a two-waves-per-replica round could gain
before its hand-over costs.  Coordinate updates are the reference's procedure (SliceSampler.jl:97-186) on the toy MVN path in units of the
chain's standard deviation: slice { v : v^2 < x^2 + 2 E / prec }, w = 10 sqrt(prec).  Usage: python tools/sim_rounds.py"""
import numpy as np

LO, WD = [0, 3, 6, 10, 14], [1, 14, 16, 17, 16]
BD, BS, P_FAST_E = 3, 9, 1.0 - 0.0233


def coordinates(prec, n, rng):
    w = 10.0 * np.sqrt(prec)
    cnt = np.zeros(n, dtype=np.int64); valid = np.ones(n, dtype=bool)
    for i in range(n):
        x = rng.standard_normal(); E = rng.exponential(); Q = x * x + 2.0 * E
        L = x - w * rng.random(); R = L + w
        kd = 0
        while kd < 20 and (L * L < Q or R * R < Q):
            if rng.random() <= 0.5: L -= (R - L)
            else: R += (R - L)
            kd += 1
        m = 0
        while True:
            v = L + rng.random() * (R - L); m += 1
            if v * v < Q: break
            if v < x: L = v
            else: R = v
        cnt[i] = 2 + kd + m
        valid[i] = kd <= BD and m <= BS and rng.random() < P_FAST_E
    return cnt, valid


def rounds(cnt, valid, lo, wd):
    G = len(lo); i = 0; n = len(cnt) - G - 1; retired = []; offs = [[] for _ in range(G + 3)]
    while i < n:
        o = cnt[i]; r = 1                                   # lane 0 always retires (it may run past the budgets)
        for g in range(1, G):
            offs[g].append(o)
            if not (lo[g] <= o < lo[g] + wd[g]) or not valid[i + g]: break
            o += cnt[i + g]; r += 1
        else:
            offs[G].append(o)
        retired.append(r); i += r
    return np.mean(retired), offs


def main():
    rng = np.random.default_rng(1)
    for prec in (1.0, 3.0, 10.0):
        cnt, valid = coordinates(prec, 120000, rng)
        c64, offs = rounds(cnt, valid, LO, WD)
        # second wave: levels 5, 6, 7 with 21 + 21 + 22 lanes, each window placed on the offsets' own distribution given the path got there
        lo, wd = list(LO), list(WD)
        sim_cnt = cnt
        for g, width in ((5, 21), (6, 21), (7, 22)):
            _, o2 = rounds(sim_cnt, valid, lo + [0], wd + [10 ** 6])          # an unbounded window at level g: where do the paths arrive?
            arr = np.array(o2[g])
            best = max(range(int(arr.min()), int(arr.max())), key=lambda a: np.sum((arr >= a) & (arr < a + width)))
            lo.append(best); wd.append(width)
        c128, _ = rounds(cnt, valid, lo, wd)
        # all 64 lanes of the second wave on ONE more level (what a "wider late window" buys at most)
        print("precision %4.1f: draws per coordinate mean %.2f sd %.2f, speculative-valid %.3f | coordinates per round: 64 hypotheses %.2f, 128 hypotheses (levels 5-7 at %s, widths %s) %.2f = x%.3f"
              % (prec, cnt.mean(), cnt.std(), valid.mean(), c64, lo[5:], wd[5:], c128, c128 / c64))


if __name__ == "__main__":
    main()
