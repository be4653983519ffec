"""Monte-Carlo of the "second wave that meets the first once per block, not once per round" schemes -- the
price of using the issue room a second wave finds on the SIMD (N = 2048 runs at 1.42x the replica-steps/s of N = 1024: each of two waves
sharing a SIMD runs at 0.71 of a lone wave's speed) for ONE replica's sweep, before any hand-over cost.

The sweep is the pointer chase o_{c+1} = o_c + n_c(x_c, o_c) (with the tree-free threshold the update of coordinate c is a pure
function of its own value and of the stream offset it starts at).  A second wave B cannot know o_c ahead of the first wave A -- but two
chases over the same coordinates that ever stand on the same (c, o) stay together for good, so B may chase from a GUESSED offset some
coordinates ahead and A adopts B's work from the first coordinate at which its true offset coincides with the one B had there:

  scheme "ahead":   B starts G coordinates ahead of A at the expected offset (A's offset + G x the mean draws per coordinate), runs the
                    same chase at the same speed and records o_B(c).  A checks o_A(c) == o_B(c) at every coordinate it retires; on the
                    first match it jumps to B's front (everything B did since is the true path) and B restarts G ahead of the new front
                    with an exact base.  If A has not merged T coordinates after B's start, B gives up and restarts from A's position.
                    While B waits (lead limit reached) A runs alone at full speed.
  scheme "band":    B evaluates, for the coordinates G .. G + K ahead, the update for EVERY offset of a band of width W around the expected
                    one (a table n_c(o), 64 entries per instruction like A's own hypotheses); A walks through the table instead of
                    computing when its offset lies in the band.  Costs B W / 64 "coordinate evaluations" per coordinate and band.

Draws: a real counter-based stream (position k -> one exponential, one uniform), the reference's coordinate update (SliceSampler.jl:97-186)
on the toy MVN path in units of the chain's standard deviation (slice { v : v^2 < x^2 + 2 E / prec }, w = 10 sqrt(prec)), so the number of
draws an update consumes depends on (x_c, o) exactly as on the device: mean 6.3, sd 3.1 at the default parameters.
Time unit: one coordinate retired by a lone wave.  Overheads NOT charged: the LDS traffic of publishing / checking o_B(c), barriers, the
refills both waves need, the second wave's prologue -- the numbers are upper bounds.  Usage: python tools/sim_coalesce.py"""
import sys
import numpy as np

SPEED2 = 0.71            # speed of each of two waves sharing a SIMD relative to a lone wave (profiles/r04_nchains.txt: 1.96 M vs 1.38 M / 2)
EVAL64 = 3.2             # cost of evaluating 64 hypotheses (head, doubling, shrinkage: ~240 of a round's 301 instructions) in the time unit: a round
                         # retires ~4 coordinates, so a round is 4 units and its evaluation part 3.2; a table LOOKUP costs A the chase hop only
HOP = 0.2                # ... ~60 of 301 instructions per 4 coordinates


class Stream:
    def __init__(self, n, prec, seed):
        rng = np.random.default_rng(seed)
        self.E = rng.exponential(size=n); self.U = rng.random(size=n)
        self.w = 10.0 * np.sqrt(prec)
        self.memo = {}

    def update(self, c, x, o):
        """draws consumed by the update of coordinate c (value x) started at stream offset o"""
        key = (c, o)
        r = self.memo.get(key)
        if r is not None:
            return r
        E, U, w = self.E, self.U, self.w
        Q = x * x + 2.0 * E[o]
        L = x - w * U[o + 1]; R = L + w
        k = o + 2
        kd = 0
        while kd < 20 and (L * L < Q or R * R < Q):
            if U[k] <= 0.5: L -= (R - L)
            else: R += (R - L)
            k += 1; kd += 1
        while True:
            v = L + U[k] * (R - L); k += 1
            if v * v < Q: break
            if v < x: L = v
            else: R = v
        self.memo[key] = k - o
        return k - o


def solo_stats(st, xs):
    o = 0; cnt = []
    for c, x in enumerate(xs):
        n = st.update(c, x, o); cnt.append(n); o += n
    return np.array(cnt)


def sim_ahead(st, xs, mean_n, G, T, lead_max):
    """-> (speed-up over a lone wave, fraction of B's evaluations that ended on the true path, merges per 1000 coordinates)"""
    n = len(xs) - 4 * max(G, lead_max) - 8
    t = 0.0
    cA, oA = 0, 0
    useful_B = work_B = merges = 0

    def spawn():
        c0 = cA + G
        return {"c0": c0, "c": c0, "o": oA + int(round(mean_n * G)), "path": {}, "alive": True}
    B = spawn()
    while cA < n:
        b_runs = B["alive"] and (B["c"] - cA) < lead_max
        dt = 1.0 / SPEED2 if b_runs else 1.0
        t += dt
        # A retires one coordinate
        if cA in B["path"] and B["path"][cA] == oA:
            # merged: everything B did from here on is the true path
            useful_B += B["c"] - cA; merges += 1
            cA, oA = B["c"], B["o"]
            B = spawn()
            continue                                    # (the jump itself is free; the tick paid for the check)
        oA += st.update(cA, xs[cA], oA); cA += 1
        if b_runs:
            B["path"][B["c"]] = B["o"]
            B["o"] += st.update(B["c"], xs[B["c"]], B["o"]); B["c"] += 1; work_B += 1
        if B["alive"] and cA >= B["c0"] + T:          # no merge T coordinates behind B's start: B's path is written off
            B = spawn()
    return n / t, useful_B / max(work_B, 1), 1000.0 * merges / n


def sim_band(st, xs, mean_n, sd_n, G, K, W):
    """B tabulates n_c(o) for c in [front + G, front + G + K) and the W offsets centred on the expected one; A reads the table when its
    offset is inside (cost: the chase hop), computes itself otherwise.  B's cost per (coordinate, band) = W / 64 evaluations of 64
    hypotheses (EVAL64 each, as in A's own round).  A and B share the SIMD while B works."""
    n = len(xs) - 4 * (G + K) - 8
    t = 0.0; cA, oA = 0, 0; hits = evals = 0
    while cA < n:
        # B builds the table for [cA + G, cA + G + K) while A walks the G coordinates up to it (and those of the table it misses)
        base = oA + int(round(mean_n * G))
        b_work = K * W / 64.0 * EVAL64                   # in units of a lone wave's time per coordinate
        # A walks G coordinates itself, sharing the SIMD with B for as long as B has work
        for _ in range(G):
            share = min(1.0, b_work)                     # B's remaining work during this coordinate
            t += share / SPEED2 + (1.0 - share); b_work -= share * 1.0
            oA += st.update(cA, xs[cA], oA); cA += 1
        t += b_work / 1.0                                # B not finished: A waits (B alone at full speed)
        for j in range(K):
            exp_o = base + int(round(mean_n * j))
            if abs(oA - exp_o) <= W // 2:
                hits += 1                                # table hit: the coordinate costs A the chase hop
                t += HOP; oA += st.update(cA, xs[cA], oA); cA += 1
            else:
                t += 1.0; oA += st.update(cA, xs[cA], oA); cA += 1
            evals += 1
    return n / t, hits / max(evals, 1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    print("# speed of each of two waves sharing a SIMD: %.2f of a lone wave (two waves together: %.2fx); no hand-over cost charged" % (SPEED2, 2 * SPEED2))
    for prec in (1.0, 10.0):
        rng = np.random.default_rng(7)
        xs = rng.standard_normal(n)
        st = Stream(8 * n + 4096, prec, seed=11)
        cnt = solo_stats(st, xs)
        mean_n, sd_n = cnt.mean(), cnt.std()
        print("\nprecision %.0f: draws per coordinate mean %.2f sd %.2f" % (prec, mean_n, sd_n))
        print("scheme 'ahead' (B chases from a guessed offset G coordinates ahead; A adopts B's path where the offsets coincide)")
        print("%6s %6s %6s | %8s %12s %14s" % ("G", "T", "lead", "speed-up", "B useful", "merges / 1000"))
        best = (0, None)
        for G in (4, 8, 16, 32, 64, 128, 256):
            for T in (G // 2, G, 2 * G, 4 * G, 10 ** 9):
                for lead in (2 * G, 4 * G):
                    s, u, m = sim_ahead(st, xs, mean_n, G, T, lead)
                    if s > best[0]: best = (s, (G, T, lead, u, m))
                    if T in (G, 10 ** 9) and lead == 2 * G:
                        print("%6d %6s %6d | %8.3f %12.3f %14.1f" % (G, "inf" if T > 10 ** 8 else T, lead, s, u, m))
        print("best of the grid: speed-up %.3f at G, T, lead = %s (B useful %.3f)" % (best[0], best[1][:3], best[1][3]))
        print("scheme 'band' (B tabulates n_c(o) for K coordinates, G ahead, W offsets wide)")
        print("%6s %6s %6s | %8s %10s" % ("G", "K", "W", "speed-up", "hit rate"))
        bestb = (0, None)
        for G in (4, 8, 16):
            for K in (4, 8, 16):
                for W in (16, 32, 64, 128):
                    s, h = sim_band(st, xs, mean_n, sd_n, G, K, W)
                    if s > bestb[0]: bestb = (s, (G, K, W, h))
        for G, K, W in ((4, 4, 32), (4, 8, 64), (8, 8, 64), (8, 16, 128), (16, 16, 128)):
            s, h = sim_band(st, xs, mean_n, sd_n, G, K, W)
            print("%6d %6d %6d | %8.3f %10.3f" % (G, K, W, s, h))
        print("best of the grid: speed-up %.3f at G, K, W = %s (hit rate %.3f)" % (bestb[0], bestb[1][:3], bestb[1][3]))


if __name__ == "__main__":
    main()
