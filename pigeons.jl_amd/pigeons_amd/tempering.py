"""Per-round boundary of the hot path: Schedule, schedule adaptation, barriers, stepping stone.

Host-side mirror of reference src/schedules/Schedule.jl, src/tempering/adaptation.jl:56-112,
src/tempering/NonReversiblePT.jl:39-74, src/evidence/stepping_stone.jl:9-43.  Runs once per
round on O(N) doubles; stays on the host exactly as in the reference.
"""
import math

import numpy as np


class Schedule:
    """A partition of [0, 1] (src/schedules/Schedule.jl:5-30)."""

    def __init__(self, grids, check=True):
        grids = np.asarray(grids, dtype=np.float64).copy()
        if not check:                                   # per-chain view of a two-leg ladder (0 -> 1 -> 0)
            self.grids = grids
            return
        if len(grids) == 1:
            assert grids[0] == 1.0
        else:
            ok = (np.all(np.diff(grids) > 0) and grids[0] == 0.0 and grids[-1] == 1.0)
            if not ok:
                raise AssertionError("Invalid schedule: %s" % grids)
        self.grids = grids

    def n_chains(self):
        return len(self.grids)


def equally_spaced_schedule(n_chains):
    """src/schedules/Schedule.jl:36-44 (the range's elements are i/(n-1))."""
    assert n_chains >= 1
    if n_chains == 1:
        return Schedule([1.0])
    g = [i / (n_chains - 1) for i in range(n_chains)]
    g[-1] = 1.0
    return Schedule(g)


class FritschCarlsonMonotonicInterpolation:
    """Interpolations.jl `interpolate(x, y, FritschCarlsonMonotonicInterpolation())`
    (third-party dependency of the reference, compat 0.14-0.16; call sites
    src/tempering/adaptation.jl:61,85)."""

    def __init__(self, x, y):
        x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
        n = len(x)
        D = np.zeros(n - 1); m = np.zeros(n)
        for k in range(n - 1):
            D[k] = (y[k + 1] - y[k]) / (x[k + 1] - x[k])
            if k == 0:
                m[k] = D[k]
            elif D[k - 1] * D[k] <= 0.0:
                m[k] = 0.0
            else:
                m[k] = (D[k - 1] + D[k]) / 2.0
        m[n - 1] = D[n - 2]
        for k in range(n - 1):
            if D[k] == 0.0:
                m[k] = 0.0; m[k + 1] = 0.0
                continue
            a = m[k] / D[k]; b = m[k + 1] / D[k]
            den = math.sqrt(a * a + b * b)
            tau = 3.0 / den if den != 0.0 else math.inf      # Julia: 3.0/0.0 == Inf
            if tau < 1.0:
                m[k] = tau * a * D[k]; m[k + 1] = tau * b * D[k]
        c = np.zeros(n - 1); d = np.zeros(n - 1)
        for k in range(n - 1):
            xd = x[k + 1] - x[k]
            c[k] = (3.0 * D[k] - 2.0 * m[k] - m[k + 1]) / xd
            d[k] = (m[k] + m[k + 1] - 2.0 * D[k]) / (xd * xd)
        self.x, self.y, self.m, self.c, self.d = x, y, m, c, d

    def _interval(self, t):
        k = int(np.searchsorted(self.x, t, side="left"))   # searchsortedfirst, 0-based
        if k > 0:
            k -= 1
        return min(k, len(self.x) - 2)

    def __call__(self, t):
        k = self._interval(t)
        xd = t - self.x[k]
        return self.y[k] + self.m[k] * xd + self.c[k] * xd * xd + self.d[k] * xd * xd * xd

    def gradient(self, t):
        k = self._interval(t)
        xd = t - self.x[k]
        return self.m[k] + 2.0 * self.c[k] * xd + 3.0 * self.d[k] * xd * xd


def rejections(swap_acceptance_mean, swap_acceptance_n):
    """src/tempering/adaptation.jl:109-112: 1 - mean acceptance, default 0.5 for absent keys."""
    m = np.asarray(swap_acceptance_mean, dtype=np.float64)
    n = np.asarray(swap_acceptance_n)
    return 1.0 - np.where(n > 0, m, 0.5)


def optimal_schedule_generator(intensity, old_schedule, nudged=False):
    """src/tempering/adaptation.jl:67-81."""
    intensity = np.asarray(intensity, dtype=np.float64)
    assert len(old_schedule) == len(intensity) + 1
    assert np.all(intensity >= 0.0), "Bad intensities: %s" % intensity
    x = np.zeros(len(intensity) + 1)
    acc = 0.0
    for i, r in enumerate(intensity):
        acc += r
        x[i + 1] = acc
    x = x / x[-1]
    if len(np.unique(x)) != len(x):
        assert not nudged
        return optimal_schedule_generator(intensity + 1e-6, old_schedule, True)
    return FritschCarlsonMonotonicInterpolation(x, old_schedule)


def optimal_schedule(intensity, old_schedule, new_schedule_n_chains):
    """src/tempering/adaptation.jl:83-88."""
    gen = optimal_schedule_generator(intensity, old_schedule)
    n = new_schedule_n_chains
    return np.array([0.0] + [gen(i / (n - 1)) for i in range(1, n - 1)] + [1.0])


class CommunicationBarriers:
    """src/tempering/adaptation.jl:56-65."""

    def __init__(self, intensity, schedule):
        intensity = np.asarray(intensity, dtype=np.float64)
        assert len(schedule) == len(intensity) + 1
        y = np.zeros(len(schedule))
        acc = 0.0
        for i, r in enumerate(intensity):
            acc += r
            y[i + 1] = acc
        self.cumulativebarrier = FritschCarlsonMonotonicInterpolation(schedule, y)
        self.globalbarrier = acc

    def localbarrier(self, beta):
        return self.cumulativebarrier.gradient(beta)


def stepping_stone_pair(lsr_up, lsr_up_n, lsr_dn, lsr_dn_n):
    """src/evidence/stepping_stone.jl:28-43."""
    e1 = 0.0
    e2 = 0.0
    for v, n in zip(lsr_up, lsr_up_n):
        if n > 0:
            e1 += v - math.log(n)
    for v, n in zip(lsr_dn, lsr_dn_n):
        if n > 0:
            e2 += v - math.log(n)
    return (e1, -e2)


def stepping_stone(pair):
    """src/evidence/stepping_stone.jl:9-18."""
    if not math.isfinite(pair[0]):
        return pair[1]
    if not math.isfinite(pair[1]):
        return pair[0]
    return (pair[0] + pair[1]) / 2.0
