"""SURVEY.md 8 row a8, last sub-row: SliceSampler's Bool / Integer coordinate methods (/root/reference/src/explorers/SliceSampler.jl:65-86,
136-142, 189) in the ORACLE -- the device has no target with such coordinates and refuses them (tested below), so this is CPU only.

What pins the restatement here (no Julia in the image, and the reference's tests hold no vector for these methods -- `parity unpinned`, as
for the rest of the oracle):
  * rand(rng, a:b): the published SamplerRangeNDL algorithm held against exact integer arithmetic, its power-of-two and full-range closed
    forms, and uniformity;
  * the DRAW ORDER of each method, replayed by hand from a twin generator on log potentials where the outcome has a closed form;
  * the number of density evaluations the reference's comments promise (Bool: ONE per coordinate);
  * the Float64 method of the mixed-state step against the oracle's own (pinned-by-KAT) Float64 SliceSampler, bit for bit;
  * invariance: the Bool method leaves a product of Bernoullis invariant, the Integer method a Binomial (sanity band, see the test).
"""
import math

import numpy as np
import pytest

import oracle as O

U64 = 1 << 64


def exact_rand_range(u64_stream, a, b):
    """Random.SamplerRangeNDL on Int64 in unbounded integers: s = b - a + 1 (mod 2^64); m = x s; reject while (m mod 2^64) < (2^64 - s) mod s"""
    s = (b - a + 1) % U64
    x = next(u64_stream)
    if s == 0:
        r = x
    else:
        m = x * s
        if m % U64 < s:
            t = (U64 - s) % s
            while m % U64 < t:
                x = next(u64_stream)
                m = x * s
        r = m >> 64
    v = (r + a) % U64
    return v - U64 if v >= 1 << 63 else v


def u64s(rng):
    while True:
        yield rng.next_u64()


RANGES = [(0, 10), (0, 0), (-5, 5), (0, 1), (3, 1023), (0, 2 ** 20 - 1), (-(2 ** 62), 2 ** 62), (-(2 ** 63), 2 ** 63 - 1), (-(2 ** 63), 0),
          (0, 2 ** 63 - 1), (1, 3 * 2 ** 61), (7, 7 + 2 ** 63), (-17, 2 ** 33 + 5)]


@pytest.mark.parametrize("a,b", RANGES)
def test_rand_range_equals_exact_arithmetic(a, b):
    for seed in (1, 2, 12345):
        r, twin = O.OracleRng(seed=seed), O.OracleRng(seed=seed)
        stream = u64s(twin)
        for _ in range(400):
            v = r.rand_range(a, b)
            assert a <= v <= b
            assert v == exact_rand_range(stream, a, b)
        assert r.state == twin.state                                # the same number of UInt64 draws were consumed


def test_rand_range_rejects_where_it_must():
    """s = 2^63 + 1: t = 2^63 - 1, about half the draws with a low product are redrawn -- the loop is exercised, not just compiled"""
    a, b = 7, 7 + 2 ** 63
    r, twin = O.OracleRng(seed=3), O.OracleRng(seed=3)
    for _ in range(2000):
        r.rand_range(a, b)
    n = 0
    while twin.state != r.state:
        twin.next_u64(); n += 1
        assert n < 10000
    assert n > 2300, n                                               # 2000 results took more than 2000 UInt64s


def test_rand_range_closed_forms():
    r, twin = O.OracleRng(seed=9), O.OracleRng(seed=9)
    for k in (1, 3, 8, 20, 63):                                      # a range of 2^k values takes the top k bits, never rejects
        for _ in range(200):
            assert r.rand_range(-4, -4 + 2 ** k - 1) == -4 + (twin.next_u64() >> (64 - k))
    for _ in range(200):                                             # typemin:typemax (s wraps to 0): x % Int64 + typemin, wrapping = x - 2^63
        assert r.rand_range(-(2 ** 63), 2 ** 63 - 1) == twin.next_u64() - 2 ** 63
    for _ in range(50):                                              # a:a draws once and returns a
        assert r.rand_range(5, 5) == 5
        twin.next_u64()
    assert r.state == twin.state


def test_rand_range_is_uniform():
    r = O.OracleRng(seed=2024)
    n = 110000
    counts = np.bincount([r.rand_range(0, 10) for _ in range(n)], minlength=11)
    chi2 = float(((counts - n / 11) ** 2 / (n / 11)).sum())
    assert chi2 < 35.0, (chi2, counts)                               # 10 degrees of freedom: P(chi2 > 35) ~ 1e-4


# ---- Bool: the full conditional ----------------------------------------------------------------------------------------------------------

def test_bool_method_draw_order_and_evaluation_count():
    """logit 0 on every coordinate: prob_zero = 1 / (1 + exp(0)) = 0.5 exactly, so coordinate c becomes  !(rand(rng) < 0.5)  of the c-th draw;
    ONE rand and ONE density evaluation per coordinate and pass (SliceSampler.jl:64), nothing recorded."""
    d, passes = 6, 3
    s = O.MixedSliceSampler(lambda x: 0.0, [O.COORD_BOOL] * d, n_passes=passes)
    r, twin = O.OracleRng(seed=5), O.OracleRng(seed=5)
    x = np.array([1.0, 0.0, 1.0, 1.0, 0.0, 0.0])
    for step in range(20):
        before = s.n_evals
        s.step(r, x)
        last = None
        for _ in range(passes):
            last = [0.0 if twin.rand() < 0.5 else 1.0 for _ in range(d)]
        assert list(x) == last
        assert s.n_evals - before == 1 + passes * d                  # cached_log_potential once, then one per coordinate
        assert r.state == twin.state
    assert s.stats.acc_n == 0 and s.stats.steps_n == 0


def test_bool_method_respects_a_constraint():
    """lp(false) = -Inf on coordinate 0: exp(lp1 - lp0) = Inf, prob_zero = 0, the coordinate stays true and the step never sees a
    non-finite cached density; started outside the support the step refuses as the reference does (:35-37)."""
    lp = lambda x: (0.0 if x[0] == 1.0 else -math.inf) + 0.3 * x[1]
    s = O.MixedSliceSampler(lp, [O.COORD_BOOL, O.COORD_BOOL])
    r = O.OracleRng(seed=1)
    x = np.array([1.0, 0.0])
    for _ in range(200):
        s.step(r, x)
        assert x[0] == 1.0 and x[1] in (0.0, 1.0)
    with pytest.raises(RuntimeError, match="initialized outside the support"):
        s.step(r, np.array([0.0, 0.0]))


def test_bool_method_leaves_bernoullis_invariant():
    logit = np.array([-2.0, -0.5, 0.0, 1.0, 3.0])
    p = 1 / (1 + np.exp(-logit))
    s = O.MixedSliceSampler(lambda x: float(logit @ x), [O.COORD_BOOL] * 5, n_passes=1)
    r = O.OracleRng(seed=77)
    x = np.zeros(5)
    n = 20000
    tot = np.zeros(5)
    for _ in range(n):
        tot += s.step(r, x)
    z = (tot / n - p) / np.sqrt(p * (1 - p) / n)                     # every update is an exact, independent draw of the conditional
    assert np.all(np.abs(z) < 4.5), z


# ---- Integer: slicing on the lattice -----------------------------------------------------------------------------------------------------

def test_integer_method_draw_order_on_a_flat_density():
    """Flat lp = 0 and p = 0 (no doubling allowed; with p > 0 a flat density doubles p times, both ends being inside the slice): z = 0 -
    randexp(rng) < 0 = lp everywhere, the first rand(rng, L:R) is taken (n = 1) and slice_accept's loop does not run (R - L = 10 <= 1.1 w):
        randexp;  L = x - rand(rng, 0:10), R = L + 10;  x' = rand(rng, L:R)       -- three draws per coordinate, in that order."""
    s = O.MixedSliceSampler(lambda x: 0.0, [O.COORD_INTEGER] * 3, p=0, n_passes=2)
    r, twin = O.OracleRng(seed=11), O.OracleRng(seed=11)
    x = np.array([0.0, -40.0, 1000.0])
    want = [int(v) for v in x]
    for step in range(50):
        before = s.n_evals
        s.step(r, x)
        for _ in range(2):
            for c in range(3):
                e = twin.randexp()
                assert e > 0.0
                L = want[c] - twin.rand_range(0, 10)
                want[c] = twin.rand_range(L, L + 10)
        assert [int(v) for v in x] == want and np.all(x == np.round(x))
        assert r.state == twin.state
        assert s.n_evals - before == 1 + 2 * 3 * 3                   # lp(L), lp(R), lp(new) per coordinate
    assert s.stats.steps_n == 50 * 2 * 3 * 2 and s.stats.steps_sum == 50 * 2 * 3 * 1.0   # (0 doublings) + (n = 1) per coordinate
    assert s.stats.acc_n == 50 * 2 * 3 and s.stats.acc_mean == 1.0


def test_integer_method_doubles_and_shrinks_on_the_lattice():
    """A wide plateau: lp = 0 on |k| <= 300, -Inf outside.  From 0 with w = 10 both ends start inside (z < 0 = lp(L), lp(R)), so the
    interval doubles until an end leaves the plateau; every intermediate position handed to the density is an integer, the accepted
    point is on the plateau, and explorer_n_steps sees the doublings."""
    seen = []

    def lp(x):
        seen.append(float(x[0]))
        return 0.0 if abs(x[0]) <= 300 else -math.inf
    s = O.MixedSliceSampler(lp, [O.COORD_INTEGER], n_passes=1)
    r = O.OracleRng(seed=4)
    x = np.array([0.0])
    for _ in range(300):
        s.step(r, x)
        assert abs(x[0]) <= 300
    assert all(v == round(v) for v in seen)
    assert s.stats.steps_sum / s.stats.steps_n > 2.0                 # ~5-6 doublings + >= 1 shrink draw per update
    assert max(abs(v) for v in seen) > 300                           # the ends did leave the plateau


def test_integer_method_needs_an_integral_width():
    s = O.MixedSliceSampler(lambda x: 0.0, [O.COORD_INTEGER], w=2.5)
    with pytest.raises(RuntimeError, match="for integer variables, the width should be an integer. Got: 2.5"):   # the @assert of :137
        s.step(O.OracleRng(seed=1), np.array([0.0]))
    s = O.MixedSliceSampler(lambda x: 0.0, [O.COORD_INTEGER], w=3.0, p=0, n_passes=1)       # width = ceil(Int, 3.0): rand(rng, 0:3), R = L + 3
    r, twin = O.OracleRng(seed=1), O.OracleRng(seed=1)
    x = s.step(r, np.array([5.0]))
    twin.randexp()
    L = 5 - twin.rand_range(0, 3)
    assert x[0] == twin.rand_range(L, L + 3) and r.state == twin.state


def test_integer_method_on_a_binomial():
    """Sanity band, not a theorem about the reference's lattice method: 40000 updates of k ~ Binomial(20, 0.3) stay on {0..20} and land
    within 0.02 total variation of the pmf (the closed intervals of the lattice method share their mid-points between the two halves
    slice_accept tests, so exact invariance is the reference's claim to make, not this file's)."""
    n_, p_ = 20, 0.3
    logpmf = np.array([math.lgamma(n_ + 1) - math.lgamma(k + 1) - math.lgamma(n_ - k + 1) + k * math.log(p_) + (n_ - k) * math.log(1 - p_) for k in range(n_ + 1)])

    def lp(x):
        k = int(x[0])
        return float(logpmf[k]) if 0 <= k <= n_ else -math.inf
    s = O.MixedSliceSampler(lp, [O.COORD_INTEGER], n_passes=1)
    r = O.OracleRng(seed=31)
    x = np.array([6.0])
    counts = np.zeros(n_ + 1)
    n = 40000
    for _ in range(n):
        s.step(r, x)
        assert 0 <= x[0] <= n_ and x[0] == round(x[0])
        counts[int(x[0])] += 1
    tv = 0.5 * np.abs(counts / n - np.exp(logpmf)).sum()
    assert tv < 0.02, tv


# ---- mixed states and the Float64 method --------------------------------------------------------------------------------------------------

def test_float64_kind_is_the_oracles_float64_slice_sampler():
    """The mixed-state step with every coordinate Float64 and the MVN path's log potential behind the call-back IS the SliceSampler the
    rest of the oracle runs (the one the device kernels are held to): same states bit for bit, replica by replica, scan after scan."""
    L = O.lib()
    N, d = 6, 5
    ref = O.OraclePT(n_chains=N, dim=d, seed=3, explorer=O.EXPLORER_SLICE)
    p0, p1 = ref.cfg.p0, ref.cfg.p1
    ref.begin_round()
    steps_total = 0.0
    for scan in range(4):
        x, chain, rng = ref.states()
        betas = ref.schedule()
        ref.run_scans(1)
        x_after, _, _ = ref.states()
        for i in range(N):
            if chain[i] == 0:
                continue                                             # the reference chain draws iid instead (src/pt/pigeons.jl:80-89)
            beta = betas[chain[i]]
            prec = (1.0 - beta) * p0 + beta * p1

            def lp(v, prec=prec):
                v = np.ascontiguousarray(v)
                return (-0.5 * prec) * L.po_sqr_norm(O._dp(v), d)
            s = O.MixedSliceSampler(lp, [O.COORD_FLOAT64] * d, w=ref.cfg.slice_w, p=ref.cfg.slice_p, n_passes=ref.cfg.slice_n_passes,
                                    max_iter=ref.cfg.slice_max_iter)
            xi = x[i].copy()
            s.step(O.OracleRng(state=tuple(int(v) for v in rng[i])), xi)
            assert np.array_equal(xi, x_after[i]), (scan, i)
            steps_total += s.stats.steps_sum
    ref.end_round()
    _, _, steps_sum, _ = ref.explorer_stats()
    assert steps_total == float(np.sum(steps_sum)) and steps_total > 0


def test_mixed_state_dispatches_per_coordinate():
    """[Float64, Integer, Bool] with independent N(0, 1) x Binomial(8, 1/2) x Bernoulli(0.7): every coordinate keeps its type, the
    moments land where they should."""
    lb = [math.lgamma(9) - math.lgamma(k + 1) - math.lgamma(9 - k) for k in range(9)]

    def lp(x):
        k = int(x[1])
        if not 0 <= k <= 8:
            return -math.inf
        return -0.5 * x[0] * x[0] + lb[k] + x[2] * math.log(0.7 / 0.3)
    s = O.MixedSliceSampler(lp, [O.COORD_FLOAT64, O.COORD_INTEGER, O.COORD_BOOL], n_passes=1)
    r = O.OracleRng(seed=8)
    x = np.array([0.1, 4.0, 0.0])
    n = 20000
    acc = np.zeros((n, 3))
    for i in range(n):
        acc[i] = s.step(r, x)
    assert np.all(acc[:, 1] == np.round(acc[:, 1])) and acc[:, 1].min() >= 0 and acc[:, 1].max() <= 8
    assert set(np.unique(acc[:, 2])) <= {0.0, 1.0}
    assert len(np.unique(acc[:, 0])) > n * 0.99                      # the Float64 coordinate is continuous
    m = acc.mean(axis=0)
    assert abs(m[0]) < 0.05 and abs(acc[:, 0].var() - 1.0) < 0.08
    assert abs(m[1] - 4.0) < 0.08 and abs(acc[:, 1].var() - 2.0) < 0.15
    assert abs(m[2] - 0.7) < 0.02
    with pytest.raises(RuntimeError, match="kind 7"):
        O.MixedSliceSampler(lp, [0, 7, 2]).step(r, x)


# ---- tier 3: a live Pigeons.jl (tools/gen_golden.jl) -----------------------------------------------------------------------------------------
import json, os
_REF_PATH = os.path.join(O.ROOT, "tests", "golden", "reference_pigeons.json")
_REF = json.load(open(_REF_PATH)) if os.path.exists(_REF_PATH) else None
needs_reference = pytest.mark.skipif(_REF is None or "rand_range" not in _REF, reason="tests/golden/reference_pigeons.json holds no rand_range / slice_* "
                                     "records (tools/gen_golden.jl needs Julia + Pigeons.jl): these methods stay unpinned against the live reference")


@needs_reference
def test_rand_range_against_live_reference():
    for entry in _REF["rand_range"]:
        r = O.OracleRng(entry["seed"]).split()
        for (a, b), draws in zip(entry["ranges"], entry["draws"]):
            got = [r.rand_range(int(a), int(b)) for _ in draws]
            bad = [i for i, (g, w) in enumerate(zip(got, draws)) if g != int(w)]
            assert not bad, "rand(rng, %s:%s), seed %d: draw %d is %d here, %s in Julia" % (a, b, entry["seed"], bad[0], got[bad[0]], draws[bad[0]])
        assert [str(v) for v in r.state] == entry["final_rng"]


_GOLDEN_LPS = {     # the same expressions, term by term, as golden_lp_* in tools/gen_golden.jl
    "slice_integer": (lambda x: -0.125 * ((x[0] - 3.0) * (x[0] - 3.0) + (x[1] + 2.0) * (x[1] + 2.0)), [O.COORD_INTEGER] * 2, [0.0, 0.0]),
    "slice_bool": (lambda x: (0.5 if x[0] else 0.0) - (1.25 if x[1] else 0.0) + (0.75 if (x[0] and x[2]) else 0.0), [O.COORD_BOOL] * 3, [0.0, 1.0, 0.0]),
    "slice_mixed": (lambda x: -0.5 * (x[0] * x[0]) - 0.125 * ((x[1] - 3.0) * (x[1] - 3.0)) + (0.75 if x[2] else 0.0),
                    [O.COORD_FLOAT64, O.COORD_INTEGER, O.COORD_BOOL], [0.25, 3.0, 1.0]),
}


@needs_reference
@pytest.mark.parametrize("name", sorted(_GOLDEN_LPS))
def test_slice_methods_against_live_reference(name):
    if name not in _REF:
        pytest.skip("the fixture has no %s record (%s)" % (name, _REF.get("slice_mixed_error", "not generated")))
    lp, kinds, x0 = _GOLDEN_LPS[name]
    g = _REF[name]
    s = O.MixedSliceSampler(lp, kinds)                               # SliceSampler() defaults, as the generator
    r = O.OracleRng(g["seed"]).split()
    x = np.array(x0)
    for i, want in enumerate(g["states"]):
        s.step(r, x)
        w = np.array([int(b) for b in want], dtype=np.uint64).view(np.float64)
        assert np.array_equal(x, w), "%s: state after step %d is %r here, %r in Julia" % (name, i + 1, x, w)
    assert [str(v) for v in r.state] == g["final_rng"]


# ---- the device keeps refusing ------------------------------------------------------------------------------------------------------------

def test_device_refuses_slice_sampling_of_bool_coordinates():
    """pte_create validates before it touches a device, so the refusal is testable here: spins are Bool coordinates."""
    import sys, os
    sys.path.insert(0, os.path.join(O.ROOT, "pigeons.jl_amd"))
    import pigeons_amd as P
    from pigeons_amd import _lib
    for kw in (dict(explorer=_lib.EXPLORER_SLICE), dict(explorer=_lib.EXPLORER_ISING_METROPOLIS, explorer2=_lib.EXPLORER_SLICE)):
        with pytest.raises(P.PteError, match="SliceSampler's Bool / Integer coordinate methods are not available on the device"):
            P.Engine(n_chains=4, dim=16, target=_lib.TARGET_ISING, target_params=[0.3], **kw)
    with pytest.raises(P.PteError, match="Bool / Integer"):
        P.PT(P.Inputs(target=P.IsingLogPotential(0.3, 4), explorer=P.SliceSampler(), n_chains=4, show_report=False))
    with pytest.raises(NotImplementedError, match="no device log-potential; use the reference CPU path"):
        P.PT(P.Inputs(target=lambda x: 0.0, explorer=P.SliceSampler(), n_chains=4, show_report=False))
