"""CPU tests of the oracle against every known answer the reference's own tests hold for this
path plus the public SplitMix64 vector.  No GPU needed."""
import math

import numpy as np
import pytest

import oracle as O


def test_splitmix64_public_vector():
    """Public SplitMix64 test vector (seed 1234567, golden gamma) == Java SplittableRandom.nextLong."""
    r = O.OracleRng(1234567)
    assert [r.next_u64() for _ in range(5)] == [
        6457827717110365317, 3203168211198807973, 9817491932198370423,
        4593380528125082431, 16408922859458223821]


def test_split_streams_disjoint():
    """reference test/test_split.jl:1-19 (split_slice: disjoint / overlapping slices)."""
    def helper(lo, hi):
        master = O.OracleRng(1)
        out = []
        for i in range(1, hi + 1):
            c = master.split()
            if i >= lo:
                out.append(c.rand())
        return out
    assert len(set(helper(1, 10) + helper(11, 20))) == 20
    assert len(set(helper(1, 15) + helper(10, 20))) == 20


def test_rand_is_in_unit_interval_with_52_bits():
    r = O.OracleRng(7)
    v = np.array([r.rand() for _ in range(20000)])
    assert v.min() >= 0.0 and v.max() < 1.0
    assert np.all(v * 2.0 ** 52 == np.floor(v * 2.0 ** 52))
    assert abs(v.mean() - 0.5) < 0.01


def test_ziggurat_table_pins():
    """The four entries of Julia's tables recalled in SURVEY.md App. B."""
    assert int(O.zig_table("ki", derived=True)[0]) == 0x0007799ec012f7b2 and int(O.zig_table("ke", derived=True)[0]) == 0x000e290a13924be3
    assert O.zig_table("wi", derived=True).view(np.float64)[0] == 1.7367254121602630e-15
    assert O.zig_table("we", derived=True).view(np.float64)[0] == 1.9311480126418366e-15


def test_randn_randexp_moments():
    r = O.OracleRng(3)
    z = np.array([r.randn() for _ in range(200000)])
    e = np.array([r.randexp() for _ in range(200000)])
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02
    assert abs(np.mean(z ** 4) - 3.0) < 0.15            # tails (slow paths) right
    assert abs(np.mean(np.abs(z) > 3.6541528853610088) - 2.5837e-4) < 1.2e-4
    assert abs(e.mean() - 1.0) < 0.01 and abs(e.var() - 1.0) < 0.03 and e.min() >= 0.0
    assert abs(np.mean(e > 7.69711747013104972) - math.exp(-7.69711747013104972)) < 2.5e-4


def test_log_sum_kat():
    """reference test/test_log_sum.jl:1-19."""
    L = O.lib()
    v1 = L.po_logaddexp(L.po_logaddexp(-math.inf, 2.1), 4.0)
    assert math.isclose(v1, math.log(math.exp(2.1) + math.exp(4)), rel_tol=1e-14)
    assert math.isclose(L.po_logaddexp(v1, 50.1), math.log(math.exp(v1) + math.exp(50.1)), rel_tol=1e-14)
    assert L.po_logaddexp(-math.inf, -math.inf) == -math.inf


@pytest.mark.parametrize("d", [1, 2, 3, 7, 64, 100, 1024])
def test_sqr_norm_tree(d):
    L = O.lib()
    x = np.random.default_rng(d).standard_normal(d)
    got = L.po_sqr_norm(O._dp(x), d)
    assert math.isclose(got, float(np.sum(x * x)), rel_tol=1e-13)
    # explicit tree
    P = 1
    while P < d:
        P *= 2
    a = np.zeros(P); a[:d] = x * x
    while len(a) > 1:
        a = a[0::2] + a[1::2]
    assert got == a[0]


def test_round_trips_kat():
    """reference test/test_round_trips.jl:1-14: TestSwapper(1.0), N=4, R=5 => 13."""
    n_chains, n_rounds = 4, 5
    pt = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=1.0, n_chains=n_chains, explorer=O.EXPLORER_NONE)
    for _ in range(n_rounds):
        pt.run_round()
    truth = sum(math.floor(max(2 ** n_rounds - i, 0) / n_chains / 2) for i in range(n_chains))
    assert pt.round_trip()[1] == truth == 13


@pytest.mark.parametrize("explorer", [O.EXPLORER_SLICE, O.EXPLORER_TOY])
def test_stepping_stone_kat(explorer):
    """reference test/test_stepping_stone.jl:15-27: |error| < 0.2 for toy_mvn_target(10), N=6, R=12."""
    pt = O.OraclePT(dim=10, n_chains=6, explorer=explorer)
    for _ in range(12):
        pt.run_round()
    truth = 0.5 * 10 * (math.log(1.0) - math.log(10.0))      # ScaledPrecisionNormalPath.jl:66-71
    p = pt.stepping_stone_pair()
    assert abs(p[0] - truth) < 0.2 and abs(p[1] - truth) < 0.2


def test_cumulative_barrier_kat():
    """reference test/test_cumulative_barrier.jl:1-11 (d=2, N=10, R=15; closed form
    ScaledPrecisionNormalPath.jl:56-64).  The reference asserts < 0.01 on ITS seed-1 stream; over
    seeds 1..8 this restatement gives a finite-N bias of -0.005 +- 0.003, so the bound here is 0.015."""
    pt = O.OraclePT(dim=2, n_chains=10, explorer=O.EXPLORER_SLICE)
    for _ in range(15):
        pt.run_round()
    for beta in np.arange(0.0, 1.0001, 0.1):
        truth = math.log(math.sqrt((1 - beta) * 1.0 + beta * 10.0))      # d = 2: 2^(2-d)/B(1,1) = 1
        assert abs(pt.cumulative_barrier(float(beta)) - truth) < 0.015


def test_target_moments_kat():
    """reference test/test_moments.jl:1-27: target chain mean ~ 0, variance ~ 0.1 within 0.03."""
    pt = O.OraclePT(dim=3, n_chains=10, explorer=O.EXPLORER_SLICE, record_online=1)
    for _ in range(12):
        pt.run_round()
    m, v, n = pt.online()
    assert n == 2 ** 12
    assert np.all(np.abs(m) < 0.03) and np.all(np.abs(v - 0.1) < 0.03)


def test_report_shape_and_counts():
    """reference test/test_apis.jl:12-20 shape: N=10 => 9 swap pairs; each pair active every other scan."""
    pt = O.OraclePT(dim=2, n_chains=10, explorer=O.EXPLORER_TOY)
    for r in range(1, 7):
        pt.run_round()
        m, n = pt.swap_pr()
        assert len(m) == 9 and np.all(n == 2 ** (r - 1))
        up, un, dn, dnn = pt.log_sum_ratio()
        assert np.all(un == 2 ** (r - 1)) and np.all(dnn == 2 ** (r - 1))
        ip = pt.index_process()
        assert ip.shape == (10, 2 ** r)
        assert np.all(np.abs(np.diff(ip, axis=1)) <= 1)
        assert np.array_equal(np.sort(ip, axis=0), np.tile(np.arange(10)[:, None], (1, 2 ** r)))
        g = pt.schedule()
        assert g[0] == 0.0 and g[-1] == 1.0 and np.all(np.diff(g) > 0)


def test_deo_partner_structure_from_index_process():
    """DEO: scan 1 of every round uses the ODD graph (pairs (1,2),(3,4).. 1-based), src/swap/DEO.jl:12."""
    pt = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=1.0, n_chains=6, explorer=O.EXPLORER_NONE)
    pt.run_round()                                     # 2 scans
    ip = pt.index_process()
    # with acceptance 1: after odd scan chains (0,1),(2,3),(4,5) swap; recorded chain is pre-swap
    assert list(ip[:, 0]) == [0, 1, 2, 3, 4, 5]
    assert list(ip[:, 1]) == [1, 0, 3, 2, 5, 4]
    pt.run_round()                                     # round 2 starts again with the odd graph
    ip = pt.index_process()
    # state before round 2: after even scan of round 1 (pairs (1,2),(3,4) 0-based)
    assert list(ip[:, 0]) == [2, 0, 4, 1, 5, 3]


def test_threads_do_not_change_results():
    """Parallelism invariance (reference src/pt/checks.jl): OpenMP threads vs serial, bit-identical."""
    a = O.OraclePT(dim=12, n_chains=7, explorer=O.EXPLORER_SLICE, n_threads=1)
    b = O.OraclePT(dim=12, n_chains=7, explorer=O.EXPLORER_SLICE, n_threads=4)
    for _ in range(4):
        a.run_round(); b.run_round()
    xa, ca, ra = a.states(); xb, cb, rb = b.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ra, rb)
    assert np.array_equal(a.index_process(), b.index_process())


def test_traces_online_and_energy_ac1_recorders():
    """SURVEY.md 8(f) rank 1.  Properties the reference's own tests assert (test/test_traces.jl:7-66):
    sample matrix of the last round is (n_scans, d + 1); the mean of the traced marginal equals the online
    mean (atol 1e-10); plus internal consistency of the restated OnlineStatsBase CovMatrix(2)."""
    N, d = 6, 3
    pt = O.OraclePT(n_chains=N, dim=d, seed=1, explorer=O.EXPLORER_SLICE, record_online=1, record_traces=1, record_energy_ac1=1)
    for r in range(7):
        pt.run_round()
    tr = pt.traces()
    assert tr.shape == (2 ** 7, d + 1)
    np.testing.assert_allclose(tr[:, -1], -0.5 * 10.0 * np.sum(tr[:, :-1] ** 2, axis=1), rtol=1e-13)   # log_density column
    m, v, n = pt.online()
    assert n == 2 ** 7
    np.testing.assert_allclose(m, tr[:, :-1].mean(axis=0), atol=1e-10)                                  # test_traces.jl:55
    np.testing.assert_allclose(v, tr[:, :-1].var(axis=0, ddof=1), rtol=1e-10)
    lm, lv, ln = pt.online_lp()
    assert ln == 2 ** 7 and abs(lm - tr[:, -1].mean()) < 1e-10 and abs(lv - tr[:, -1].var(ddof=1)) < 1e-9
    cor, cn, raw = pt.energy_ac1()
    assert np.array_equal(cn, np.full(N, 2 ** 7))
    assert np.all(np.abs(cor) <= 1.0) and abs(cor[0]) < 0.35          # the reference chain is refreshed i.i.d.
    # "after" at the target chain is the traced log density: its running mean is the CovMatrix's b[2]
    assert abs(raw[N - 1, 1] - tr[:, -1].mean()) < 1e-10


def test_energy_ac1_merge_matches_numpy_corrcoef():
    """GroupBy(Int, CovMatrix(2)) merged over replicas == the plain correlation of all (before, after) pairs at a chain."""
    pt = O.OraclePT(n_chains=1, dim=4, seed=3, explorer=O.EXPLORER_SLICE, record_traces=1, record_energy_ac1=1)
    for r in range(8):
        prev_last = None
        pt.run_round()
    tr = pt.traces()                                  # single chain: after_t == before_{t+1}
    cor, cn, raw = pt.energy_ac1()
    after = tr[:, -1]
    np.testing.assert_allclose(raw[0, 1], after.mean(), rtol=1e-12)
    # before_t = after_{t-1} for t >= 1 within the round; the first "before" is the last "after" of the previous round
    b = after[:-1]; a = after[1:]
    approx = np.corrcoef(b, a)[0, 1]
    assert abs(cor[0] - approx) < 0.05


@pytest.mark.parametrize("kw", [dict(explorer=O.EXPLORER_MALA, am_step_size=0.3),
                                dict(explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA),
                                dict(explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_MALA, am_step_size=0.3)])
def test_mala_and_compose_target_moments_and_logz(kw):
    """SURVEY.md 8(f) rank 2.  Same analytic checks the reference applies to its explorers
    (test/test_stepping_stone.jl:15-27, test/test_moments.jl): toy MVN, d = 4, target N(0, I/10)."""
    d, N = 4, 6
    pt = O.OraclePT(n_chains=N, dim=d, seed=1, record_online=1, **kw)
    for _ in range(11):
        pt.run_round()
    m, v, n = pt.online()
    assert n == 2 ** 11
    assert np.all(np.abs(m) < 0.05) and np.all(np.abs(v - 0.1) < 0.02)
    truth = 0.5 * d * math.log(1.0 / 10.0)         # log Z1/Z0 of exp(-prec/2 |x|^2), prec 1 -> 10
    p = pt.stepping_stone_pair()
    assert abs(p[0] - truth) < 0.2 and abs(p[1] - truth) < 0.2


def test_two_references_double_the_restarts_kat():
    """reference test/test_variational.jl:44-57: TestSwapper(0.5), n_chains = 5, 15 rounds, seed 1 -- a second leg of
    5 chains (StabilizedPT with variational = nothing) doubles the tempered restarts: |2 - ratio| <= 0.05."""
    def restarts(**kw):
        pt = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=0.5, explorer=O.EXPLORER_NONE, seed=1, record_index_process=0, **kw)
        for _ in range(15):
            pt.run_round()
        return pt.round_trip()[0]
    assert abs(2.0 - restarts(n_chains=5, n_chains_variational=5) / restarts(n_chains=5)) <= 0.05


def test_two_leg_tempering_adapts_both_legs():
    """reference test/test_two_legs.jl:19-27: both legs' global barriers agree with the one-leg barrier (rtol 0.1 there,
    on a Turing target; here the analytic toy MVN, whose barrier is known: test_cumulative_barrier_kat)."""
    one = O.OraclePT(n_chains=8, dim=4, explorer=O.EXPLORER_TOY)
    two = O.OraclePT(n_chains=8, n_chains_variational=7, dim=4, explorer=O.EXPLORER_TOY)
    for _ in range(11):
        one.run_round(); two.run_round()
    b1, b2f, b2v = one.global_barrier(), two.global_barrier(), two.global_barrier_variational()
    assert abs(b2f - b1) / b1 < 0.1 and abs(b2v - b1) / b1 < 0.1
    s = two.schedule()
    assert s[0] == 0.0 and s[6] == 1.0 and s[7] == 1.0 and s[-1] == 0.0          # references at both ends, targets in the middle
    assert np.all(np.diff(s[:7]) > 0) and np.all(np.diff(s[7:]) < 0)
    truth = 0.5 * 4 * math.log(1.0 / 10.0)
    p = two.stepping_stone_pair()                                                   # variational leg only
    assert abs(p[0] - truth) < 0.2 and abs(p[1] - truth) < 0.2


def test_gaussian_reference_on_the_funnel():
    """GaussianReference (src/variational/GaussianReference.jl): activates at first_tuning_round, is refitted from the
    target chains' online mean / std, and -- being a normalised density like the funnel -- brings log(Z1/Z0) to 0
    (the stepping-stone check the reference applies in test/test_stepping_stone.jl:4-13 to its Turing target)."""
    pt = O.OraclePT(n_chains=8, dim=2, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, am_preconditioner=2,
                    variational_first_tuning_round=4, record_online=1)
    for r in range(1, 12):
        pt.run_round()
        v = pt.variational()
        assert (v is None) == (r < 4)
    m, var, n = pt.online()
    np.testing.assert_allclose(v[0], m, rtol=1e-13); np.testing.assert_allclose(v[1], np.sqrt(var), rtol=1e-13)
    assert abs(v[0][0]) < 0.6 and abs(v[1][0] - 3.0) < 0.6                  # z[1] ~ Normal(0, 3)
    p = pt.stepping_stone_pair()
    assert abs(0.5 * (p[0] + p[1])) < 0.25


@pytest.mark.parametrize("w,p,n_passes", [(10.0, 20, 3), (1.0, 20, 2), (0.2, 20, 2), (0.05, 8, 1), (100.0, 20, 2)])
def test_slice_accept_never_rejects_on_the_mvn_path(w, p, n_passes):
    """The device's speculative rounds do not execute the acceptance check of the doubling scheme (SliceSampler.jl:192-237):
    on the scaled-precision MVN path the slice is an interval for the floating-point predicate too, so the check cannot reject
    (pigeons.jl_amd/csrc/pte_slice8.hpp, DESIGN.md 5 round 3 c).  The oracle DOES execute it, records 1.0 / 0.0 per call as the
    reference does (explorer_acceptance_pr) -- its mean must be exactly 1 on every chain, also where almost every interval is
    doubled (w far below the slice width) and where the doubling budget p is hit."""
    for N, d, seed in [(6, 40, 1), (12, 7, 2), (4, 300, 3)]:
        pt = O.OraclePT(n_chains=N, dim=d, seed=seed, explorer=O.EXPLORER_SLICE, slice_w=w, slice_p=p, slice_n_passes=n_passes)
        for _ in range(4):
            pt.run_round()
            am, an, ss, sn = pt.explorer_stats()
            assert np.all(an[1:] > 0) and np.all(am[1:] == 1.0), (w, p, N, d, am)
