"""k_explore_langevin_mw (round 6): AutoMALA / MALA at 512 < d <= 1024 with four wavefronts per replica.

Reference procedure: src/explorers/AutoMALA.jl:84-275, src/explorers/MALA.jl:74-97, src/explorers/hamiltonian_dynamics.jl:40-84,
src/explorers/Preconditioner.jl:57-77.  The kernel must be the SAME function as the one-wave kernel it replaces (sixteen 64-coordinate
blocks per lane, kept in the test build behind PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE): same fixed reduction tree -- a wave's four blocks are
one subtree, the top two levels are added by every wave from the partial sums exchanged through LDS --, same stream positions, same
branches.  So everything is compared bit for bit: states, RNG counters, chains, index process, every recorder, the adapted step size,
preconditioner statistics and schedule; and against the oracle on top (integers exact, floats 1e-9 on the MVN path, 1e-6 on the funnel:
ocml vs glibc exp / log)."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import _mk_am, _check_am_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _inputs(P, path, N, d, rounds, seed, explorer, record=None, **kw):
    rec = record or [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1]
    if path == "mvn":
        return P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=explorer, seed=seed, record=rec, show_report=False, **kw)
    return P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N, n_rounds=rounds,
                    explorer=explorer, seed=seed, record=rec, show_report=False, **kw)


def _run(P, inp_fn, rounds, flags):
    pt = P.PT(inp_fn(), debug_kernel=flags)
    out = []
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        row = [red.index_process.copy(), np.array(red.round_trip), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(), red.log_sum_ratio[2].copy(),
               red.explorer_n_steps[0].copy(), red.explorer_n_steps[1].copy(), red.explorer_acceptance_pr[0].copy(), red.explorer_acceptance_pr[1].copy(),
               red.am_factors[0].copy(), red.am_factors[1].copy(), red.reversibility_rate[0].copy(), np.array(pt.shared.tempering.schedule.grids).copy(),
               np.array(red.online[0]).copy(), np.array(red.online[1]).copy(), np.array(red.energy_ac1[2]).copy()]
        if red.traces is not None and np.size(red.traces):
            row.append(np.array(red.traces).copy())
        out.append(row)
    name = pt.replicas.kernel_name()
    st = [np.array(a).copy() for a in pt.replicas.states()]
    return name, out, st


def _same(a, b):
    assert len(a) == len(b)
    for r, (ra, rb) in enumerate(zip(a, b)):
        assert len(ra) == len(rb)
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y, equal_nan=True), (r, k, np.max(np.abs(np.asarray(x, dtype=float) - np.asarray(y, dtype=float))))


CASES = [
    ("mvn", 5, 1024, 4, 1, "automala"),        # whole blocks (the mask-free instantiation)
    ("mvn", 4, 600, 4, 2, "automala"),         # ragged: wave 2 holds a partial block, wave 3 nothing
    ("mvn", 3, 513, 3, 3, "automala"),         # one coordinate past the one-wave kernel's eight blocks
    ("mvn", 3, 1023, 3, 4, "automala"),
    ("mvn", 3, 769, 3, 9, "mala"),
    ("funnel", 4, 1024, 4, 1, "automala"),
    ("funnel", 3, 700, 4, 2, "automala"),
    ("funnel", 3, 1024, 3, 5, "mala"),
    ("funnel", 3, 832, 3, 6, "automala_identity"),
    ("mvn", 3, 896, 3, 7, "automala_diagonal"),
]


def _explorer(P, kind):
    return {"automala": lambda: P.AutoMALA(), "mala": lambda: P.MALA(step_size=0.02),
            "automala_identity": lambda: P.AutoMALA(preconditioner=P.IdentityPreconditioner()),
            "automala_diagonal": lambda: P.AutoMALA(preconditioner=P.DiagonalPreconditioner())}[kind]()


@pytest.mark.parametrize("path,N,d,rounds,seed,kind", CASES)
def test_four_waves_per_replica_equal_the_one_wave_kernel(P, path, N, d, rounds, seed, kind):
    from pigeons_amd import _lib
    mk = lambda: _inputs(P, path, N, d, rounds, seed, _explorer(P, kind))
    na, a, sa = _run(P, mk, rounds, _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)
    assert na.startswith("k_explore_automala [test build")
    nb, b, sb = _run(P, mk, rounds, _lib.KERNEL_TEST_BITS & 0)          # the default kernel of the PRODUCT library
    assert nb == "k_explore_langevin_mw"
    _same(a, b)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_four_waves_with_traces_compose_and_a_thousand_chains(P):
    """the epilogue's recorders read the row the four waves stored (traces of every chain); Compose(SliceSampler, AutoMALA) runs the kernel as the
    second explorer of a scan; N = 1024 at d = 1024: every compute unit holds four workgroups (the occupancy the kernel is built for)"""
    from pigeons_amd import _lib
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1, P.traces]
    mk = lambda: _inputs(P, "mvn", 6, 640, 3, 3, P.AutoMALA(), record=rec, extended_traces=True)
    _, a, sa = _run(P, mk, 3, _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)
    _, b, sb = _run(P, mk, 3, 0)
    _same(a, b)
    mk = lambda: _inputs(P, "mvn", 5, 700, 3, 4, P.Compose(P.SliceSampler(), P.AutoMALA()))
    _, a, sa = _run(P, mk, 3, _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)
    _, b, sb = _run(P, mk, 3, 0)
    _same(a, b)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    mk = lambda: _inputs(P, "funnel", 1024, 1024, 2, 1, P.AutoMALA(), record=[P.round_trip, P.index_process, P.log_sum_ratio])
    _, a, sa = _run(P, mk, 2, _lib.KERNEL_TEST_LANGEVIN_ONE_WAVE)
    _, b, sb = _run(P, mk, 2, 0)
    _same(a, b)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_four_waves_with_a_gaussian_reference(P):
    """StabilizedPT's variational leg at d > 512: the GaussianReference's constants are read per wave, its log density is one more cross-wave sum"""
    from pigeons_amd import _lib
    d = 600
    def mk():
        return P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=4, n_chains_variational=4, n_rounds=4,
                        variational=P.GaussianReference(first_tuning_round=2), explorer=P.AutoMALA(), seed=3,
                        record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1], show_report=False)
    pa = P.PT(mk(), debug_kernel=_lib.KERNEL_TEST_LANGEVIN_ONE_WAVE); pb = P.PT(mk())
    for _ in range(4):
        outs = []
        for pt in (pa, pb):
            assert P.next_round(pt)
            red = P.run_one_round(pt); P.adapt(pt, red)
            outs.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(), red.explorer_n_steps[0].copy(),
                         red.am_factors[0].copy(), [np.array(a).copy() for a in pt.replicas.states()]))
        for x, y in zip(outs[0][:-1], outs[1][:-1]):
            assert np.array_equal(x, y)
        for x, y in zip(outs[0][-1], outs[1][-1]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("path,N,d,rounds,seed", [("mvn", 3, 1024, 3, 1), ("mvn", 4, 600, 3, 5), ("funnel", 3, 700, 3, 2), ("funnel", 3, 1024, 3, 1)])
def test_four_waves_against_the_oracle(P, path, N, d, rounds, seed):
    pt, ref = _mk_am(P, N, d, rounds, path, seed=seed)
    assert pt.replicas.kernel_name() == "k_explore_langevin_mw"
    for _ in range(rounds):
        if path == "mvn":
            _check_am_round(P, pt, ref, rtol=1e-9)
        else:
            _check_am_round(P, pt, ref, rtol=1e-6, acc_rtol=1e-5, state_atol=1e-6)


def test_quotient_procedure_is_the_ieee_quotient():
    """pte_test_quotient: the quotient procedure of the Langevin-family kernels (Markstein's q' = fma(fma(-q, b, a), r, q) behind its guards) against
    numpy's a / b, bit for bit: 4 M random pairs over 600 binades, divisors the preconditioner and the funnel's sigma produce, and every edge the guards
    exist for -- zeros of both signs, subnormals, infinities, NaN, quotients that over / underflow, divisors with an all-ones significand or an extreme
    exponent.  The fast path must take the bulk (else the test tests nothing) and the edges must take the division."""
    import ctypes as C
    from pigeons_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(7)
    n = 1 << 22
    a = rng.standard_normal(n) * np.exp2(rng.integers(-300, 300, n).astype(np.float64))
    b = (1.0 + rng.random(n)) * np.exp2(rng.integers(-300, 300, n).astype(np.float64)) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
    b[: n // 4] = 1.0 / (1e-3 + 5.0 * rng.random(n // 4))                                   # 1 / sd and mix + (1 - mix) / sd
    b[n // 4: n // 2] = np.exp(rng.standard_normal(n // 4) * 3.0)                            # sigma = exp(y / 2)
    ones = np.uint64(0x000FFFFFFFFFFFFF)
    edges_a = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-310, np.inf, -np.inf, np.nan, 1.7e308, -1.7e308, 1e-300, 3.0, -3.0, 1.0, 2.0 ** -1000, 2.0 ** 1000])
    edges_b = np.array([1.0, 3.0, 1e-300, 1e300, 2.0 ** -600, 2.0 ** 600, 7.0, 0.1, (np.array([0x3FF0000000000000], dtype=np.uint64) | ones).view(np.float64)[0],
                        (np.array([0x4000000000000000], dtype=np.uint64) | ones).view(np.float64)[0], -2.5, 1e-320])
    ea, eb = np.meshgrid(edges_a, edges_b)
    a = np.concatenate([a, ea.ravel()]); b = np.concatenate([b, eb.ravel()])
    out = np.zeros(len(a)); took = np.zeros(len(a), dtype=np.int32)
    dp = C.POINTER(C.c_double)
    assert L.pte_test_quotient(0, a.ctypes.data_as(dp), b.ctypes.data_as(dp), len(a), out.ctypes.data_as(dp), took.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    with np.errstate(all="ignore"):
        want = a / b
    same = (out.view(np.uint64) == want.view(np.uint64)) | (np.isnan(out) & np.isnan(want))
    assert same.all(), (int((~same).sum()), a[~same][:5], b[~same][:5], out[~same][:5], want[~same][:5])
    assert took[:n].mean() < 0.02                                                            # the fast path takes the bulk
    assert took[n:].mean() > 0.5                                                             # the edges take the division
    assert (took[n:][(ea.ravel() == 0.0) | ~np.isfinite(ea.ravel())] == 1).all()             # zeros (of both signs), infinities, NaN: always
