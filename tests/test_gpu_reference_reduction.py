"""PTE_RECORD_REFERENCE_REDUCTION (VERDICT r04 "missing" item 4): swap_acceptance_pr and log_sum_ratio reduced the way the reference
reduces them -- every replica's own Mean (mu += (x - mu) / n) and LogSum fitted in scan order, merged over the binary tree on the replica
index (src/recorders/recorders.jl:88-130, src/recorders/LogSum.jl:1-24, src/mpi_utils/Entangler.jl:188-251) -- instead of the device's
chain-keyed sums.  By default the two agree to ~1e-12 (tests/test_gpu_parity.py asserts 1e-9); with the flag the recorders, and therefore
the adapted schedule, every later state and the stepping-stone estimate, equal the oracle's BIT FOR BIT wherever the states do
(SliceSampler / toy explorer on the scaled-precision path, Ising: the log ratios are then the same doubles)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _exact_round(P, pt, ref):
    assert P.next_round(pt)
    red = P.run_one_round(pt)
    P.adapt(pt, red)
    ref.run_round()
    assert np.array_equal(red.index_process, ref.index_process())
    m, n = red.swap_acceptance_pr
    mr, nr = ref.swap_pr()
    assert np.array_equal(n, nr)
    assert np.array_equal(m, mr), np.abs(m - mr).max()
    up, un, dn, dnn = red.log_sum_ratio
    upr, unr, dnr, dnnr = ref.log_sum_ratio()
    assert np.array_equal(un, unr) and np.array_equal(dnn, dnnr)
    assert np.array_equal(up, upr) and np.array_equal(dn, dnr), (np.abs(up - upr).max(), np.abs(dn - dnr).max())
    assert np.array_equal(pt.shared.tempering.schedule.grids, ref.schedule()), np.abs(pt.shared.tempering.schedule.grids - ref.schedule()).max()
    if pt.inputs.n_chains > 1:
        assert np.array_equal(P.stepping_stone_pair(pt), ref.stepping_stone_pair())
        assert P.global_barrier(pt) == ref.global_barrier()
    return red


@pytest.mark.parametrize("two_launches", [False, True])          # the fused scan loop's hand-shake and k_swap both write the log
@pytest.mark.parametrize("kind,N,d,rounds,seed", [
    ("slice", 6, 10, 8, 1),        # reference test_stepping_stone.jl shape
    ("slice", 7, 64, 7, 2),        # odd N: the tree's right edge climbs alone
    ("slice", 12, 100, 6, 1),
    ("slice", 33, 7, 8, 4),        # 2^5 + 1 replicas: the last one merges at the top only
    ("slice", 2, 33, 6, 5),        # one pair
    ("toy", 16, 128, 7, 3),
    ("toy", 37, 5, 8, 9),
    ("slice", 64, 8, 10, 6),       # 2046 scans: every replica visits most pairs; the last round replays 1024 scans
])
def test_recorders_and_schedule_equal_the_oracle_bit_for_bit(P, kind, N, d, rounds, seed, two_launches):
    from pigeons_amd import _lib
    exp = {"toy": P.ToyExplorer(), "slice": P.SliceSampler()}[kind]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp, seed=seed,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False),
              reference_reduction=True, debug_kernel=_lib.KERNEL_TWO_LAUNCHES if two_launches else 0)
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, explorer={"toy": O.EXPLORER_TOY, "slice": O.EXPLORER_SLICE}[kind])
    for _ in range(rounds):
        _exact_round(P, pt, ref)
    x, chain, rng = pt.replicas.states()
    xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)


@pytest.mark.parametrize("kind,N,d,rounds,seed", [("slice", 6, 10, 8, 1), ("slice", 33, 7, 7, 4), ("toy", 16, 128, 7, 3), ("automala", 8, 64, 6, 2), ("compose", 6, 12, 6, 5)])
def test_energy_ac1_equals_the_oracle_bit_for_bit(P, kind, N, d, rounds, seed):
    """energy_ac1 (recorder.jl:113: CovMatrix(2) of the log density before / after explore!, src/pt/pigeons.jl:133-143) under the flag: the pair of every
    chain and scan is logged by k_log_energy around the explorer kernels and replayed with OnlineStats' arithmetic per replica, merged over the replica
    tree -- the correlation energy_ac1s reports EQUALS the oracle's (round 6, late).  Runs the launch-per-scan loop (the log is written between launches)."""
    ex = {"toy": P.ToyExplorer(), "slice": P.SliceSampler(), "automala": P.AutoMALA(), "compose": P.Compose(P.AutoMALA(), P.SliceSampler())}[kind]
    okw = {"toy": dict(explorer=O.EXPLORER_TOY), "slice": dict(explorer=O.EXPLORER_SLICE), "automala": dict(explorer=O.EXPLORER_AUTOMALA, am_preconditioner=2),
           "compose": dict(explorer=O.EXPLORER_AUTOMALA, explorer2=O.EXPLORER_SLICE, am_preconditioner=2)}[kind]
    langevin = kind in ("automala", "compose")
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1] + ([P.online, P.traces] if langevin else [])
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=ex, seed=seed, record=rec, show_report=False), reference_reduction=True)
    assert pt.replicas.scan_loop_name() == ""
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, record_energy_ac1=1, **okw, **(dict(record_online=1, record_traces=1) if langevin else {}))
    for _ in range(rounds):
        red = _exact_round(P, pt, ref)
        cor, cn, _ = red.energy_ac1
        corr, cnr, _ = ref.energy_ac1()
        assert np.array_equal(cn, cnr) and np.array_equal(cor, corr, equal_nan=True), np.nanmax(np.abs(cor - corr))
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(x, xr) and np.array_equal(chain, cr) and np.array_equal(rng, rr)


@pytest.mark.parametrize("kw", [dict(), dict(device_messages=True), dict(transport="group")])
@pytest.mark.parametrize("kind,N,d,G,rounds", [("slice", 8, 40, 2, 7), ("slice", 12, 100, 4, 6), ("toy", 9, 5, 3, 8), ("slice", 16, 30, 8, 6), ("toy", 6, 3, 6, 7)])
def test_chain_shards_replay_their_own_pairs(P, kind, N, d, G, rounds, kw):
    """a chain-shard holds the log and the index_process rows of the pairs whose lower chain it owns -- everything the replay of those pairs
    needs; the merge tree runs over the GLOBAL replica index, so G shards give the oracle's numbers exactly like one engine does"""
    exp = {"toy": P.ToyExplorer(), "slice": P.SliceSampler()}[kind]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp, seed=4, show_report=False,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces, P.energy_ac1]), n_shards=G, reference_reduction=True, **kw)
    ref = O.OraclePT(n_chains=N, dim=d, seed=4, explorer={"toy": O.EXPLORER_TOY, "slice": O.EXPLORER_SLICE}[kind], record_online=1, record_traces=1, record_energy_ac1=1)
    for _ in range(rounds):
        red = _exact_round(P, pt, ref)
        assert np.array_equal(red.energy_ac1[1], ref.energy_ac1()[1]) and np.array_equal(red.energy_ac1[0], ref.energy_ac1()[0], equal_nan=True)
        if np.array_equal(red.traces, ref.traces()):
            assert np.array_equal(red.online[0], ref.online()[0]) and np.array_equal(red.online[1], ref.online()[1])
    assert pt.shards.n_boundary_swaps > 0


@pytest.mark.parametrize("kind,nf,nv,d,rounds", [("slice", 6, 5, 4, 8), ("slice", 4, 4, 70, 6), ("toy", 3, 7, 9, 7)])
def test_two_legs(P, kind, nf, nv, d, rounds):
    """StabilizedPT (two legs sharing the target, src/tempering/StabilizedPT.jl): two target chains fit the same online statistics, the
    pair between the legs never swaps; both barriers and the concatenated schedule equal the oracle's"""
    ex = {"slice": P.SliceSampler(), "toy": P.ToyExplorer()}[kind]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=nf, n_chains_variational=nv, variational=None, n_rounds=rounds, explorer=ex,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces], show_report=False), reference_reduction=True)
    ref = O.OraclePT(n_chains=nf, n_chains_variational=nv, dim=d, explorer={"slice": O.EXPLORER_SLICE, "toy": O.EXPLORER_TOY}[kind],
                     record_online=1, record_traces=1)
    for _ in range(rounds):
        red = _exact_round(P, pt, ref)
        assert P.global_barrier_variational(pt) == ref.global_barrier_variational()
        if np.array_equal(red.traces, ref.traces()):
            assert np.array_equal(red.online[0], ref.online()[0]) and np.array_equal(red.online[1], ref.online()[1])


def test_ising(P):
    L, N, rounds = 8, 9, 8
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(0.7, L), n_chains=N, n_rounds=rounds, seed=3,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), reference_reduction=True)
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=0.7, n_chains=N, seed=3, slice_n_passes=3)
    for _ in range(rounds):
        _exact_round(P, pt, ref)


@pytest.mark.parametrize("extended", [False, True])
@pytest.mark.parametrize("kind,N,d,rounds,seed", [("slice", 6, 10, 8, 1), ("slice", 13, 70, 6, 2), ("toy", 9, 65, 7, 3)])
def test_online_statistics_replayed_from_the_traces(P, kind, N, d, rounds, seed, extended):
    """:online with :traces recorded: the target chain's Mean / Variance per coordinate (and of the log density) are rebuilt from the traced
    samples per replica and tree-merged -- equal to the oracle's wherever the traced samples are (they are bit-identical except where a
    ziggurat slow path went through libm vs ocml)"""
    exp = {"toy": P.ToyExplorer(), "slice": P.SliceSampler()}[kind]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp, seed=seed, extended_traces=extended,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces], show_report=False), reference_reduction=True)
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, explorer={"toy": O.EXPLORER_TOY, "slice": O.EXPLORER_SLICE}[kind], record_online=1,
                     record_traces=2 if extended else 1)
    exact_rounds = 0
    for _ in range(rounds):
        red = _exact_round(P, pt, ref)
        om, ov, on = red.online
        omr, ovr, onr = ref.online()
        assert on == onr
        if np.array_equal(red.traces, ref.traces()):
            exact_rounds += 1
            assert np.array_equal(om, omr) and np.array_equal(ov, ovr), (np.abs(om - omr).max(), np.abs(ov - ovr).max())
            lm, lv, ln = ref.online_lp()
            assert tuple(red.online_log_density) == (lm, lv)
        else:
            np.testing.assert_allclose(om, omr, rtol=1e-9, atol=1e-12)
    assert exact_rounds >= rounds - 1


def test_default_reduction_stays_within_its_tolerance_of_the_replayed_one(P):
    """the same run with and without the flag: integers equal, floats 1e-10 apart at most (and NOT all equal: the flag does something)"""
    N, d, rounds = 24, 20, 8
    mk = lambda rr: P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=P.SliceSampler(), seed=11,
                                  record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), reference_reduction=rr)
    a, b = mk(False), mk(True)
    differs = False
    for _ in range(rounds):
        ra, rb = P.run_one_round(a) if P.next_round(a) else None, P.run_one_round(b) if P.next_round(b) else None
        P.adapt(a, ra); P.adapt(b, rb)
        assert np.array_equal(ra.index_process, rb.index_process)
        assert np.array_equal(ra.swap_acceptance_pr[1], rb.swap_acceptance_pr[1])
        np.testing.assert_allclose(ra.swap_acceptance_pr[0], rb.swap_acceptance_pr[0], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(ra.log_sum_ratio[0], rb.log_sum_ratio[0], rtol=1e-10)
        np.testing.assert_allclose(a.shared.tempering.schedule.grids, b.shared.tempering.schedule.grids, rtol=1e-9)
        differs = differs or not np.array_equal(ra.swap_acceptance_pr[0], rb.swap_acceptance_pr[0])
    assert differs


@pytest.mark.parametrize("form", ["wg", "one_chain", "two_launches"])
@pytest.mark.parametrize("kind,N,d,rounds,seed", [("automala", 6, 10, 7, 1), ("automala", 8, 128, 6, 1), ("automala", 5, 64, 6, 2), ("automala", 13, 20, 7, 3),
                                                   ("mala", 6, 64, 6, 4), ("compose", 6, 12, 6, 5), ("sharded", 8, 10, 6, 6)])
def test_automala_on_the_mvn_path_equals_the_oracle_bit_for_bit(P, kind, N, d, rounds, seed, form):
    """AutoMALA / MALA on the scaled-precision path hold no transcendental in the state's arithmetic, so their kernels ARE bit-exact -- what
    used to put an ulp between engine and oracle from round 2-4 on was the step size, adapted on the mean of am_factors (AutoMALA.jl:70-79),
    and the preconditioner, refitted to the target chain's online variance.  With the flag both are reduced with the reference's arithmetic
    (am_factors: the exponents of every step-size search logged and replayed; online: from the traces) and every state word, the step size
    and the schedule equal the oracle's in every round"""
    from pigeons_amd import _lib
    flags = {"wg": 0, "one_chain": _lib.KERNEL_SCAN_LOOP_ONE_CHAIN, "two_launches": _lib.KERNEL_TWO_LAUNCHES}[form]
    if kind in ("compose", "sharded") and form != "wg":
        pytest.skip("Compose and chain-shards run the launch-per-scan loop whatever the flag says")
    ex = {"automala": P.AutoMALA(), "sharded": P.AutoMALA(), "mala": P.MALA(step_size=0.15), "compose": P.Compose(P.AutoMALA(), P.SliceSampler())}[kind]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=ex, seed=seed,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces], show_report=False), reference_reduction=True, debug_kernel=flags,
              **(dict(n_shards=2, transport="group") if kind == "sharded" else {}))
    okw = {"automala": dict(explorer=O.EXPLORER_AUTOMALA), "sharded": dict(explorer=O.EXPLORER_AUTOMALA), "mala": dict(explorer=O.EXPLORER_MALA, am_step_size=0.15),
           "compose": dict(explorer=O.EXPLORER_AUTOMALA, explorer2=O.EXPLORER_SLICE)}[kind]
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, am_preconditioner=2, record_online=1, record_traces=1, **okw)
    for _ in range(rounds):
        red = _exact_round(P, pt, ref)
        ex_now = pt.shared.explorer.first if kind == "compose" else pt.shared.explorer
        assert ex_now.step_size == ref.step_size()
        assert np.array_equal(red.traces, ref.traces())
        assert np.array_equal(red.online[0], ref.online()[0]) and np.array_equal(red.online[1], ref.online()[1])
        fm, fn, rm, rn = ref.automala_stats()
        assert np.array_equal(red.am_factors[1], fn) and np.array_equal(red.am_factors[0], fm)
        if kind in ("automala", "sharded"):            # reversibility_rate (AutoMALA.jl:294) out of the same log: EQUAL too (round 6; nothing adapts on it)
            assert np.array_equal(red.reversibility_rate[1], rn) and np.array_equal(red.reversibility_rate[0], rm), (red.reversibility_rate[0] - rm)
        x, chain, rng = (pt.shards if pt.shards is not None else pt.replicas).states(); xr, cr, rr = ref.states()
        assert np.array_equal(chain, cr) and np.array_equal(rng, rr) and np.array_equal(x, xr)


def test_automala_on_the_funnel_stays_inside_its_tolerance(P):
    """the funnel's density holds exp / log (ocml vs glibc: states an ulp apart from round 1), so there is nothing exact to reproduce; the
    replay must still agree with the oracle inside the parity tolerance"""
    N, d, rounds = 8, 8, 6
    pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d), n_chains=N, n_rounds=rounds, explorer=P.AutoMALA(), seed=1,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), reference_reduction=True)
    ref = O.OraclePT(n_chains=N, dim=d, seed=1, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1 / 9., am_preconditioner=2)
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        np.testing.assert_allclose(red.swap_acceptance_pr[0], ref.swap_pr()[0], rtol=1e-6, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-6)
        np.testing.assert_allclose(pt.shared.explorer.step_size, ref.step_size(), rtol=1e-9)


def test_the_flag_has_its_preconditions(P):
    from pigeons_amd import _lib
    from pigeons_amd.engine import Engine
    with pytest.raises(Exception, match="INDEX_PROCESS"):
        Engine(n_chains=4, dim=3, record_flags=_lib.RECORD_REFERENCE_REDUCTION, explorer=_lib.EXPLORER_SLICE)
