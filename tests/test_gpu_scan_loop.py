"""The two forms of pte_run_scans -- the reference's `while next_scan!(pt)` loop, src/pt/pigeons.jl:46-55 -- are the same function.

Round 5: where one GPU holds the whole ladder and every workgroup is resident, all the scans of a call run as ONE kernel (k_scans_*:
workgroup c holds chain c, the DEO swap of a pair is a release / acquire hand-shake between its two waves, pte_kernels.hpp "ScanLoop");
elsewhere, or with PTE_KERNEL_TWO_LAUNCHES in pte_config.debug_kernel, every scan is an explore launch and a swap launch as in rounds
1-4 (k_swap: the pair's SwapStats exchanged by __shfl_xor).  Same arithmetic, same single rand(replica.rng) per replica and scan, same
recorder updates (src/swap/swap.jl:6-39, src/swap/pair_swapper.jl:42-88): everything must agree bit for bit -- index process, round trips,
swap / log-sum recorders, explorer statistics, online / traces / energy_ac1, the adapted schedule, states, chains, RNG counters.
Every oracle parity test of the SliceSampler on the MVN path runs the fused form (it is the default); this module holds it to the
launch-per-scan form at the shapes the oracle cannot afford, and checks that the choice is the documented one."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _grids(pt):
    t = pt.shared.tempering
    if hasattr(t, "schedule"):
        return np.array(t.schedule.grids).copy()
    return np.concatenate([np.array(t.variational_leg.schedule.grids), np.array(t.fixed_leg.schedule.grids)])      # StabilizedPT: both legs


def _run(P, N, d, rounds, seed, two_launches, record=None, nv=0, explorer=None):
    from pigeons_amd import _lib
    rec = record or [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1]
    kw = dict(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, seed=seed, explorer=explorer or P.SliceSampler(), record=rec, show_report=False)
    if nv:
        kw.update(n_chains_variational=nv)
    pt = P.PT(P.Inputs(**kw), debug_kernel=_lib.KERNEL_TWO_LAUNCHES if two_launches else 0)
    out = []
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        row = [red.index_process.copy(), np.array(red.round_trip), red.swap_acceptance_pr[0].copy(), red.swap_acceptance_pr[1].copy(),
               red.log_sum_ratio[0].copy(), red.log_sum_ratio[2].copy(), red.explorer_n_steps[0].copy(), red.explorer_acceptance_pr[0].copy(),
               _grids(pt)]
        if P.online in rec:
            row += [np.array(red.online[0]).copy(), np.array(red.online[1]).copy()]
        if P.energy_ac1 in rec:
            row += [np.array(red.energy_ac1[2]).copy()]
        if P.traces in rec:
            row += [np.array(red.traces).copy()]
        out.append(row)
    return pt, out


def _same(a, b):
    assert len(a) == len(b)
    for r, (ra, rb) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y, equal_nan=True), (r, k)


@pytest.mark.parametrize("N,d,rounds,seed,kernel", [
    (10, 2, 6, 1, "k_scans_slice8"),              # BASELINE configs[0]: launch bound in the two-launch form
    (7, 64, 5, 2, "k_scans_slice8"),              # odd N: the last chain is idle on the odd graph
    (2, 33, 6, 5, "k_scans_slice8"),              # reference + target only
    (1, 8, 4, 1, "k_scans_slice8"),               # a single chain: no partner ever
    (256, 1024, 3, 3, "k_scans_slice8"),          # BASELINE configs[1]
    (1024, 1024, 3, 1, "k_scans_slice8"),         # the metric configuration: one wave per SIMD
    (1024, 2048, 2, 4, "k_scans_slice8"),         # the longest rows the fused loop takes (16 KB)
    (1000, 300, 3, 6, "k_scans_slice8"),
])
def test_fused_scan_loop_equals_launch_per_scan(P, N, d, rounds, seed, kernel):
    pa, a = _run(P, N, d, rounds, seed, two_launches=True)
    assert pa.replicas.scan_loop_name() == ""
    sa = pa.replicas.states(); del pa
    pb, b = _run(P, N, d, rounds, seed, two_launches=False)
    assert pb.replicas.scan_loop_name() == kernel and pb.replicas.kernel_name().startswith("k_explore_slice8")
    limit, _, _ = pb.replicas.scan_loop_info()
    assert limit >= N
    sb = pb.replicas.states(); del pb
    _same(a, b)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_fused_scan_loop_with_traces_and_off_default_parameters(P):
    """the per-scan quantities a launch used to carry as arguments (trace row, index-process row) advance inside the kernel; the generic
    instantiation (p outside the FAST range) has its own fused kernel"""
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.traces, P.online, P.energy_ac1]
    for expl, name in ((P.SliceSampler(), "k_scans_slice8"), (P.SliceSampler(w=0.5, p=25, n_passes=2), "k_scans_slice8_generic")):
        pa, a = _run(P, 6, 70, 5, 3, True, record=rec, explorer=expl)
        pb, b = _run(P, 6, 70, 5, 3, False, record=rec, explorer=expl)
        assert pb.replicas.scan_loop_name() == name and pa.replicas.scan_loop_name() == ""
        _same(a, b)
        for x, y in zip(pa.replicas.states(), pb.replicas.states()):
            assert np.array_equal(x, y)


def test_fused_scan_loop_two_legs(P):
    """StabilizedPT: two reference chains, two targets in the middle -- the partner map is the same function of the global chain index"""
    pa, a = _run(P, 6, 20, 6, 2, True, nv=5)
    pb, b = _run(P, 6, 20, 6, 2, False, nv=5)
    assert pb.replicas.scan_loop_name() == "k_scans_slice8"
    _same(a, b)


def test_fused_scan_loop_against_the_oracle_with_split_calls(P):
    """pte_run_scans called piecewise (3 + 1 + 4 scans of one round: epochs carry over between launches, DEO parity follows the scan
    number) against the oracle's round"""
    N, d = 9, 40
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=3, seed=4, explorer=P.SliceSampler(), record=rec, show_report=False))
    ref = O.OraclePT(n_chains=N, dim=d, seed=4, explorer=O.EXPLORER_SLICE)
    e = pt.replicas
    assert e.scan_loop_name() == "k_scans_slice8"
    for r in (1, 2):                               # rounds 1-2 the usual way (schedule adaptation in between)
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        ref.run_round()
    e.run_scans(1, 3); e.run_scans(4, 1); e.run_scans(5, 4)          # round 3 = 8 scans, in three calls
    e.reduce()
    ref.run_round()
    assert np.array_equal(e.index_process(), ref.index_process())
    x, chain, rng = e.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-12, atol=0)


def _run_am(P, N, d, rounds, seed, two_launches, target="mvn", explorer=None, nv=0, flags=0):
    from pigeons_amd import _lib
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1]
    kw = dict(n_chains=N, n_rounds=rounds, seed=seed, explorer=explorer or P.AutoMALA(), record=rec, show_report=False)
    if target == "funnel":
        kw.update(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., d))
    else:
        kw.update(target=P.toy_mvn_target(d))
    if nv:
        kw.update(n_chains_variational=nv)
    pt = P.PT(P.Inputs(**kw), debug_kernel=(_lib.KERNEL_TWO_LAUNCHES if two_launches else 0) | flags)
    out = []
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        out.append([red.index_process.copy(), np.array(red.round_trip), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(), red.log_sum_ratio[2].copy(),
                    red.explorer_n_steps[0].copy(), red.explorer_acceptance_pr[0].copy(), np.array(red.am_factors[0]).copy(), np.array(red.reversibility_rate[0]).copy(),
                    _grids(pt), np.array(red.online[0]).copy(), np.array(red.online[1]).copy(), np.array(red.energy_ac1[2]).copy()])
    return pt, out


@pytest.mark.parametrize("target,N,d,rounds,seed,explorer", [
    ("funnel", 1024, 128, 3, 1, "automala"),      # BASELINE configs[2]
    ("funnel", 8, 70, 5, 2, "automala"),          # ragged second block
    ("mvn", 6, 10, 7, 1, "automala"),
    ("mvn", 5, 512, 3, 3, "automala"),            # E = 8: the largest register layout the fused loop takes
    ("mvn", 5, 64, 5, 4, "mala"),
    ("funnel", 6, 8, 6, 5, "mala"),
    ("mvn", 13, 20, 7, 6, "automala"),            # three full workgroups of four chains + one wave alone
    ("mvn", 2, 33, 6, 7, "automala"),             # one pair, one workgroup
    ("funnel", 37, 16, 6, 8, "automala"),
    ("mvn", 1, 8, 4, 9, "automala"),              # a single chain: nobody to shake hands with
])
def test_fused_langevin_scan_loop_equals_launch_per_scan(P, target, N, d, rounds, seed, explorer):
    """AutoMALA / MALA: the one-kernel scan loops (refreshes + pairwise swap hand-shake for all the scans of a call; the `scan != 1` rule of
    AutoMALA.jl:87,96-102 decided per scan inside the kernel) against the launch-per-scan loop, bit for bit -- both forms: k_scans_automala_wg
    (four consecutive chains per workgroup, three of four pairs shake hands through LDS; the default) and k_scans_automala (one chain per
    workgroup, every pair through the agent-scope hand-off)"""
    from pigeons_amd import _lib
    ex = (lambda: P.AutoMALA()) if explorer == "automala" else (lambda: P.MALA())
    pa, a = _run_am(P, N, d, rounds, seed, True, target, ex())
    pb, b = _run_am(P, N, d, rounds, seed, False, target, ex())
    pc, c = _run_am(P, N, d, rounds, seed, False, target, ex(), flags=_lib.KERNEL_SCAN_LOOP_ONE_CHAIN)
    assert pa.replicas.scan_loop_name() == "" and pb.replicas.scan_loop_name() == "k_scans_automala_wg" and pc.replicas.scan_loop_name() == "k_scans_automala"
    _same(a, b); _same(a, c)
    for x, y, z in zip(pa.replicas.states(), pb.replicas.states(), pc.replicas.states()):
        assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("N,d,rounds,seed,explorer", [
    (1024, 1024, 3, 1, "automala"),               # every workgroup slot of the device taken: four 256-thread workgroups per compute unit
    (64, 1024, 5, 2, "automala"),
    (33, 600, 5, 3, "automala"),                  # ragged blocks, odd N
    (16, 1024, 5, 4, "mala"),
    (7, 513, 5, 5, "automala"),
    (2, 777, 6, 6, "automala"),                   # reference + target only
    (1, 1024, 4, 7, "automala"),                  # a single chain
])
def test_fused_langevin_mw_scan_loop_equals_launch_per_scan(P, N, d, rounds, seed, explorer):
    """512 < d <= 1024 on the scaled-precision MVN path (round 6, late): k_scans_langevin_mw -- the four-waves-per-chain body as a called function,
    reading the kernel-argument segment through the implicit-argument pointer, thread 0 shaking hands, wave priorities following who waited for
    whom -- against explore + swap launches per scan, bit for bit: every recorder of every round, states, chains, streams."""
    ex = (lambda: P.AutoMALA()) if explorer == "automala" else (lambda: P.MALA(step_size=0.05))
    pa, a = _run_am(P, N, d, rounds, seed, True, "mvn", ex())
    pb, b = _run_am(P, N, d, rounds, seed, False, "mvn", ex())
    assert pa.replicas.scan_loop_name() == "" and pb.replicas.scan_loop_name() == "k_scans_langevin_mw"
    assert pa.replicas.kernel_name() == pb.replicas.kernel_name() == "k_explore_langevin_mw"
    _same(a, b)
    for x, y in zip(pa.replicas.states(), pb.replicas.states()):
        assert np.array_equal(x, y)
    calls, aborts, poisoned = pb.replicas.scan_loop_stats()
    assert calls == rounds and aborts == 0 and not poisoned


def test_fused_langevin_mw_loop_with_traces_and_split_calls(P):
    """the per-scan words the called body takes through v_readfirstlane (trace row, the scan != 1 rule) across several pte_run_scans calls per round"""
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.traces, P.online]
    def run(two):
        from pigeons_amd import _lib
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(640), n_chains=9, n_rounds=5, seed=11, explorer=P.AutoMALA(), record=rec, show_report=False),
                  debug_kernel=_lib.KERNEL_TWO_LAUNCHES if two else 0)
        e = pt.replicas
        e.run_scans(1, 1); e.run_scans(2, 3); e.run_scans(5, 2); e.run_scans(7, 1)     # scan 1 alone (no MH step), then a call that starts at scan 2
        from pigeons_amd.pt import reduce_recorders
        red = reduce_recorders(pt)
        return pt, [red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(), np.array(red.traces).copy(),
                    np.array(red.online[0]).copy(), red.explorer_acceptance_pr[0].copy(), np.array(red.am_factors[0]).copy()]
    pa, a = run(True); pb, b = run(False)
    assert pb.replicas.scan_loop_name() == "k_scans_langevin_mw" and pa.replicas.scan_loop_name() == ""
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y, equal_nan=True), k
    for x, y in zip(pa.replicas.states(), pb.replicas.states()):
        assert np.array_equal(x, y)


def test_which_engines_run_the_fused_loop(P):
    """the documented choice: SliceSampler on the MVN path with the default kernel generation, one engine, all workgroups resident"""
    from pigeons_amd import _lib
    mk = lambda **kw: P.PT(P.Inputs(**dict(dict(target=P.toy_mvn_target(8), n_chains=6, n_rounds=2, explorer=P.SliceSampler(), show_report=False,
                                                 record=[P.round_trip, P.log_sum_ratio]), **kw)))
    assert mk().replicas.scan_loop_name() == "k_scans_slice8"
    assert mk(explorer=P.ToyExplorer()).replicas.scan_loop_name() == ""
    assert mk(explorer=P.AutoMALA()).replicas.scan_loop_name() == "k_scans_automala_wg"
    assert mk(explorer=P.AutoMALA(), target=P.toy_mvn_target(600)).replicas.scan_loop_name() == "k_scans_langevin_mw"   # 512 < d <= 1024, MVN path: four waves per chain
    assert mk(explorer=P.AutoMALA(), target=P.Funnel(600), reference=P.ScaledPrecisionNormalLogPotential(1 / 9., 600)).replicas.scan_loop_name() == ""   # the funnel's loop measured no gain at d > 512
    assert mk(explorer=P.AutoMALA(), target=P.toy_mvn_target(600), n_chains=1025).replicas.scan_loop_name() == ""       # more 256-thread workgroups than the device holds at once
    assert mk(explorer=P.Compose(P.SliceSampler(), P.AutoMALA())).replicas.scan_loop_name() == ""
    assert mk(target=P.toy_mvn_target(64), n_chains=8192).replicas.scan_loop_name() == ""            # more workgroups than the GPU holds at once
    assert mk(target=P.toy_mvn_target(64), n_chains=1025).replicas.scan_loop_name() == ""            # more than one wave per SIMD: measured slower (0.91x at 2048 chains)
    assert mk(target=P.toy_mvn_target(64), n_chains=1024).replicas.scan_loop_name() == "k_scans_slice8"
    assert mk(target=P.toy_mvn_target(4096), n_chains=8).replicas.scan_loop_name() == ""             # rows beyond 16 KB: measured slower (0.99x at d = 4096)
    seq = P.PT(P.Inputs(target=P.toy_mvn_target(8), n_chains=6, n_rounds=2, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip]),
               debug_kernel=_lib.KERNEL_SLICE_SEQUENTIAL)
    assert seq.replicas.scan_loop_name() == "" and seq.replicas.kernel_name() == "k_explore_slice"
    sh = P.PT(P.Inputs(target=P.toy_mvn_target(8), n_chains=6, n_rounds=2, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip]),
              n_shards=2, transport="group")
    assert sh.replicas.scan_loop_name() == ""


def test_fused_loop_timing_hooks(P):
    """pte_timing_get(kernel = 4): one sample per pte_run_scans call; pte_scan_loop_info counts the scans inside; kinds 0 / 1 stay empty"""
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(64), n_chains=32, n_rounds=6, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip]))
    e = pt.replicas
    e.timing_reset(True)
    e.run_scans(1, 5); e.run_scans(6, 3)
    ms, n = e.timing(4)
    limit, launches, scans = e.scan_loop_info()
    assert n == 2 and launches == 2 and scans == 8 and ms > 0 and limit >= 32
    assert e.timing(0)[1] == 0 and e.timing(1)[1] == 0
    assert len(e.timing_samples(4)) == 2
    e.timing_reset(False)
