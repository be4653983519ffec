"""Forward progress of the one-kernel scan loop (round 6; VERDICT r05 "next round" item 1).

The reference's `while next_scan!(pt)` loop (src/pt/pigeons.jl:46-55) cannot hang and cannot leave replicas at different scans.  The
one-kernel form of pte_run_scans spins on pairwise hand-shakes, so it must (a) never start unless every workgroup is on the device
(the residency gate of pte_kernels.hpp: enforced inside the launch, fallback to explore + swap launches with identical results),
(b) end at once when one wave gives up (the waits poll the engine's error word), and (c) refuse to go on with replicas that stopped at
different scans (the engine is poisoned until pte_set_state).  This module tries to break each of the three:

  * two engines driven from two host threads at once, and from two fresh processes on device 0: both finish, neither times out, both
    equal their serial runs bit for bit -- whether each call ran as one launch or fell back (pte_scan_loop_stats says which);
  * the test build's PTE_KERNEL_TEST_LATE_WORKGROUP: one workgroup reaches the gate 80 ms late, the launch aborts with nothing written
    and the call falls back, results equal to the undisturbed run's, back-off 1, 2, 4 calls;
  * the test build's PTE_KERNEL_TEST_DEAD_CHAIN: the wave of chain 7 dies, the call returns within 10 s (one 3 s time-out, not one per
    blocked wave) with the hand-shake error, every further call is refused, pte_set_state revives the engine and it then runs correctly.
"""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _make(P, N, d, seed, flags=0, explorer=None, rounds=9, target=None):
    return P.PT(P.Inputs(target=target or P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, seed=seed, explorer=explorer or P.SliceSampler(),
                         record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), debug_kernel=flags)


def _rounds(P, pt, rounds):
    out = []
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        out.append([red.index_process.copy(), np.array(red.round_trip), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(),
                    red.log_sum_ratio[2].copy(), red.explorer_n_steps[0].copy(), np.array(pt.shared.tempering.schedule.grids).copy()])
    out.append(list(pt.replicas.states()))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for r, (ra, rb) in enumerate(zip(a, b)):
        assert len(ra) == len(rb)
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y, equal_nan=True), (r, k)


def test_two_engines_from_two_host_threads(P):
    """N = 1024, d = 64 each, rounds 1-9 (1022 scans) run concurrently from two threads on one device: whatever the two launches do to each
    other -- share the SIMDs, or one finds the other in its way and falls back -- both runs equal their serial runs and nobody times out"""
    serial = []
    for seed in (1, 2):
        pt = _make(P, 1024, 64, seed)
        assert pt.replicas.scan_loop_name() == "k_scans_slice8"
        serial.append(_rounds(P, pt, 9)); del pt
    pts = [_make(P, 1024, 64, seed) for seed in (1, 2)]
    res, errs = [None, None], []

    def work(i):
        try:
            res[i] = _rounds(P, pts[i], 9)
        except Exception as exc:          # noqa: BLE001
            errs.append((i, repr(exc)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join(120)
    assert not any(t.is_alive() for t in th), "a scan loop hangs"
    assert not errs, errs
    assert time.time() - t0 < 60
    for i in range(2):
        _same(serial[i], res[i])
        fused, aborts, poisoned = pts[i].replicas.scan_loop_stats()
        assert not poisoned and fused + aborts >= 1, (fused, aborts)       # each of the 9 calls either ran as one launch or fell back after an abort


def test_two_four_wave_scan_loops_from_two_host_threads(P):
    """k_scans_langevin_mw takes EVERY workgroup slot of the device (1024 chains x 256 threads, four per compute unit): two such engines run concurrently
    from two threads cannot both be resident -- one launch's gate finds workgroups missing, aborts with nothing written, and that call runs as explore + swap
    launches.  Both runs equal their serial runs, nobody times out, nobody is poisoned."""
    mk = lambda seed: _make(P, 1024, 600, seed, explorer=P.AutoMALA(), rounds=5)
    serial = []
    for seed in (1, 2):
        pt = mk(seed)
        assert pt.replicas.scan_loop_name() == "k_scans_langevin_mw"
        serial.append(_rounds(P, pt, 5)); del pt
    pts = [mk(seed) for seed in (1, 2)]
    res, errs = [None, None], []

    def work(i):
        try:
            res[i] = _rounds(P, pts[i], 5)
        except Exception as exc:          # noqa: BLE001
            errs.append((i, repr(exc)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join(120)
    assert not any(t.is_alive() for t in th), "a scan loop hangs"
    assert not errs, errs
    assert time.time() - t0 < 60
    for i in range(2):
        _same(serial[i], res[i])
        fused, aborts, poisoned = pts[i].replicas.scan_loop_stats()
        assert not poisoned and fused + aborts >= 1, (fused, aborts)


_CHILD = r"""
import sys, hashlib, numpy as np
sys.path[:0] = [%r, %r, %r]
import pigeons_amd as P
seed = int(sys.argv[1])
pt = P.PT(P.Inputs(target=P.toy_mvn_target(64), n_chains=1024, n_rounds=9, seed=seed, explorer=P.SliceSampler(),
                   record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False))
h = hashlib.sha256()
for _ in range(9):
    assert P.next_round(pt)
    red = P.run_one_round(pt); P.adapt(pt, red)
    for a in (red.index_process, red.swap_acceptance_pr[0], red.log_sum_ratio[0], np.array(pt.shared.tempering.schedule.grids)):
        h.update(np.ascontiguousarray(a).tobytes())
for a in pt.replicas.states():
    h.update(np.ascontiguousarray(a).tobytes())
f, a, p = pt.replicas.scan_loop_stats()
print("RESULT", h.hexdigest(), f, a, int(p))
""" % (ROOT, os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tests"))


def _child(seed):
    return subprocess.Popen([sys.executable, "-c", _CHILD, str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _result(p, timeout):
    out, err = p.communicate(timeout=timeout)
    assert p.returncode == 0, err[-2000:]
    line = [l for l in out.splitlines() if l.startswith("RESULT")][-1].split()
    return line[1], int(line[2]), int(line[3]), int(line[4])


def test_two_engines_from_two_processes():
    """the same from two fresh PROCESSES on device 0 (two HSA queues the runtime knows nothing about each other): identical digests of
    every recorder and state to the two serial runs, no time-out, no poisoning"""
    serial = [_result(_child(seed), 300) for seed in (3, 4)]
    t0 = time.time()
    ps = [_child(seed) for seed in (3, 4)]
    both = [_result(p, 300) for p in ps]
    assert time.time() - t0 < 240
    for s, b in zip(serial, both):
        assert s[0] == b[0]
        assert b[3] == 0 and b[1] + b[2] >= 1
    assert serial[0][0] != serial[1][0]


@pytest.mark.parametrize("explorer,N,d,kernel", [("slice", 64, 96, "k_scans_slice8"), ("automala", 37, 24, "k_scans_automala_wg"), ("automala", 12, 600, "k_scans_langevin_mw")])
def test_late_workgroup_aborts_the_launch_and_the_call_falls_back(P, explorer, N, d, kernel):
    """PTE_KERNEL_TEST_LATE_WORKGROUP (test build): workgroup 3 arrives 80 ms late at every launch's gate -> the gate's 50 ms bound passes,
    every workgroup leaves with NOTHING written, pte_run_scans runs explore + swap launches instead.  Same results as the undisturbed
    engine; aborts are counted and backed off (calls 1, 3, 6 attempt the launch: 1, then 2 skipped calls ... in six rounds = three aborts)"""
    from pigeons_amd import _lib
    ex = (lambda: P.SliceSampler()) if explorer == "slice" else (lambda: P.AutoMALA())
    ref = _make(P, N, d, 5, explorer=ex())
    assert ref.replicas.scan_loop_name() == kernel
    a = _rounds(P, ref, 6)
    fused, aborts, poisoned = ref.replicas.scan_loop_stats()
    assert (fused, aborts, poisoned) == (6, 0, False)
    pt = _make(P, N, d, 5, flags=_lib.KERNEL_TEST_LATE_WORKGROUP, explorer=ex())
    assert pt.replicas.scan_loop_name() == kernel
    t0 = time.time()
    b = _rounds(P, pt, 6)
    assert time.time() - t0 < 20
    _same(a, b)
    fused, aborts, poisoned = pt.replicas.scan_loop_stats()
    assert fused == 0 and aborts == 3 and not poisoned, (fused, aborts)          # call 1 aborts (skip 1), call 3 aborts (skip 2), call 6 aborts


def test_product_library_refuses_the_fault_injection_flags(P):
    from pigeons_amd import _lib
    from pigeons_amd.engine import Engine
    with pytest.raises(P.PteError, match="fault-injection"):
        Engine(n_chains=16, dim=8, explorer=_lib.EXPLORER_SLICE, debug_kernel=_lib.KERNEL_TEST_DEAD_CHAIN)


@pytest.mark.parametrize("explorer,N,d", [("slice", 64, 96), ("automala", 40, 24), ("automala", 12, 600)])      # (d = 600: k_scans_langevin_mw, 256 threads per chain)
def test_dead_chain_times_out_once_poisons_the_engine_and_set_state_revives_it(P, explorer, N, d):
    from pigeons_amd import _lib
    ex = (lambda: P.SliceSampler()) if explorer == "slice" else (lambda: P.AutoMALA())
    # what the engine should do from the snapshot on, from an undisturbed twin
    good = _make(P, N, d, 9, explorer=ex())
    assert P.next_round(good); red = P.run_one_round(good); P.adapt(good, red)           # round 1: 2 scans
    snap = [np.array(a).copy() for a in good.replicas.states()]
    grids = np.array(good.shared.tempering.schedule.grids).copy()
    good.replicas.run_scans(1, 4); good.replicas.reduce()
    want = [good.replicas.index_process().copy()] + [np.array(a).copy() for a in good.replicas.states()]

    pt = _make(P, N, d, 9, flags=_lib.KERNEL_TEST_DEAD_CHAIN, explorer=ex())
    eng = pt.replicas
    assert P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)                 # 2 scans: the fault needs a third swap
    assert eng.scan_loop_stats() == (1, 0, False)
    for x, y in zip(snap, eng.states()):
        assert np.array_equal(x, y)
    t0 = time.time()
    with pytest.raises(P.PteError, match="gave up waiting for its swap partner") as ei:
        eng.run_scans(1, 4)
    dt = time.time() - t0
    assert 2.5 < dt < 10.0, dt                   # ONE 3 s time-out: the other waves saw the error word and left
    assert "only pte_set_state" in str(ei.value)
    assert eng.scan_loop_stats()[2] is True
    for call in (lambda: eng.run_scans(1, 1), lambda: eng.explore(1), lambda: eng.swap(1), lambda: eng.reduce(), lambda: eng.states(),
                 lambda: eng.set_schedule(grids)):
        with pytest.raises(P.PteError, match="poisoned"):
            call()
    with pytest.raises(P.PteError, match="needs state, chain and rng"):
        eng.set_states(x=snap[0])
    eng.set_states(x=snap[0], chain=snap[1], rng=snap[2])                                  # revive: every field of every replica, recorders discarded
    assert eng.scan_loop_stats()[2] is False
    for x, y in zip(snap, eng.states()):
        assert np.array_equal(x, y)
    # the engine works again -- through the launch-per-scan entry points (the fault would strike the one-kernel loop again) ...
    for s in range(1, 5):
        eng.explore(s); eng.swap(s)
    eng.reduce()
    got = [eng.index_process().copy()] + [np.array(a).copy() for a in eng.states()]
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    # ... and the one-kernel loop itself runs again where the fault does not reach (two scans per call): flags re-based, no stale epoch
    eng.run_scans(1, 2); eng.run_scans(3, 2); eng.reduce()
    good.replicas.run_scans(1, 2); good.replicas.run_scans(3, 2); good.replicas.reduce()
    assert eng.scan_loop_stats()[0] == 4                        # round 1, the failed call (it did run as one launch), and these two
    assert np.array_equal(eng.index_process(), good.replicas.index_process())
    for x, y in zip(good.replicas.states(), eng.states()):
        assert np.array_equal(x, y)


def test_step_size_search_log_refuses_the_scan_that_would_overrun_it(P):
    """with PTE_RECORD_REFERENCE_REDUCTION the AutoMALA kernels write row `scans_in_round` of the search log; the launch-per-scan
    path checked the capacity at the SWAP, after the explore kernel had written past the end.  Now pte_explore refuses first."""
    from pigeons_amd import _lib
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(8), n_chains=6, n_rounds=2, seed=1, explorer=P.AutoMALA(),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False),
              debug_kernel=_lib.KERNEL_TWO_LAUNCHES, reference_reduction=True)
    eng = pt.replicas
    cap = int(eng.cfg.max_scans_per_round)
    for s in range(1, cap + 1):
        eng.explore(s); eng.swap(s)
    with pytest.raises(P.PteError, match="step-size-search log full"):
        eng.explore(cap + 1)
    eng.reduce()
    eng.explore(1); eng.swap(1)
