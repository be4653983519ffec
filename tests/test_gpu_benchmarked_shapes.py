"""The shapes bench.py and profiles/ quote, checked against the ORACLE (not against properties, not against another kernel of
this engine) -- VERDICT r02 "next round" item 1:

  * the metric configuration itself, toy_mvn_target(1024), N = 1024, SliceSampler, rounds 1-2 (6 scans x 1024 replicas)
  * BASELINE configs[2] at full size: funnel d = 128, N = 1024, AutoMALA, 3 rounds
  * k_explore_slice8_lds10k -- the kernel every run with more than 2048 chains per GPU launches (the strong-scaling anchor) --
    at N = 3000, d = 70 and N = 4096, d = 256, with kernel_name() asserted
  * the same kernel at the tree depths it is QUOTED at: N = 2304, d = 1024 against the oracle; N = 2100 / 8192 at
    d = 4096, N = 2100 at d = 1500, N = 2500 at d = 1024 bit for bit against the sequential kernel + the run's properties
  * ToyExplorer at N = 8192, d = 4096 (the shape the HBM-bound kernels are profiled at), 2 rounds

Reference procedure being restated by the oracle: src/explorers/SliceSampler.jl:24-237, src/explorers/AutoMALA.jl:106-182,
src/targets/toy_mvn_target.jl:15-21.  Integers exact, floats 1e-9 (funnel: 1e-6, ocml vs glibc exp/log)."""
import os

import numpy as np
import pytest

import oracle as O
from test_gpu_parity import _check_round, _check_am_round, _mk_am

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _threads():
    return max(1, len(os.sched_getaffinity(0)))


def _mk_slice(P, N, d, rounds, seed):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=P.SliceSampler(), seed=seed,
                       record=rec, show_report=False))
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, record_online=1, explorer=O.EXPLORER_SLICE, n_threads=_threads())
    return pt, ref


def test_metric_config_against_the_oracle(P):
    """bench.py's workload, rounds 1-2: every replica's states, RNG counters, chains, the index process, recorders and the
    adapted schedule against the oracle's full O(d) recompute (about 25 s of one host core)."""
    pt, ref = _mk_slice(P, 1024, 1024, 2, seed=1)
    assert pt.replicas.kernel_name() == "k_explore_slice8"
    for _ in range(2):
        _check_round(P, pt, ref)


@pytest.mark.parametrize("N,d,rounds,seed", [(3000, 70, 3, 11), (4096, 256, 2, 5)])
def test_many_replica_slice_kernel_against_the_oracle(P, N, d, rounds, seed):
    """k_explore_slice8_lds10k (256-draw window, 10 KB of LDS per replica) against the oracle."""
    pt, ref = _mk_slice(P, N, d, rounds, seed)
    assert pt.replicas.kernel_name() == "k_explore_slice8_lds10k"
    for _ in range(rounds):
        _check_round(P, pt, ref)


def test_strong_scaling_anchor_kernel_against_the_oracle(P):
    """k_explore_slice8_lds10k<4, 9> -- the tree depth of d = 1024 with more than 2048 replicas on the GPU, one of
    the instantiations that spill VGPRs to scratch (profiles/r04_kernel_resources.txt) -- rounds 1-2 against the ORACLE's full O(d)
    recompute (SliceSampler.jl:89-237): states, RNG counters, chains, index process, recorders, schedule (about a minute of one host core)."""
    pt, ref = _mk_slice(P, 2304, 1024, 2, seed=3)
    assert pt.replicas.kernel_name() == "k_explore_slice8_lds10k"
    for _ in range(2):
        _check_round(P, pt, ref)


def _run_slice_rounds(P, N, d, rounds, seed, impl, kernel="k_explore_slice8_lds10k"):
    from pigeons_amd import _lib
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, seed=seed, explorer=P.SliceSampler(),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), debug_kernel=impl)
    assert pt.replicas.kernel_name() == ("k_explore_slice" if (impl & ~_lib.KERNEL_FLAG_BITS) == 1 else kernel)
    out = []
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        out.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.explorer_n_steps[0].copy(), red.explorer_acceptance_pr[0].copy(),
                    red.log_sum_ratio[0].copy(), np.array(pt.shared.tempering.schedule.grids).copy()))
    return pt, out


@pytest.mark.parametrize("N,d,rounds,seed", [
    (2100, 4096, 2, 7),       # <6, 9>: the deepest tree, just past the switch to the 10 KB kernel
    (8192, 4096, 2, 1),       # <6, 9> at BASELINE configs[3] on ONE GPU: the shape behind "C4 on one GPU" in the bench line and DESIGN's strong-scaling anchor
    (2100, 1500, 2, 4),       # <5, 9>: 24 blocks, a ragged 256-coordinate block and a tree padded to 32
    (2500, 1024, 3, 9),       # <4, 9> once more, three rounds
])
def test_strong_scaling_anchor_kernel_equals_sequential_kernel(P, N, d, rounds, seed):
    """the instantiations of k_explore_slice8_lds10k for d = 1024 ... 4096 -- the ones with scratch spills, the ones
    every N >= 2304 cell of DESIGN's chains-per-GPU table and the 1-GPU strong-scaling anchor run -- bit for bit against the plain sequential
    kernel (debug_kernel = 1, which the oracle pins at every size it can afford, incl. d = 4096 and the metric shape), then the
    size-independent properties of the run (tests/test_gpu_configs.py)."""
    from test_gpu_configs import _mvn_properties
    pa, a = _run_slice_rounds(P, N, d, rounds, seed, 1)
    sa = pa.replicas.states(); del pa
    pb, b = _run_slice_rounds(P, N, d, rounds, seed, 0)
    sb = pb.replicas.states(); del pb
    for r, (ra, rb) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y), (r, k)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    del a, b, sa, sb
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=2, seed=seed + 100, explorer=P.SliceSampler(),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False))
    assert pt.replicas.kernel_name() == "k_explore_slice8_lds10k"
    _mvn_properties(P, pt, N, d, 2)


def test_config2_full_size_against_the_oracle(P):
    """BASELINE configs[1] AS QUOTED in the bench line -- toy_mvn_target(1024), n_chains = 256, SliceSampler, the
    one-kernel scan loop -- rounds 1-2 against the oracle's full O(d) recompute (SliceSampler.jl:89-237): states, RNG counters, chains, index
    process, recorders, adapted schedule (about 6 s of the host's cores)."""
    pt, ref = _mk_slice(P, 256, 1024, 2, seed=2)
    assert pt.replicas.kernel_name() == "k_explore_slice8" and pt.replicas.scan_loop_name() == "k_scans_slice8"
    for _ in range(2):
        _check_round(P, pt, ref)


@pytest.mark.parametrize("two_launches", [False, True])
def test_config4_shard_shape_equals_sequential_kernel(P, two_launches):
    """the C4 shard AS QUOTED -- toy_mvn_target(4096), 1024 chains on one GPU, k_explore_slice8<6, 9> (the 512-draw
    kernel at the deepest tree, not the anchor's _lds10k twin) -- two rounds bit for bit against the plain sequential kernel (debug_kernel = 1,
    oracle-pinned at d = 4096 in tests/test_gpu_parity.py), under both settings of the scan loop (rows of 32 KB stay on two launches per scan
    either way: pte_scan_loop_name is "" -- asserted, so that the day the fused loop takes d = 4096 this test holds IT to the sequential kernel)."""
    from pigeons_amd import _lib
    flag = _lib.KERNEL_TWO_LAUNCHES if two_launches else 0
    pa, a = _run_slice_rounds(P, 1024, 4096, 2, 13, 1 | flag)
    sa = pa.replicas.states(); del pa
    pb, b = _run_slice_rounds(P, 1024, 4096, 2, 13, flag, kernel="k_explore_slice8")
    assert pb.replicas.scan_loop_name() == ""
    sb = pb.replicas.states(); del pb
    for r, (ra, rb) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y), (r, k)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_config5_shard_shape_equals_byte_lattice_kernel(P):
    """the C5 shard AS QUOTED -- Ising 256 x 256, 512 chains, IsingMetropolis(3 sweeps), k_explore_ising_spec (56
    lane hypotheses per 16-site chunk) -- two rounds bit for bit against the scalar byte-lattice sweep (PTE_KERNEL_ISING_BYTES, the raster
    loop of examples/ising.jl:96-116 site by site; oracle-pinned at 256 x 256 in tests/test_gpu_parity.py): index process, recorders, adapted
    schedule, every spin, every RNG counter."""
    from pigeons_amd import _lib
    outs = []
    for impl, name in ((_lib.KERNEL_ISING_BYTES, "k_explore_ising"), (0, "k_explore_ising_spec")):
        pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, 256), n_chains=512, n_rounds=2, seed=3,
                           record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), debug_kernel=impl)
        assert pt.replicas.kernel_name() == name
        rows = []
        for _ in range(2):
            assert P.next_round(pt)
            red = P.run_one_round(pt); P.adapt(pt, red)
            rows.append((red.index_process.copy(), np.array(red.round_trip), red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(),
                         red.log_sum_ratio[2].copy(), red.explorer_acceptance_pr[0].copy(), red.explorer_acceptance_pr[1].copy(),
                         np.array(pt.shared.tempering.schedule.grids).copy()))
        outs.append((rows, pt.replicas.states())); del pt
    (a, sa), (b, sb) = outs
    for r, (ra, rb) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert np.array_equal(x, y), (r, k)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    assert 0.0 < sa[0].mean() < 1.0                        # (spins did move)


def test_config3_full_size_against_the_oracle(P):
    """BASELINE configs[2] as quoted: funnel d = 128, n_chains = 1024, AutoMALA, rounds 1-3."""
    pt, ref = _mk_am(P, 1024, 128, 3, "funnel", seed=1)
    assert pt.replicas.kernel_name().startswith("k_explore_automala")
    # explorer_acceptance_pr is a mean over a handful of MH steps of exp(difference of two log densities); in the funnel's neck
    # those log densities reach 1e6 and beyond, so one ulp of ocml-vs-glibc exp / log shows up at a few 1e-6 relative in a chain
    # or two out of 1024 (measured: 2 chains, 3.2e-6); likewise 2 of the 131,072 state coordinates, of magnitude 4e-3 in replicas
    # whose other coordinates are O(1), differ by 1.1e-7 absolute after 14 scans of leapfrog steps (states are not a recorder: the
    # test asks 1e-6 of the replica's scale there).  Every integer, the swap recorders and the schedule stay at 1e-6 relative.
    # The cause is demonstrated, not only stated, by test_config3_residual_is_one_ulp_of_exp_and_log below.
    for _ in range(3):
        _check_am_round(P, pt, ref, rtol=1e-6, acc_rtol=1e-5, state_atol=1e-6)


def test_config3_residual_is_one_ulp_of_exp_and_log(P):
    """Why BASELINE configs[2] at full size needs acc_rtol = 1e-5 / state_atol = 1e-6 in round 3 (the test above), shown instead of asserted:
    the oracle's OWN exp / log nudged by one ulp (po_set_libm_nudge: up, down, alternating -- glibc and ocml both err by less than that,
    differently) moves exactly these outputs by exactly this much and nothing else.  Per round, four oracle runs and the device:
      * every integer (index process, leapfrog counts, RNG counters, chains) is the same in all five;
      * rounds 1-2: the device agrees with the plain oracle to 1e-9 everywhere;
      * round 3: wherever the device leaves the 1e-9 band -- a few chains' mean MH acceptance, a few hundred state coordinates in the
        funnel's neck -- its residual is no larger than the spread the nudged oracles show among themselves (x2 for the head-room of three samples)."""
    N, d, rounds = 1024, 128, 3
    runs = {}
    try:
        for mode in (0, 1, 2, 3):
            O.set_libm_nudge(mode)
            ref = O.OraclePT(n_chains=N, dim=d, seed=1, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, am_preconditioner=2,
                             n_threads=_threads())
            out = []
            for _ in range(rounds):
                ref.run_round()
                am, an, ss, sn = ref.explorer_stats()
                x, chain, rng = ref.states()
                out.append(dict(index=ref.index_process().copy(), steps=np.array(ss).copy(), steps_n=np.array(sn).copy(), acc=np.array(am).copy(),
                                acc_n=np.array(an).copy(), swap=np.array(ref.swap_pr()[0]).copy(), sched=np.array(ref.schedule()).copy(),
                                x=x.copy(), chain=chain.copy(), rng=rng.copy()))
            runs[mode] = out
    finally:
        O.set_libm_nudge(0)
    pt, _ = _mk_am(P, N, d, rounds, "funnel", seed=1)
    for r in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        x, chain, rng = pt.replicas.states()
        base = runs[0][r]
        for mode in (1, 2, 3):                                      # the nudge moves no integer
            o = runs[mode][r]
            assert all(np.array_equal(o[k], base[k]) for k in ("index", "steps", "steps_n", "acc_n", "chain", "rng")), (r, mode)
        assert np.array_equal(red.index_process, base["index"]) and np.array_equal(chain, base["chain"]) and np.array_equal(rng, base["rng"])
        assert np.array_equal(red.explorer_n_steps[0], base["steps"]) and np.array_equal(red.explorer_acceptance_pr[1], base["acc_n"])
        rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        band_acc = max(rel(runs[m][r]["acc"], base["acc"]).max() for m in (1, 2, 3))
        band_x = max(np.abs(runs[m][r]["x"] - base["x"]).max() for m in (1, 2, 3))
        dev_acc = rel(red.explorer_acceptance_pr[0], base["acc"]).max()
        dev_x = np.abs(x - base["x"]).max()
        np.testing.assert_allclose(red.swap_acceptance_pr[0], base["swap"], rtol=1e-6, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, base["sched"], rtol=1e-6)
        if r < 2:
            assert dev_acc < 1e-9 and dev_x < 1e-9, (r, dev_acc, dev_x)
        else:
            assert band_acc > 1e-7 and band_x > 1e-8, (band_acc, band_x)        # the sensitivity is real: one ulp opens a band five to six orders above it
            assert dev_acc <= 2.0 * band_acc and dev_x <= 2.0 * band_x, (dev_acc, band_acc, dev_x, band_x)


def test_toy_explorer_hbm_shape_against_the_oracle(P):
    """ToyExplorer at N = 8192, d = 4096 (256 MiB of state; the shape k_explore_toy / k_init are profiled at), 2 rounds."""
    N, d = 8192, 4096
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=2, explorer=P.ToyExplorer(), record=rec, show_report=False))
    assert pt.replicas.kernel_name() == "k_explore_toy"
    ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_TOY, n_threads=_threads())
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()          # k_init at this shape
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-14, atol=0)
    del x, xr
    for _ in range(2):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process()) and red.round_trip == ref.round_trip()
        m, n = red.swap_acceptance_pr; mr, nr = ref.swap_pr()
        assert np.array_equal(n, nr)
        np.testing.assert_allclose(m, mr, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-9)
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    assert np.mean(x == xr) > 0.985                                           # fast-path draws are bit-identical
    np.testing.assert_allclose(x, xr, rtol=1e-12, atol=0)
