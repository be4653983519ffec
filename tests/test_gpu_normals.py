"""The bulk normal generator of k_init / k_explore_toy (pigeons.jl_amd/csrc/pte_normals.hpp: 512-output chunks, events resolved
lane-parallel, four outputs per lane, x / sd by a reciprocal multiply + two fma) held to the PLAIN procedure, bit for bit:
    state == (randn #0 .. d-1 of the replica's stream, drawn by the sequential block procedure k_test_rng runs) / sd     (IEEE division)
    rng   == the stream advanced by exactly the draws those d normals consume
for every d around the chunk / group / block boundaries, many streams (so that wedge accepts, wedge rejects, adjacent events
and tails all occur), and the ladder's own divisors.  The sequential procedure itself is pinned to the oracle in
test_gpu_parity.py::test_device_rng_matches_oracle; the swap statistic (fixed tree) to a recompute.
Reference: src/targets/toy_mvn_target.jl:10-11,15-21, src/explorers/ToyExplorer.jl:7-12."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _streams(seed, n):
    master = O.OracleRng(seed)
    return [master.split().state for _ in range(n)]


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 300, 511, 512, 513, 575, 576, 577, 767, 768, 1000, 1024, 1025, 2047, 4000, 4096])
def test_init_states_equal_sequential_draws_divided(P, d):
    from pigeons_amd.engine import Engine, test_rng_fill
    from pigeons_amd import _lib
    N, seed = 24, 7 + d
    e = Engine(n_chains=N, dim=d, seed=seed, explorer=_lib.EXPLORER_TOY)
    x, chain, rng = e.states()
    sd = np.sqrt(10.0)                                  # initialization(::ScaledPrecisionNormalPath): randn / sqrt(precision1)
    for i, st in enumerate(_streams(seed, N)):
        z, st1 = test_rng_fill(st, 1, d)
        assert np.array_equal(x[i], z / sd), (i, np.flatnonzero(x[i] != z / sd)[:5])
        assert tuple(int(v) for v in rng[i]) == st1, i


@pytest.mark.parametrize("d,N", [(1, 5), (70, 9), (256, 40), (513, 33), (1024, 200), (4096, 64), (1500, 300)])
def test_toy_explore_equals_sequential_draws_divided(P, d, N):
    from pigeons_amd.engine import Engine, test_rng_fill, test_sqr_norm
    from pigeons_amd import _lib
    e = Engine(n_chains=N, dim=d, seed=3, explorer=_lib.EXPLORER_TOY)
    betas = e.schedule()
    sd = np.sqrt((1.0 - betas) * 1.0 + betas * 10.0)     # upload_ladder: sqrt(precision(beta)), ScaledPrecisionNormalPath.jl:45-48
    for scan in (1, 2, 3):
        _, chain0, rng0 = e.states()
        e.explore(scan)
        x, chain, rng = e.states()
        assert np.array_equal(chain, chain0)
        for i in range(N):
            z, st1 = test_rng_fill(tuple(int(v) for v in rng0[i]), 1, d)
            want = z / sd[chain[i]]
            assert np.array_equal(x[i], want), (scan, i, np.flatnonzero(x[i] != want)[:5])
            assert not np.any(np.signbit(x[i]) & (x[i] == 0.0))
            assert tuple(int(v) for v in rng[i]) == st1, (scan, i)
        e.swap(scan)                                     # (moves chain labels: the next scan divides by another chain's sd)
    # the swap statistic the kernel leaves (fixed tree, two levels in the lane + four DPP steps per 256 outputs) == a full recompute:
    # a fresh engine given these states takes the same swap decisions
    x, chain, rng = e.states()
    f = Engine(n_chains=N, dim=d, seed=3, explorer=_lib.EXPLORER_TOY)
    f.set_states(x=x, chain=chain, rng=rng)
    e.swap(4); f.swap(4)
    assert np.array_equal(e.states()[1], f.states()[1]) and np.array_equal(e.states()[2], f.states()[2])


def test_divisor_with_an_all_ones_significand_takes_the_ieee_division(P):
    """The reciprocal-multiply division is proved for every divisor except a significand of all ones: that one branches to x / sd."""
    from pigeons_amd.engine import Engine, test_rng_fill
    from pigeons_amd import _lib
    N, d = 4, 700
    e = Engine(n_chains=N, dim=d, seed=11, explorer=_lib.EXPLORER_TOY, target_params=[1.0, 4.0])
    # precision (1 - b) * 1 + b * 4 = 1 + 3 b; sd = sqrt(precision).  Choose b so that sd's significand is all ones: sd = 2 - 2^-52
    sd_target = np.nextafter(2.0, 0.0)
    b = (sd_target * sd_target - 1.0) / 3.0
    betas = np.array([0.0, b * 0.5, b, 1.0])
    e.set_schedule(betas)
    prec = (1.0 - betas) * 1.0 + betas * 4.0
    sd = np.sqrt(prec)
    _, chain0, rng0 = e.states()
    e.explore(1)
    x, chain, rng = e.states()
    for i in range(N):
        z, st1 = test_rng_fill(tuple(int(v) for v in rng0[i]), 1, d)
        assert np.array_equal(x[i], z / sd[chain[i]]), i
    if not any((np.float64(s).view(np.uint64) & np.uint64(0xfffffffffffff)) == np.uint64(0xfffffffffffff) for s in sd):
        pytest.skip("could not construct a divisor with an all-ones significand through the schedule (sd = %r)" % (sd,))


@pytest.mark.parametrize("d,N", [(3, 6), (256, 12), (512, 40), (577, 33), (1024, 64), (1500, 50), (4096, 24)])
def test_cut_short_and_fallback_paths_equal_sequential_draws(P, d, N):
    """libpte_nrmcut.so is the product source with -DNRM_MAX_EV_=2: the generator lists TWO events per chunk where ~9 occur, so nearly
    every chunk is cut short at its second event (pos_limit), most groups of 256 outputs stop at the `break` (the next chunk starts at the
    first output not emitted) and a chunk whose second event lies inside its first group produces that group with the block-by-block
    fallback.  A live tail that leaves the chunk cuts the chunk through the same pos_limit.  The product build reaches these paths with
    probability ~0 (more than 32 events in 576 positions); here they carry most of the outputs, and every state and stream position
    must still be the sequential procedure's."""
    import os
    from pigeons_amd.engine import Engine, test_rng_fill
    from pigeons_amd import _lib
    if not os.path.exists(_lib.NRMCUT_LIB_PATH):
        pytest.fail("libpte_nrmcut.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    e = Engine(test_build=_lib.NRMCUT_LIB_PATH, n_chains=N, dim=d, seed=5 + d, explorer=_lib.EXPLORER_TOY)
    p = Engine(n_chains=N, dim=d, seed=5 + d, explorer=_lib.EXPLORER_TOY)          # the product build: same states, same streams
    x, chain, rng = e.states()
    xp, chainp, rngp = p.states()
    assert np.array_equal(x, xp) and np.array_equal(rng, rngp) and np.array_equal(chain, chainp)
    sd0 = np.sqrt(10.0)
    for i, st in enumerate(_streams(5 + d, N)):
        z, st1 = test_rng_fill(st, 1, d)
        assert np.array_equal(x[i], z / sd0), (i, np.flatnonzero(x[i] != z / sd0)[:5])
        assert tuple(int(v) for v in rng[i]) == st1, i
    betas = e.schedule()
    sd = np.sqrt((1.0 - betas) * 1.0 + betas * 10.0)
    for scan in (1, 2):
        _, chain0, rng0 = e.states()
        e.explore(scan); p.explore(scan)
        x, chain, rng = e.states()
        xp, chainp, rngp = p.states()
        assert np.array_equal(x, xp) and np.array_equal(rng, rngp)
        for i in range(N):
            z, st1 = test_rng_fill(tuple(int(v) for v in rng0[i]), 1, d)
            assert np.array_equal(x[i], z / sd[chain[i]]), (scan, i)
            assert tuple(int(v) for v in rng[i]) == st1, (scan, i)
        e.swap(scan); p.swap(scan)
        assert np.array_equal(e.states()[1], p.states()[1])      # same swap decisions: the fallback leaves the same block sums
