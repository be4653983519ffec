"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same seed.

Bar (BASELINE.json north_star): integer outputs (swap-index sequence, chains, RNG counters,
round trips, step counts) bit-exact; floating-point recorders / schedule within 1e-6 relative
(asserted much tighter here: 1e-9).  States are compared at 1e-12 relative: they are bit-identical
except where a ziggurat slow path went through libm vs ocml log/exp (<= 1 ulp apart).
"""
import math
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


def _mk(P, N, d, rounds, explorer, seed=1, record=None, debug_kernel=0):
    exp = {"toy": P.ToyExplorer(), "slice": P.SliceSampler()}[explorer]
    record = record or [P.round_trip, P.index_process, P.log_sum_ratio, P.online]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp, seed=seed,
                       record=record, show_report=False), debug_kernel=debug_kernel)
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, record_online=1,
                     explorer={"toy": O.EXPLORER_TOY, "slice": O.EXPLORER_SLICE}[explorer])
    return pt, ref


def _check_round(P, pt, ref, check_states=True):
    assert P.next_round(pt)
    red = P.run_one_round(pt)
    P.adapt(pt, red)
    ref.run_round()
    # integers: exact
    assert np.array_equal(red.index_process, ref.index_process())
    assert red.round_trip == ref.round_trip()
    m, n = red.swap_acceptance_pr
    mr, nr = ref.swap_pr()
    assert np.array_equal(n, nr)
    np.testing.assert_allclose(m, mr, rtol=RTOL, atol=1e-300)      # atol: subnormal exp() results
    up, un, dn, dnn = red.log_sum_ratio
    upr, unr, dnr, dnnr = ref.log_sum_ratio()
    assert np.array_equal(un, unr) and np.array_equal(dnn, dnnr)
    np.testing.assert_allclose(up, upr, rtol=RTOL)
    np.testing.assert_allclose(dn, dnr, rtol=RTOL)
    am, an = red.explorer_acceptance_pr
    ss, sn = red.explorer_n_steps
    amr, anr, ssr, snr = ref.explorer_stats()
    assert np.array_equal(an, anr) and np.array_equal(sn, snr)
    assert np.array_equal(ss, ssr)                      # integer-valued sums
    np.testing.assert_allclose(am, amr, rtol=RTOL)
    # adapted schedule + stepping stone + barrier
    np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=RTOL)
    if pt.inputs.n_chains > 1:
        np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=RTOL)
        np.testing.assert_allclose(P.global_barrier(pt), ref.global_barrier(), rtol=RTOL)
    om, ov, on = red.online
    omr, ovr, onr = ref.online()
    assert on == onr
    np.testing.assert_allclose(om, omr, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(ov, ovr, rtol=1e-7, atol=1e-12)
    if check_states:
        x, chain, rng = pt.replicas.states()
        xr, cr, rr = ref.states()
        assert np.array_equal(chain, cr)
        assert np.array_equal(rng, rr)                  # every replica consumed exactly the same draws
        np.testing.assert_allclose(x, xr, rtol=1e-12, atol=0)


@pytest.mark.parametrize("kind,n", [(0, 1000), (1, 20000), (2, 20000)])
@pytest.mark.parametrize("seed", [1, 2, 12345])
def test_device_rng_matches_oracle(P, kind, n, seed):
    from pigeons_amd.engine import test_rng_fill
    master = O.OracleRng(seed)
    r = master.split()
    out, st = test_rng_fill(r.state, kind, n)
    f = [r.rand, r.randn, r.randexp][kind]
    ref = np.array([f() for _ in range(n)])
    assert st == r.state                              # same number of raw draws consumed
    exact = np.mean(out == ref)
    assert exact > 0.985, exact                        # fast paths are bit-identical
    np.testing.assert_allclose(out, ref, rtol=1e-14, atol=0)   # slow paths: libm vs ocml, <= 1 ulp


@pytest.mark.parametrize("d", [1, 2, 3, 10, 63, 64, 65, 100, 128, 1000, 1024, 1025, 4096])
def test_sqr_norm_tree_matches_oracle(P, d):
    from pigeons_amd.engine import test_sqr_norm
    rng = np.random.default_rng(d)
    x = rng.standard_normal((7, d)) * np.exp(rng.standard_normal((7, 1)) * 3)
    out = test_sqr_norm(x)
    L = O.lib()
    ref = np.array([L.po_sqr_norm(O._dp(np.ascontiguousarray(row)), d) for row in x])
    assert np.array_equal(out, ref)                   # identical reduction tree: bit-exact


@pytest.mark.parametrize("N,d", [(10, 2), (5, 100), (3, 1024)])
def test_create_replicas_matches_oracle(P, N, d):
    pt, ref = _mk(P, N, d, 1, "toy")
    x, chain, rng = pt.replicas.states()
    xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-14, atol=0)
    np.testing.assert_allclose(pt.replicas.schedule(), ref.schedule(), rtol=0, atol=0)


def test_config1_quickstart_toy(P):
    """BASELINE configs[0] with the target's default explorer: toy_mvn_target(2), N=10, 5 rounds."""
    pt, ref = _mk(P, 10, 2, 5, "toy")
    for _ in range(5):
        _check_round(P, pt, ref)


def test_config1_quickstart_slice(P):
    """BASELINE configs[0]: toy_mvn_target(2), n_chains=10, n_rounds=5, SliceSampler."""
    pt, ref = _mk(P, 10, 2, 5, "slice")
    for _ in range(5):
        _check_round(P, pt, ref)


@pytest.mark.parametrize("N,d,rounds,seed", [
    (6, 10, 7, 1),        # reference test_stepping_stone.jl shape
    (7, 64, 4, 2),        # exactly one block, odd N
    (4, 65, 4, 3),        # ragged second block
    (12, 100, 4, 1),
    (2, 33, 5, 5),        # only reference + target
    (1, 8, 4, 1),         # single chain: no reference, no swaps
    (9, 300, 3, 4),
    (5, 256, 3, 6),       # exactly one 256-coordinate block of the default kernel
    (4, 257, 3, 8),       # ... and one coordinate more
    (3, 4096, 2, 2),      # BASELINE configs[3] dimension: 16 blocks, the deepest reduction tree
])
def test_slice_sampler_parity(P, N, d, rounds, seed):
    pt, ref = _mk(P, N, d, rounds, "slice", seed=seed)
    for _ in range(rounds):
        _check_round(P, pt, ref)


def test_slice_sampler_parity_d1024(P):
    """BASELINE configs[1] dimension (d=1024) at a chain count the O(d^2) oracle finishes in seconds."""
    pt, ref = _mk(P, 6, 1024, 2, "slice")
    for _ in range(2):
        _check_round(P, pt, ref)


@pytest.mark.parametrize("impl", [1, 2, 5, 7, 8])
@pytest.mark.parametrize("N,d,rounds,seed", [(7, 64, 4, 2), (4, 65, 4, 3), (5, 3, 6, 1), (6, 200, 3, 7), (3, 1024, 2, 1)])
def test_every_slice_kernel_version_matches_oracle(P, impl, N, d, rounds, seed):
    """All SliceSampler kernel generations are the same function.  pte_config.debug_kernel selects: 1 (the sequential
    kernel) ships in libpte.so, 2 / 5 / 7 (and 8 named explicitly) only in the test build libpte_test.so."""
    pt, ref = _mk(P, N, d, rounds, "slice", seed=seed, debug_kernel=impl)
    assert pt.replicas.kernel_name() == {1: "k_explore_slice", 2: "k_explore_slice2", 5: "k_explore_slice5", 7: "k_explore_slice7", 8: "k_explore_slice8"}[impl]
    for _ in range(rounds):
        _check_round(P, pt, ref)


def test_product_library_refuses_kernels_it_does_not_contain(P):
    """libpte.so holds the default kernels and their exact fallbacks only; anything else fails loudly in pte_create
    (nothing is selected through the environment)."""
    from pigeons_amd.engine import Engine
    from pigeons_amd import _lib
    for dk in (2, 5, 7, 3, 99, _lib.KERNEL_ISING_BITS):
        with pytest.raises(P.PteError, match="debug_kernel"):
            Engine(n_chains=4, dim=8, explorer=_lib.EXPLORER_SLICE, debug_kernel=dk)
    with pytest.raises(P.PteError, match="debug_kernel"):
        Engine(n_chains=4, dim=8, explorer=_lib.EXPLORER_TOY, debug_kernel=1)
    e = Engine(n_chains=4, dim=8, explorer=_lib.EXPLORER_SLICE, debug_kernel=_lib.KERNEL_SLICE_SEQUENTIAL)
    assert e.kernel_name() == "k_explore_slice"
    assert Engine(n_chains=4, dim=8, explorer=_lib.EXPLORER_SLICE).kernel_name() == "k_explore_slice8"


@pytest.mark.parametrize("N,d,rounds", [(16, 128, 6), (5, 1024, 4), (33, 7, 6)])
def test_toy_explorer_parity(P, N, d, rounds):
    pt, ref = _mk(P, N, d, rounds, "toy")
    for _ in range(rounds):
        _check_round(P, pt, ref)


def test_round_trips_kat_test_swapper(P):
    """reference test/test_round_trips.jl:1-14: TestSwapper(1.0), N=4, 5 rounds => 13 round trips."""
    n_chains, n_rounds = 4, 5
    pt = P.pigeons(target=P.TestSwapper(1.0), record=[P.round_trip, P.index_process], n_chains=n_chains,
                   n_rounds=n_rounds, show_report=False)
    truth = sum(math.floor(max(2 ** n_rounds - i, 0) / n_chains / 2) for i in range(n_chains))
    assert P.n_round_trips(pt) == truth == 13
    ref = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=1.0, n_chains=n_chains, explorer=O.EXPLORER_NONE)
    for _ in range(n_rounds):
        ref.run_round()
    assert np.array_equal(pt.reduced_recorders.index_process, ref.index_process())
    assert pt.reduced_recorders.round_trip == ref.round_trip()


@pytest.mark.parametrize("pr", [0.0, 0.3, 0.7])
def test_test_swapper_parity(P, pr):
    pt = P.pigeons(target=P.TestSwapper(pr), record=[P.round_trip, P.index_process], n_chains=9, n_rounds=6,
                   show_report=False, seed=3)
    ref = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=pr, n_chains=9, explorer=O.EXPLORER_NONE, seed=3)
    for _ in range(6):
        ref.run_round()
    assert np.array_equal(pt.reduced_recorders.index_process, ref.index_process())
    assert pt.reduced_recorders.round_trip == ref.round_trip()


def test_stepping_stone_kat(P):
    """reference test/test_stepping_stone.jl:15-27: |logZ error| < 0.2 at d=10, N=6, 12 rounds."""
    pt = P.pigeons(target=P.toy_mvn_target(10), explorer=P.SliceSampler(), n_chains=6, n_rounds=12, show_report=False)
    p = P.stepping_stone_pair(pt)
    truth = P.analytic_lognormalization(P.toy_mvn_target(10))
    assert abs(p[0] - truth) < 0.2 and abs(p[1] - truth) < 0.2


def test_set_state_roundtrip_and_explore_swap_split(P):
    """pte_explore + pte_swap called separately == pte_run_scans; get/set_state round-trips."""
    a, _ = _mk(P, 8, 40, 4, "slice")
    b, _ = _mk(P, 8, 40, 4, "slice")
    a.replicas.run_scans(1, 6)
    for s in range(1, 7):
        b.replicas.explore(s)
        b.replicas.swap(s)
    xa, ca, ra = a.replicas.states()
    xb, cb, rb = b.replicas.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ra, rb)
    c, _ = _mk(P, 8, 40, 4, "slice", seed=99)
    c.replicas.set_states(xa, ca, ra)
    a.replicas.run_scans(7, 4)
    c.replicas.run_scans(7, 4)
    xa, ca, ra = a.replicas.states()
    xc, cc, rc = c.replicas.states()
    assert np.array_equal(xa, xc) and np.array_equal(ca, cc) and np.array_equal(ra, rc)


def test_errors_are_loud(P):
    with pytest.raises(P.PteError):
        P.Engine(n_chains=4, dim=5000)                # beyond the register-resident tree
    e = P.Engine(n_chains=4, dim=8)
    with pytest.raises(P.PteError):
        e.set_schedule([0.0, 0.5, 0.4, 1.0])          # Schedule validity assert
    with pytest.raises(P.PteError):
        e.set_schedule([0.0, 1.0])
    import pigeons_amd._lib as L
    bad = [dict(n_chains=0), dict(n_chains=4, dim=0), dict(n_chains=4, dim=8, explorer=99), dict(n_chains=5, world_size=2),
           dict(n_chains=4, n_chains_variational=4, world_size=2, rank=0),                         # two legs are single-engine
           dict(n_chains=4, dim=8, explorer=L.EXPLORER_SLICE, explorer2=L.EXPLORER_TOY),            # Compose set
           dict(n_chains=4, dim=8, target=L.TARGET_FUNNEL, explorer=L.EXPLORER_TOY),               # funnel: AutoMALA / MALA / SliceSampler
           dict(n_chains=4, dim=2000, target=L.TARGET_FUNNEL, explorer=L.EXPLORER_SLICE),          # ... register-resident: dim <= 1024
           dict(n_chains=4, dim=8, target=L.TARGET_FUNNEL, explorer=L.EXPLORER_SLICE, debug_kernel=1),
           dict(n_chains=4, dim=2000, explorer=L.EXPLORER_AUTOMALA),                               # register-resident bound
           dict(n_chains=4, dim=49, target=L.TARGET_ISING, explorer=L.EXPLORER_SLICE),
           dict(n_chains=4, dim=8, record_flags=L.RECORD_TRACES_EXTENDED)]                         # extended needs traces
    for kw in bad:
        with pytest.raises(P.PteError):
            P.Engine(**kw)
    with pytest.raises(P.PteError):                       # a variational reference needs the interpolated path
        e.set_variational_reference(np.zeros(8), np.ones(8), np.ones(4, dtype=np.int32))
    f = P.Engine(n_chains=4, dim=6, target=L.TARGET_FUNNEL, explorer=L.EXPLORER_AUTOMALA, target_params=[1.0 / 9.0])
    with pytest.raises(P.PteError):
        f.set_variational_reference(np.zeros(6), np.array([1.0, 1.0, 0.0, 1.0, 1.0, 1.0]), np.ones(4, dtype=np.int32))   # std must be > 0
    cfg = L.PteConfig(); L.load().pte_default_config(cfg); cfg.struct_size += 8
    import ctypes as C
    h = C.c_void_p()
    assert L.load().pte_create(C.byref(cfg), C.byref(h)) != 0 and b"ABI mismatch" in L.load().pte_last_error(None)


def test_full_size_properties_metric_config(P):
    """BASELINE metric configuration (d=1024, N=1024, SliceSampler): size-independent properties."""
    from pigeons_amd.engine import test_sqr_norm
    N, d = 1024, 1024
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=3, explorer=P.SliceSampler(),
                       record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False))
    x0, _, rng0 = pt.replicas.states()
    for r in range(1, 4):
        assert P.next_round(pt)
        red = P.run_one_round(pt)
        P.adapt(pt, red)
        ip = red.index_process
        assert ip.shape == (N, 2 ** r)
        # every scan's chain assignment is a permutation; DEO moves a replica by at most one chain
        assert np.array_equal(np.sort(ip, axis=0), np.tile(np.arange(N)[:, None], (1, 2 ** r)))
        assert np.all(np.abs(np.diff(ip, axis=1)) <= 1)
        m, n = red.swap_acceptance_pr
        assert np.all((m >= 0) & (m <= 1)) and np.all(n == 2 ** (r - 1))
        g = pt.shared.tempering.schedule.grids
        assert g[0] == 0.0 and g[-1] == 1.0 and np.all(np.diff(g) > 0)
    x, chain, rng = pt.replicas.states()
    assert np.array_equal(np.sort(chain), np.arange(N))
    assert np.all(rng[:, 1] == rng0[:, 1]) and np.all(rng[:, 0] != rng0[:, 0])   # gammas fixed, seeds advanced
    assert np.all(np.isfinite(x))
    # coordinates are resampled by every slice pass: nothing stuck at its initial value
    moved = np.mean(x != x0)
    assert moved > 0.999, moved
    # the incrementally maintained tree root equals a full recompute (device) and the oracle's tree
    L = O.lib()
    full_dev = test_sqr_norm(x)
    full_ref = np.array([L.po_sqr_norm(O._dp(np.ascontiguousarray(row)), d) for row in x])
    assert np.array_equal(full_dev, full_ref)
    # marginal scale sanity: chain c has precision in [1, 10]
    prec = 1.0 + 9.0 * pt.shared.tempering.schedule.grids
    v = np.var(x, axis=1)[np.argsort(chain)]
    assert np.all(v < 2.0 / prec * 1.5) and np.all(v > 0.5 / prec / 1.5)


# ---------------------------------------------------------------------------------------------
# chain sharding: G shard engines on ONE GPU, boundary bytes moved by the host.
# The output must be identical to the single-engine run for any G (parallelism invariance).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,d,G,explorer,rounds", [
    (8, 40, 2, "slice", 5),
    (12, 100, 4, "slice", 4),
    (12, 7, 3, "toy", 6),       # K = 4
    (9, 5, 3, "toy", 6),        # K = 3 (odd): boundary pairs active on both graph parities
    (16, 130, 8, "slice", 4),   # K = 2
    (6, 3, 6, "toy", 5),        # K = 1: every pair is a boundary pair
])
@pytest.mark.parametrize("device_messages", [False, True])   # True: the stream-ordered RCCL path's kernels
def test_sharded_engines_equal_single_engine(P, N, d, G, explorer, rounds, device_messages):
    exp = {"toy": P.ToyExplorer, "slice": P.SliceSampler}[explorer]
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online]
    mk = lambda: P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp(), record=rec, show_report=False)
    one = P.PT(mk())
    many = P.PT(mk(), n_shards=G, device_messages=device_messages)
    for _ in range(rounds):
        assert P.next_round(one) and P.next_round(many)
        ra = P.run_one_round(one); P.adapt(one, ra)
        rb = P.run_one_round(many); P.adapt(many, rb)
        assert np.array_equal(ra.index_process, rb.index_process)
        assert ra.round_trip == rb.round_trip
        for a, b in zip(ra.swap_acceptance_pr + ra.log_sum_ratio + ra.explorer_acceptance_pr + ra.explorer_n_steps,
                        rb.swap_acceptance_pr + rb.log_sum_ratio + rb.explorer_acceptance_pr + rb.explorer_n_steps):
            assert np.array_equal(a, b)                 # same arithmetic in the same order: bit-identical
        assert np.array_equal(one.shared.tempering.schedule.grids, many.shared.tempering.schedule.grids)
        assert np.array_equal(ra.online[0], rb.online[0]) and np.array_equal(ra.online[1], rb.online[1])
    xa, ca, ga = one.replicas.states()
    xb, cb, gb = many.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    assert many.shards.n_boundary_swaps > 0           # the exchange path was exercised


def test_sharded_test_swapper_round_trips(P):
    pt = P.PT(P.Inputs(target=P.TestSwapper(1.0), n_chains=4, n_rounds=5, record=[P.round_trip, P.index_process],
                       show_report=False), n_shards=2)
    P.pigeons(pt)
    assert P.n_round_trips(pt) == 13


DIST_WORKER = r'''
import os, sys, json
import numpy as np
sys.path[:0] = [%(root)r, %(root)r + "/pigeons.jl_amd", %(root)r + "/tests"]
import torch, torch.distributed as dist
import pigeons_amd as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share cuda:0 (nccl refuses that)
mk = lambda: P.Inputs(target=P.toy_mvn_target(70), n_chains=8, n_rounds=5, explorer=P.SliceSampler(), show_report=False,
                      record=[P.round_trip, P.index_process, P.log_sum_ratio])
pt = P.PT(mk(), rank=rank, world=world, dist_device=torch.device("cuda", 0), transport="host")   # host-driven two-phase exchange, device payload buffers
one = P.PT(mk()) if rank == 0 else None
ok = True
for _ in range(5):
    P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
    if rank == 0:
        P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
        ok &= bool(np.array_equal(ra.index_process, red.index_process)) and ra.round_trip == red.round_trip
        ok &= bool(np.array_equal(ra.swap_acceptance_pr[0], red.swap_acceptance_pr[0]))
        ok &= bool(np.array_equal(one.shared.tempering.schedule.grids, pt.shared.tempering.schedule.grids))
x, chain, rng = pt.shards.states()
if rank == 0:
    xa, ca, ga = one.replicas.states()
    ok &= bool(np.array_equal(x, xa) and np.array_equal(chain, ca) and np.array_equal(rng, ga))
    print(json.dumps({"ok": ok, "boundary_swaps": pt.shards.n_boundary_swaps}))
dist.destroy_process_group()
'''


def test_dist_shard_two_ranks_device_payloads(P, tmp_path):
    """DistShard (the host-driven two-phase driver) with device-resident payload buffers, two ranks over gloo on one GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(DIST_WORKER % {"root": root})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["ok"] and res["boundary_swaps"] > 0, res


RCCL_WORLD1_WORKER = r'''
import os, sys, json
import numpy as np
sys.path[:0] = [%(root)r, %(root)r + "/pigeons.jl_amd", %(root)r + "/tests"]
import pigeons_amd as P
from pigeons_amd.engine import comm_unique_id
mk = lambda: P.Inputs(target=P.toy_mvn_target(70), n_chains=8, n_rounds=4, explorer=P.SliceSampler(), show_report=False,
                      record=[P.round_trip, P.index_process, P.log_sum_ratio])
one = P.PT(mk())
from pigeons_amd.sharded import RcclShard
pt = P.PT(mk())
pt.shards = RcclShard(pt.replicas, 0, 1, id_bytes=comm_unique_id())       # ncclCommInitRank inside libpte, no torch.distributed
ok = pt.shards.n_ranks_seen == 1 and pt.replicas.comm_info()[0] == 1
for _ in range(4):
    P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
    P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
    ok &= bool(np.array_equal(ra.index_process, red.index_process)) and ra.round_trip == red.round_trip
    ok &= bool(np.array_equal(ra.swap_acceptance_pr[0], red.swap_acceptance_pr[0]))
x, chain, rng = pt.shards.states(); xa, ca, ga = one.replicas.states()
ok &= bool(np.array_equal(x, xa) and np.array_equal(chain, ca) and np.array_equal(rng, ga))
pt.shards.barrier()
ok &= float(pt.shards.allreduce_max([3.5])[0]) == 3.5
print(json.dumps({"ok": ok, "torch_distributed_loaded": "torch.distributed" in sys.modules and sys.modules["torch.distributed"].is_initialized()}))
'''


def test_rccl_transport_behind_the_abi_world1(P, tmp_path):
    # The RCCL transport of libpte (pte_comm_unique_id / pte_comm_init / collectives; RCCL mapped with dlopen) on a 1-rank
    # communicator: no peers, but communicator creation, the sharded pte_run_scans entry, the gather plumbing run --
    # and no torch.distributed process group exists anywhere in the process.
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker1.py"
    script.write_text(RCCL_WORLD1_WORKER % {"root": root})
    p = subprocess.run([sys.executable, str(script)], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and not res["torch_distributed_loaded"], res


@pytest.mark.parametrize("G,N,d,explorer", [(2, 8, 70, "slice"), (4, 8, 33, "slice"), (3, 9, 20, "toy"), (8, 16, 130, "slice"), (4, 4, 16, "slice")])
def test_group_transport_equals_single_engine(P, G, N, d, explorer):
    # pte_group_run_scans: G engines of one process driven by the library itself (stream-ordered device copies between the
    # pack and decide kernels, cross-stream events) -- the kernels and the ordering of the RCCL path, bit-identical to G = 1.
    exp = lambda: P.SliceSampler() if explorer == "slice" else P.ToyExplorer()
    mk = lambda: P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=6, explorer=exp(), show_report=False, seed=5,
                          record=[P.round_trip, P.index_process, P.log_sum_ratio])
    one = P.PT(mk()); grp = P.PT(mk(), n_shards=G, transport="group")
    for _ in range(6):
        P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
        P.next_round(grp); rb = P.run_one_round(grp); P.adapt(grp, rb)
        assert np.array_equal(ra.index_process, rb.index_process) and ra.round_trip == rb.round_trip
        assert np.array_equal(ra.swap_acceptance_pr[0], rb.swap_acceptance_pr[0])
        assert np.array_equal(ra.log_sum_ratio[0], rb.log_sum_ratio[0])
    xa, ca, ga = one.replicas.states(); xb, cb, gb = grp.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    if N > G:
        assert grp.shards.n_boundary_swaps > 0


def test_sharded_run_scans_without_a_communicator_fails_loudly(P):
    from pigeons_amd.engine import Engine
    from pigeons_amd import _lib
    e = Engine(n_chains=8, dim=8, explorer=_lib.EXPLORER_SLICE, rank=0, world_size=2)
    with pytest.raises(P.PteError, match="pte_comm_init"):
        e.run_scans(1, 2)


# ---------------------------------------------------------------------------------------------
# AutoMALA: MVN path with the analytic gradient, and BASELINE config 3,
# Neal's funnel through the linear InterpolatingPath from a normal reference.
# Integer outputs exact; floats to 1e-6 relative (north star) -- the funnel evaluates exp/log on
# ocml (device) vs glibc (oracle), <= 1 ulp apart per call.
# ---------------------------------------------------------------------------------------------
def _mk_am(P, N, d, rounds, target="mvn", seed=1, precond=None):
    precond = precond or P.MixDiagonalPreconditioner()
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    kind = {"IdentityPreconditioner": 0, "DiagonalPreconditioner": 1, "MixDiagonalPreconditioner": 2}[type(precond).__name__]
    if target == "mvn":
        inp = P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=P.AutoMALA(preconditioner=precond),
                       seed=seed, record=rec, show_report=False)
        ref = O.OraclePT(n_chains=N, dim=d, seed=seed, explorer=O.EXPLORER_AUTOMALA, am_preconditioner=kind)
    else:
        inp = P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N,
                       n_rounds=rounds, explorer=P.AutoMALA(preconditioner=precond), seed=seed, record=rec, show_report=False)
        ref = O.OraclePT(n_chains=N, dim=d, seed=seed, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1.0 / 9.0,
                         am_preconditioner=kind)
    return P.PT(inp), ref


def _check_am_round(P, pt, ref, rtol, acc_rtol=None, state_atol=None):
    assert P.next_round(pt)
    red = P.run_one_round(pt)
    P.adapt(pt, red)
    pt.reduced_recorders = red
    ref.run_round()
    assert np.array_equal(red.index_process, ref.index_process())
    assert red.round_trip == ref.round_trip()
    ss, sn = red.explorer_n_steps
    amr, anr, ssr, snr = ref.explorer_stats()
    assert np.array_equal(sn, snr) and np.array_equal(ss, ssr)         # leapfrog counts: integers
    fm, fn = red.am_factors
    rm, rn = red.reversibility_rate
    fmr, fnr, rmr, rnr = ref.am_stats()
    assert np.array_equal(fn, fnr) and np.array_equal(rn, rnr)
    np.testing.assert_allclose(fm, fmr, rtol=1e-12)                      # means of exact powers of two
    np.testing.assert_allclose(rm, rmr, rtol=1e-12)
    am, an = red.explorer_acceptance_pr
    assert np.array_equal(an, anr)
    np.testing.assert_allclose(am, amr, rtol=acc_rtol or rtol, atol=1e-300)
    m, n = red.swap_acceptance_pr
    mr, nr = ref.swap_pr()
    assert np.array_equal(n, nr)
    np.testing.assert_allclose(m, mr, rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=rtol)
    ex = pt.shared.explorer
    if hasattr(ex, "first"):                                               # Compose: the gradient-based component
        ex = ex.first if hasattr(ex.first, "step_size") else ex.second
    np.testing.assert_allclose(ex.step_size, ref.step_size(), rtol=1e-12)
    std = ref.target_std()
    if std is not None:
        np.testing.assert_allclose(ex.estimated_target_std_deviations, std, rtol=1e-6, atol=1e-12)
    x, chain, rng = pt.replicas.states()
    xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=rtol, atol=state_atol or 1e-9 * rtol / 1e-6)


@pytest.mark.parametrize("N,d,rounds,seed", [(6, 10, 7, 1), (5, 64, 5, 2), (8, 128, 5, 1), (4, 200, 4, 3), (3, 1000, 3, 1)])
def test_automala_mvn_parity(P, N, d, rounds, seed):
    pt, ref = _mk_am(P, N, d, rounds, "mvn", seed)
    for _ in range(rounds):
        _check_am_round(P, pt, ref, rtol=1e-9)


def test_automala_follows_the_rng_policy(P):
    """libpte is built from two translation units and each holds its own copy of the policy word (include/pte_rng_policy.h;
    csrc/pte_automala_params.hpp): pte_set_rng_policy must reach the Langevin-family kernels too.  AutoMALA draws ~0.3 M momentum
    normals here, ~100 of them on the ziggurat's tail, whose formula the policy picks: engine == oracle under -log1p(-u), and the
    oracle under the default policy is somewhere else (the switch is live)."""
    from pigeons_amd.engine import set_rng_policy
    pol = O.RNG_TAIL_LOG1P
    N, d, rounds, seed = 24, 256, 4, 5
    try:
        set_rng_policy(pol); O.set_rng_policy(pol)
        pt, ref = _mk_am(P, N, d, rounds, "mvn", seed)
        for _ in range(rounds):
            _check_am_round(P, pt, ref, rtol=1e-9)
        x_pol = ref.states()[0].copy()
        set_rng_policy(0); O.set_rng_policy(0)
        _, ref0 = _mk_am(P, N, d, rounds, "mvn", seed)
        for _ in range(rounds):
            ref0.run_round()
        assert not np.array_equal(ref0.states()[0], x_pol)
    finally:
        set_rng_policy(0); O.set_rng_policy(0)


@pytest.mark.parametrize("precond", ["IdentityPreconditioner", "DiagonalPreconditioner"])
def test_automala_other_preconditioners(P, precond):
    pt, ref = _mk_am(P, 5, 20, 5, "mvn", 4, precond=getattr(P, precond)())
    for _ in range(5):
        _check_am_round(P, pt, ref, rtol=1e-9)


@pytest.mark.parametrize("N,d,rounds,seed", [(8, 8, 7, 1), (8, 128, 5, 1), (6, 70, 5, 2)])
def test_automala_funnel_parity(P, N, d, rounds, seed):
    """BASELINE configs[2] family: Neal's funnel, AutoMALA (d=128 at a chain count the oracle finishes fast)."""
    pt, ref = _mk_am(P, N, d, rounds, "funnel", seed)
    for _ in range(rounds):
        _check_am_round(P, pt, ref, rtol=1e-6)


@pytest.mark.parametrize("path,N,d,rounds,seed", [("mvn", 4, 256, 4, 1), ("mvn", 3, 512, 3, 2), ("mvn", 3, 1024, 3, 1),
                                                  ("funnel", 5, 64, 5, 3), ("funnel", 4, 256, 4, 1), ("funnel", 3, 300, 3, 2),
                                                  ("funnel", 3, 512, 3, 1), ("funnel", 3, 700, 3, 2), ("funnel", 3, 1024, 3, 1)])
def test_automala_every_register_layout_against_the_oracle(P, path, N, d, rounds, seed):
    """Every instantiation of k_explore_automala: E = 1 .. 16 blocks of 64 coordinates per replica, with and without a ragged last
    block (d = 64 E takes the mask-free instantiation), on both paths -- incl. d = 1024 (E = 16), the largest the Langevin kernels take.
    The reductions differ by layout (one lockstep pass of eight packed chains at E <= 4 on the funnel, four at a time beyond), the
    reuse of evaluations (start-point gradient carried, proposal = the search's kept trial) does not."""
    pt, ref = _mk_am(P, N, d, rounds, path, seed)
    for _ in range(rounds):
        _check_am_round(P, pt, ref, rtol=1e-9 if path == "mvn" else 1e-6)


def test_automala_stepping_stone_kat(P):
    """reference test/test_stepping_stone.jl:15-27 with AutoMALA(): |logZ error| < 0.2 at d=10, N=6, 12 rounds."""
    pt = P.pigeons(target=P.toy_mvn_target(10), explorer=P.AutoMALA(), n_chains=6, n_rounds=12, show_report=False)
    p = P.stepping_stone_pair(pt)
    truth = P.analytic_lognormalization(P.toy_mvn_target(10))
    assert abs(p[0] - truth) < 0.2 and abs(p[1] - truth) < 0.2


def test_config3_funnel_full_size_properties(P):
    """BASELINE configs[2]: funnel d=128, n_chains=1024, AutoMALA -- size-independent properties."""
    N, d = 1024, 128
    pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N, n_rounds=4,
                       explorer=P.AutoMALA(), record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False))
    for r in range(1, 5):
        assert P.next_round(pt)
        red = P.run_one_round(pt)
        P.adapt(pt, red)
        ip = red.index_process
        assert np.array_equal(np.sort(ip, axis=0), np.tile(np.arange(N)[:, None], (1, 2 ** r)))
        assert np.all(np.abs(np.diff(ip, axis=1)) <= 1)
        fm, fn = red.am_factors
        assert fn[0] == 0 and np.all(fn[1:] > 0)                       # chain 1 is refreshed i.i.d.
        assert np.all(np.log2(fm[1:] * 1.0) < 8)
        rm, rn = red.reversibility_rate
        assert np.all((rm >= 0) & (rm <= 1))
        assert pt.shared.explorer.step_size > 0
    x, chain, rng = pt.replicas.states()
    assert np.all(np.isfinite(x)) and np.array_equal(np.sort(chain), np.arange(N))


# ---------------------------------------------------------------------------------------------
# SliceSampler on a path WITHOUT a closed-form single-coordinate update (slice_sample! takes any
# log_potential, SliceSampler.jl:105-118): the interpolated funnel path, every proposal a full log-potential evaluation.
# ---------------------------------------------------------------------------------------------
def _mk_slice_funnel(P, N, d, rounds, seed=1, explorer=None, okw=None):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    inp = P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N, n_rounds=rounds,
                   explorer=explorer or P.SliceSampler(), seed=seed, record=rec, show_report=False)
    ref = O.OraclePT(n_chains=N, dim=d, seed=seed, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, **(okw or dict(explorer=O.EXPLORER_SLICE)))
    return P.PT(inp), ref


def _check_funnel_round(P, pt, ref, rtol=1e-6):
    assert P.next_round(pt)
    red = P.run_one_round(pt); P.adapt(pt, red)
    ref.run_round()
    assert np.array_equal(red.index_process, ref.index_process()) and red.round_trip == ref.round_trip()
    ss, sn = red.explorer_n_steps; am, an = red.explorer_acceptance_pr
    amr, anr, ssr, snr = ref.explorer_stats()
    assert np.array_equal(sn, snr) and np.array_equal(ss, ssr) and np.array_equal(an, anr)      # every step count, exactly
    np.testing.assert_allclose(am, amr, rtol=rtol)
    m, n = red.swap_acceptance_pr; mr, nr = ref.swap_pr()
    assert np.array_equal(n, nr)
    np.testing.assert_allclose(m, mr, rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=rtol)
    np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=rtol)
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)                                # same draws consumed by every replica
    np.testing.assert_allclose(x, xr, rtol=rtol, atol=1e-12)


@pytest.mark.parametrize("N,d,rounds,seed", [(6, 3, 5, 1), (5, 64, 4, 2), (4, 65, 4, 3), (5, 128, 3, 1), (1, 4, 4, 5)])
def test_slice_sampler_on_the_funnel_path_parity(P, N, d, rounds, seed):
    pt, ref = _mk_slice_funnel(P, N, d, rounds, seed)
    assert pt.replicas.kernel_name() == "k_explore_automala"       # the register-resident path kernel, in its SliceSampler mode
    for _ in range(rounds):
        _check_funnel_round(P, pt, ref)


def test_slice_sampler_funnel_compose_shards_and_errors(P):
    # Compose(SliceSampler, AutoMALA) on the funnel path against the oracle
    pt, ref = _mk_slice_funnel(P, 5, 20, 4, explorer=P.Compose(P.SliceSampler(), P.AutoMALA()),
                               okw=dict(explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA, am_preconditioner=2))
    for _ in range(4):
        _check_funnel_round(P, pt, ref)
    # chain shards (the library's group transport) == one engine, bit for bit
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    mk = lambda: P.Inputs(target=P.Funnel(12), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 12), n_chains=8, n_rounds=5,
                          explorer=P.SliceSampler(), record=rec, show_report=False, seed=3)
    one, many = P.PT(mk()), P.PT(mk(), n_shards=4, transport="group")
    for _ in range(5):
        P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
        P.next_round(many); rb = P.run_one_round(many); P.adapt(many, rb)
        assert np.array_equal(ra.index_process, rb.index_process) and np.array_equal(ra.explorer_n_steps[0], rb.explorer_n_steps[0])
    xa, ca, ga = one.replicas.states(); xb, cb, gb = many.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    # slice_shrink!'s iteration cap is an error here too; a GaussianReference under it is refused, not approximated
    bad = P.PT(P.Inputs(target=P.Funnel(6), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 6), n_chains=4, n_rounds=6,
                        explorer=P.SliceSampler(w=1000.0, max_iter=2), record=[P.log_sum_ratio], show_report=False))
    with pytest.raises(P.PteError, match="Maximum number of iterations"):
        for _ in range(6):
            P.next_round(bad); P.run_one_round(bad)
    with pytest.raises(P.PteError, match="GaussianReference under SliceSampler"):
        v = P.PT(P.Inputs(target=P.Funnel(6), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 6), n_chains=4, n_rounds=4,
                          explorer=P.SliceSampler(), variational=P.GaussianReference(first_tuning_round=1), show_report=False))
        for _ in range(3):
            P.next_round(v); r = P.run_one_round(v); P.adapt(v, r)


# ---------------------------------------------------------------------------------------------
# 2-D Ising (BASELINE configs[4], reference examples/ising.jl): integer path.  Spins, chains, RNG
# counters and the index process must be bit-exact; sum_pair_products is an integer.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,N,rounds,beta,seed", [(5, 10, 7, 1.0, 1), (8, 6, 6, 0.6, 2), (16, 5, 5, 1.0, 3), (32, 4, 3, 0.44, 1)])
def test_ising_parity(P, L, N, rounds, beta, seed):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(beta, L), n_chains=N, n_rounds=rounds, seed=seed, record=rec, show_report=False))
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=beta, n_chains=N, seed=seed, slice_n_passes=3)
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt)
        P.adapt(pt, red)
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        assert red.round_trip == ref.round_trip()
        m, n = red.swap_acceptance_pr
        mr, nr = ref.swap_pr()
        assert np.array_equal(n, nr)
        np.testing.assert_allclose(m, mr, rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=RTOL)
        np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=RTOL)
        x, chain, rng = pt.replicas.states()
        xr, cr, rr = ref.states()
        assert np.array_equal(x, xr) and np.array_equal(chain, cr) and np.array_equal(rng, rr)


@pytest.mark.parametrize("L", [8, 32, 64])           # the byte kernel, the one-word-per-row and the general instantiation of the speculative kernel
def test_ising_ladders_with_tiny_betas(P, L):
    """the second chain of a ladder adapted on a handful of scans sits at beta ~ 1e-7 (C5 after its round 2: 2.6e-7): the guard-banded
    thresholds decide there too (valid while 4 beta beta_target >> 2^-53: PTE_ISING_FILTER_MIN = 1e-13; rounds 3-5 sent every decision
    of such a chain through the exact arithmetic with a recount of the lattice, 207 ms per scan at 256 x 256), and below the limit the
    exact arithmetic still takes over -- states, chains and RNG counters equal the oracle's either way"""
    betas = np.array([0.0, 1e-18, 3e-16, 5e-14, 2e-13, 1e-11, 2.6e-7, 5e-6, 3e-3, 0.2, 0.6, 1.0])
    N = len(betas)
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, L), n_chains=N, n_rounds=6, seed=5, record=[P.round_trip, P.index_process], show_report=False))
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=1.0, n_chains=N, seed=5, slice_n_passes=3)
    pt.replicas.set_schedule(betas); ref.set_schedule(betas)
    ref.begin_round()
    for k in range(3):
        pt.replicas.run_scans(1 + 8 * k, 8); ref.run_scans(8)
        x, chain, rng = pt.replicas.states()
        xr, cr, rr = ref.states()
        assert np.array_equal(chain, cr) and np.array_equal(rng, rr) and np.array_equal(x, xr)


def test_ising_sharded_equals_single(P):
    mk = lambda: P.Inputs(target=P.IsingLogPotential(1.0, 8), n_chains=8, n_rounds=5, show_report=False,
                          record=[P.round_trip, P.index_process, P.log_sum_ratio])
    one, many = P.PT(mk()), P.PT(mk(), n_shards=4)
    for _ in range(5):
        P.next_round(one); P.next_round(many)
        ra = P.run_one_round(one); P.adapt(one, ra)
        rb = P.run_one_round(many); P.adapt(many, rb)
        assert np.array_equal(ra.index_process, rb.index_process) and ra.round_trip == rb.round_trip
    xa, ca, ga = one.replicas.states()
    xb, cb, gb = many.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    assert many.shards.n_boundary_swaps > 0


def test_ising_ground_state_and_logz(P):
    """5x5 periodic lattice at beta = 1: the target chain freezes into a ground state and
    log(Z1/Z0) ~ log(2 e^50 / 2^25) = 33.36 (50 bonds)."""
    pt = P.pigeons(target=P.IsingLogPotential(1.0, 5), n_chains=10, n_rounds=10, show_report=False)
    assert abs(P.stepping_stone(pt) - (50 + math.log(2) - 25 * math.log(2))) < 0.5


def test_config5_ising_full_size_properties(P):
    """BASELINE configs[4] shape per GPU: 256 x 256 spins (a few chains; the sweep is sequential per replica)."""
    L, N = 256, 16
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(1.0, L), n_chains=N, n_rounds=2, show_report=False,
                       record=[P.round_trip, P.index_process, P.log_sum_ratio]))
    for r in range(1, 3):
        assert P.next_round(pt)
        red = P.run_one_round(pt)
        P.adapt(pt, red)
        assert np.array_equal(np.sort(red.index_process, axis=0), np.tile(np.arange(N)[:, None], (1, 2 ** r)))
    x, chain, rng = pt.replicas.states()
    assert set(np.unique(x)) <= {0.0, 1.0} and np.array_equal(np.sort(chain), np.arange(N))
    # magnetisation grows with beta: hot chains are disordered, the target chain is mostly aligned after a few sweeps
    mag = np.abs(2 * x.mean(axis=1) - 1)[np.argsort(chain)]
    assert mag[0] < 0.05 and mag[-1] > mag[0]


@pytest.mark.parametrize("L,N", [(32, 5), (64, 4)])
def test_ising_bitpacked_kernel_equals_byte_kernel_and_oracle(P, L, N):
    """L % 32 == 0 selects the lane-speculative bit-packed kernel; debug_kernel = PTE_KERNEL_ISING_BITS / _BYTES select the
    scalar bit-packed (test build) and the byte-lattice kernels.  All three are the same function."""
    from pigeons_amd import _lib
    mk = lambda dk: P.PT(P.Inputs(target=P.IsingLogPotential(0.5, L), n_chains=N, n_rounds=3, show_report=False, seed=7,
                                  record=[P.round_trip, P.index_process, P.log_sum_ratio]), debug_kernel=dk)
    pts = {}
    for impl, dk in (("spec", 0), ("bits", _lib.KERNEL_ISING_BITS), ("bytes", _lib.KERNEL_ISING_BYTES)):
        pts[impl] = mk(dk)
        assert pts[impl].replicas.kernel_name() == {"spec": "k_explore_ising_spec", "bits": "k_explore_ising_bits", "bytes": "k_explore_ising"}[impl]
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=0.5, n_chains=N, seed=7, slice_n_passes=3)
    for _ in range(3):
        ref.run_round()
        for impl, x in pts.items():
            P.next_round(x); r = P.run_one_round(x); P.adapt(x, r)
            assert np.array_equal(r.index_process, ref.index_process()), impl
            np.testing.assert_allclose(P.stepping_stone_pair(x), ref.stepping_stone_pair(), rtol=RTOL)
    for impl, x in pts.items():
        xs, cs, rs = x.replicas.states()
        xr, cr, rr = ref.states()
        assert np.array_equal(xs, xr) and np.array_equal(cs, cr) and np.array_equal(rs, rr), impl


# ---------------------------------------------------------------------------------------------
# target-chain sample recorders on the device -- traces ([state; log density]
# per scan), online over the same d+1 entries, energy_ac1 (per-chain correlation of the log density
# before / after explore!).
# ---------------------------------------------------------------------------------------------
def _check_sample_recorders(P, red, ref, rtol, N, exact_traces=True):
    cor, cn, mom = red.energy_ac1
    corr, cnr, rawr = ref.energy_ac1()
    assert np.array_equal(cn, cnr)
    ok = np.isfinite(corr)
    np.testing.assert_allclose(cor[ok], corr[ok], rtol=max(rtol, 1e-7), atol=1e-9)   # A - b b' cancels ~1e3 * eps in the reference's form
    np.testing.assert_allclose(mom[:, :2], rawr[:, :2], rtol=rtol)                   # running means of before / after
    tr, trr = red.traces, ref.traces()
    if trr.size:
        assert tr.shape == trr.shape
        if exact_traces:
            np.testing.assert_allclose(tr, trr, rtol=1e-12, atol=1e-300)
        else:
            np.testing.assert_allclose(tr, trr, rtol=rtol, atol=1e-9)
        lm, lv, ln = ref.online_lp()
        if ln:
            np.testing.assert_allclose(red.online_log_density, (lm, lv), rtol=max(rtol, 1e-9))


@pytest.mark.parametrize("kind,N,d,rounds", [("slice", 6, 10, 6), ("slice", 5, 130, 4), ("toy", 7, 65, 5), ("slice", 1, 8, 5)])
def test_traces_online_energy_ac1_parity_mvn(P, kind, N, d, rounds):
    exp = {"toy": P.ToyExplorer(), "slice": P.SliceSampler()}[kind]
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces, P.energy_ac1]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=exp, record=rec, show_report=False))
    ref = O.OraclePT(n_chains=N, dim=d, explorer={"toy": O.EXPLORER_TOY, "slice": O.EXPLORER_SLICE}[kind], record_online=1,
                     record_traces=1, record_energy_ac1=1)
    for r in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); pt.reduced_recorders = red
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        _check_sample_recorders(P, red, ref, 1e-9, N)
        # the reference's own assertions (test/test_traces.jl:16-27,50-58)
        assert P.sample_array(pt).shape == (2 ** (r + 1), d + 1, 1)
        assert len(P.sample_names(pt)) == d + 1 and "log_density" in P.sample_names(pt)
        marginal = np.array([P.get_sample(pt, N, i + 1)[0] for i in range(2 ** (r + 1))])
        assert abs(marginal.mean() - P.mean(pt)[0]) < 1e-10
        assert np.array_equal(P.get_sample(pt, N)[:, 0], marginal)
    assert len(P.energy_ac1s(pt)) == N and len(P.energy_ac1s(pt, True)) == max(N - 1, 1)


@pytest.mark.parametrize("target,N,d,rounds,rtol", [("mvn", 5, 20, 6, 1e-9), ("funnel", 6, 8, 6, 1e-6), ("funnel", 4, 70, 4, 1e-6)])
def test_traces_energy_ac1_parity_automala(P, target, N, d, rounds, rtol):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.traces, P.energy_ac1]
    if target == "mvn":
        inp = P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=P.AutoMALA(), record=rec, show_report=False)
        ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_AUTOMALA, am_preconditioner=2, record_traces=1, record_energy_ac1=1)
    else:
        inp = P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N,
                       n_rounds=rounds, explorer=P.AutoMALA(), record=rec, show_report=False)
        ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, am_preconditioner=2,
                         record_traces=1, record_energy_ac1=1)
    pt = P.PT(inp)
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        _check_sample_recorders(P, red, ref, rtol, N, exact_traces=False)


def test_energy_ac1_parity_ising(P):
    L, N, rounds = 8, 6, 6
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1]
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(0.6, L), n_chains=N, n_rounds=rounds, seed=2, record=rec, show_report=False))
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=0.6, n_chains=N, seed=2, slice_n_passes=3,
                     record_energy_ac1=1)
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red)
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        cor, cn, mom = red.energy_ac1
        corr, cnr, rawr = ref.energy_ac1()
        assert np.array_equal(cn, cnr)
        np.testing.assert_allclose(mom[:, :2], rawr[:, :2], rtol=1e-12, atol=1e-12)
        ok = np.isfinite(corr)
        np.testing.assert_allclose(cor[ok], corr[ok], rtol=1e-7, atol=1e-9)


def test_sharded_traces_and_energy_ac1_equal_single_engine(P):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.traces, P.energy_ac1]
    mk = lambda: P.Inputs(target=P.toy_mvn_target(33), n_chains=8, n_rounds=4, explorer=P.SliceSampler(), record=rec, show_report=False)
    one, many = P.PT(mk()), P.PT(mk(), n_shards=4, device_messages=True)
    for _ in range(4):
        assert P.next_round(one) and P.next_round(many)
        ra = P.run_one_round(one); P.adapt(one, ra)
        rb = P.run_one_round(many); P.adapt(many, rb)
        assert np.array_equal(ra.traces, rb.traces)
        for a, b in zip(ra.energy_ac1, rb.energy_ac1):
            assert np.array_equal(a, b, equal_nan=True)
        assert ra.online_log_density == rb.online_log_density


# ---------------------------------------------------------------------------------------------
# MALA (src/explorers/MALA.jl) and Compose (src/explorers/Compose.jl) --
# the explorer set of the reference's own invariance matrix (test/test_parallelism_invariance.jl:19).
# ---------------------------------------------------------------------------------------------
def _explorer_pair(P, name):
    """(device explorer object, oracle kwargs)"""
    return {
        "mala": (P.MALA(step_size=0.25), dict(explorer=O.EXPLORER_MALA, am_step_size=0.25, am_preconditioner=2)),
        "mala_id": (P.MALA(step_size=0.4, preconditioner=P.IdentityPreconditioner(), base_n_refresh=5),
                    dict(explorer=O.EXPLORER_MALA, am_step_size=0.4, am_preconditioner=0, am_base_n_refresh=5)),
        "slice+automala": (P.Compose(P.SliceSampler(), P.AutoMALA()),
                           dict(explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA, am_preconditioner=2)),
        "automala+slice": (P.Compose(P.AutoMALA(), P.SliceSampler()),
                           dict(explorer=O.EXPLORER_AUTOMALA, explorer2=O.EXPLORER_SLICE, am_preconditioner=2)),
        "slice+mala": (P.Compose(P.SliceSampler(n_passes=1), P.MALA(step_size=0.3)),
                       dict(explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_MALA, slice_n_passes=1, am_step_size=0.3, am_preconditioner=2)),
    }[name]


@pytest.mark.parametrize("name,N,d,rounds", [("mala", 5, 10, 7), ("mala_id", 4, 70, 5), ("slice+automala", 4, 1, 8),
                                             ("slice+automala", 6, 40, 5), ("automala+slice", 5, 12, 6), ("slice+mala", 5, 130, 4)])
def test_mala_and_compose_parity_mvn(P, name, N, d, rounds):
    ex, okw = _explorer_pair(P, name)
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1, P.traces]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=ex, record=rec, show_report=False))
    ref = O.OraclePT(n_chains=N, dim=d, record_energy_ac1=1, record_traces=1, **okw)
    for _ in range(rounds):
        _check_am_round(P, pt, ref, 1e-9) if "automala" in name else _check_mala_round(P, pt, ref, 1e-9)
        _check_sample_recorders(P, pt.reduced_recorders, ref, 1e-9, N, exact_traces=False)


def _check_mala_round(P, pt, ref, rtol):
    assert P.next_round(pt)
    red = P.run_one_round(pt)
    P.adapt(pt, red)
    pt.reduced_recorders = red
    ref.run_round()
    assert np.array_equal(red.index_process, ref.index_process())
    assert red.round_trip == ref.round_trip()
    am, an = red.explorer_acceptance_pr
    ss, sn = red.explorer_n_steps
    amr, anr, ssr, snr = ref.explorer_stats()
    assert np.array_equal(an, anr) and np.array_equal(sn, snr) and np.array_equal(ss, ssr)
    np.testing.assert_allclose(am, amr, rtol=rtol, atol=1e-300)
    m, n = red.swap_acceptance_pr
    mr, nr = ref.swap_pr()
    assert np.array_equal(n, nr)
    np.testing.assert_allclose(m, mr, rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=rtol)
    x, chain, rng = pt.replicas.states()
    xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=rtol, atol=1e-9 * rtol / 1e-6)


def test_mala_funnel_parity(P):
    N, d, rounds = 6, 8, 6
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=N, n_rounds=rounds,
                       explorer=P.MALA(step_size=0.2), record=rec, show_report=False))
    ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_MALA, am_step_size=0.2, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, am_preconditioner=2)
    for _ in range(rounds):
        _check_mala_round(P, pt, ref, 1e-6)


def test_compose_rejects_what_the_device_cannot_run(P):
    with pytest.raises((P.PteError, NotImplementedError)):
        P.PT(P.Inputs(target=P.toy_mvn_target(4), n_chains=4, explorer=P.Compose(P.ToyExplorer(), P.SliceSampler()), show_report=False))
    with pytest.raises((P.PteError, NotImplementedError)):
        P.PT(P.Inputs(target=P.toy_mvn_target(4), n_chains=4, explorer=P.Compose(P.AutoMALA(), P.MALA(step_size=0.5)), show_report=False))


def test_sharded_compose_equals_single_engine(P):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1, P.traces]
    mk = lambda: P.Inputs(target=P.toy_mvn_target(20), n_chains=8, n_rounds=4, explorer=P.Compose(P.SliceSampler(), P.AutoMALA()),
                          record=rec, show_report=False)
    one, many = P.PT(mk()), P.PT(mk(), n_shards=4, device_messages=True)
    for _ in range(4):
        assert P.next_round(one) and P.next_round(many)
        ra = P.run_one_round(one); P.adapt(one, ra)
        rb = P.run_one_round(many); P.adapt(many, rb)
        assert np.array_equal(ra.index_process, rb.index_process) and np.array_equal(ra.traces, rb.traces)
        for a, b in zip(ra.energy_ac1 + ra.explorer_acceptance_pr + ra.am_factors, rb.energy_ac1 + rb.explorer_acceptance_pr + rb.am_factors):
            assert np.array_equal(a, b, equal_nan=True)
    xa, ca, ga = one.replicas.states(); xb, cb, gb = many.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)


@pytest.mark.parametrize("explorer", ["slice", "automala", "slice+automala", "mala"])
@pytest.mark.parametrize("transport", ["group", "device_messages", "host"])
def test_parallelism_invariance_matrix_of_the_reference(P, explorer, transport):
    """The reference's own invariance test (test/test_parallelism_invariance.jl:5-43): toy_mvn_target(1), n_chains = 4,
    n_rounds = 10, explorers SliceSampler / AutoMALA / Compose(SliceSampler, AutoMALA), record = [swap_acceptance_pr,
    index_process, log_sum_ratio, round_trip, energy_ac1], two processes against one -- `compare_checkpoints` must find the
    runs identical.  Here: 2 chain shards against one engine through each in-process transport, every recorder and the final
    replicas bit-identical (d = 1: a one-coordinate state row, the smallest the layout allows)."""
    ex = {"slice": P.SliceSampler, "automala": P.AutoMALA, "mala": lambda: P.MALA(step_size=0.3),
          "slice+automala": lambda: P.Compose(P.SliceSampler(), P.AutoMALA())}[explorer]
    rec = [P.swap_acceptance_pr, P.index_process, P.log_sum_ratio, P.round_trip, P.energy_ac1]
    mk = lambda: P.Inputs(target=P.toy_mvn_target(1), n_chains=4, n_rounds=10, explorer=ex(), record=rec, show_report=False)
    kw = {"group": dict(transport="group"), "device_messages": dict(device_messages=True), "host": {}}[transport]
    one, two = P.PT(mk()), P.PT(mk(), n_shards=2, **kw)
    for _ in range(10):
        assert P.next_round(one) and P.next_round(two)
        ra = P.run_one_round(one); P.adapt(one, ra)
        rb = P.run_one_round(two); P.adapt(two, rb)
        assert np.array_equal(ra.index_process, rb.index_process) and ra.round_trip == rb.round_trip
        for a, b in zip(ra.swap_acceptance_pr + ra.log_sum_ratio + ra.energy_ac1 + ra.explorer_acceptance_pr + ra.explorer_n_steps + ra.am_factors,
                        rb.swap_acceptance_pr + rb.log_sum_ratio + rb.energy_ac1 + rb.explorer_acceptance_pr + rb.explorer_n_steps + rb.am_factors):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(one.shared.tempering.schedule.grids, two.shared.tempering.schedule.grids)
    xa, ca, ga = one.replicas.states(); xb, cb, gb = two.shards.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    assert P.stepping_stone_pair(one) == P.stepping_stone_pair(two)


# ---------------------------------------------------------------------------------------------
# checkpoint / resume through pte_get_state / pte_set_state
# (reference src/pt/checkpoint.jl; its own end-to-end check is "a resumed run equals an uninterrupted one",
# test/test_checkpoint.jl, src/pt/checks.jl:52-78).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["slice", "automala", "slice+automala", "funnel-automala", "ising"])
def test_checkpoint_resume_equals_uninterrupted_run(P, tmp_path, name):
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1]
    def mk(n_rounds):
        if name == "funnel-automala":
            return P.Inputs(target=P.Funnel(8), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 8), n_chains=6, n_rounds=n_rounds,
                            explorer=P.AutoMALA(), record=rec, show_report=False, checkpoint=True)
        if name == "ising":
            return P.Inputs(target=P.IsingLogPotential(0.7, 8), n_chains=5, n_rounds=n_rounds, record=rec, show_report=False, checkpoint=True)
        ex = {"slice": P.SliceSampler(), "automala": P.AutoMALA(), "slice+automala": P.Compose(P.SliceSampler(), P.AutoMALA())}[name]
        return P.Inputs(target=P.toy_mvn_target(12), n_chains=6, n_rounds=n_rounds, explorer=ex, record=rec, show_report=False, checkpoint=True)
    import dataclasses
    straight = P.pigeons(P.PT(dataclasses.replace(mk(6), checkpoint=False)))
    folder = str(tmp_path / "exec")
    first = P.pigeons(P.PT(mk(3)), exec_folder=folder)
    assert P.latest_checkpoint_folder(folder) == 3
    assert first.shared.iterators.round == 3
    resumed = P.load_checkpoint(folder, n_rounds_increment=3)          # PT(exec_folder) + increment_n_rounds!
    assert resumed.shared.iterators.round == 3 and resumed.inputs.n_rounds == 6
    resumed = P.pigeons(resumed)
    ra, rb = straight.reduced_recorders, resumed.reduced_recorders
    assert np.array_equal(ra.index_process, rb.index_process) and ra.round_trip == rb.round_trip
    for a, b in zip(ra.swap_acceptance_pr + ra.log_sum_ratio + ra.energy_ac1, rb.swap_acceptance_pr + rb.log_sum_ratio + rb.energy_ac1):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(straight.shared.tempering.schedule.grids, resumed.shared.tempering.schedule.grids)
    xa, ca, ga = straight.replicas.states(); xb, cb, gb = resumed.replicas.states()
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb) and np.array_equal(ga, gb)
    # a live PT can be extended as well (increment_n_rounds!(pt, k))
    more = P.pigeons(P.increment_n_rounds(first, 3))
    assert np.array_equal(more.reduced_recorders.index_process, ra.index_process)


@pytest.mark.parametrize("name", ["slice", "automala", "ising"])
def test_checkpoint_resumed_on_the_oracle(P, tmp_path, name):
    """SURVEY.md 8(f) rank 3 against the ORACLE, not against the engine itself: the checkpoint the engine writes after round 3
    (replica files = what pte_get_state returns; schedule and explorer adaptation from shared) is loaded into a fresh CPU
    oracle, which continues rounds 4..6 -- and must land exactly where the uninterrupted oracle and the resumed engine land."""
    import pickle
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    N = 6
    if name == "ising":
        inputs = lambda n: P.Inputs(target=P.IsingLogPotential(0.7, 8), n_chains=N, n_rounds=n, record=rec, show_report=False, checkpoint=True)
        okw = dict(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=64, p0=0.7, n_chains=N, slice_n_passes=3)
    else:
        ex = lambda: P.SliceSampler() if name == "slice" else P.AutoMALA()
        inputs = lambda n: P.Inputs(target=P.toy_mvn_target(12), n_chains=N, n_rounds=n, explorer=ex(), record=rec, show_report=False, checkpoint=True)
        okw = dict(dim=12, n_chains=N, explorer=O.EXPLORER_SLICE if name == "slice" else O.EXPLORER_AUTOMALA, am_preconditioner=2)
    folder = str(tmp_path / "exec")
    P.pigeons(P.PT(inputs(3)), exec_folder=folder)                      # engine: rounds 1..3, checkpoint after each
    straight = O.OraclePT(**okw)
    for _ in range(6):
        straight.run_round()
    # the oracle resumes from the ENGINE's files
    ck = os.path.join(folder, "round=3", "checkpoint")
    reps = [np.load(os.path.join(ck, "replica=%d.npz" % (i + 1))) for i in range(N)]
    assert [int(r["replica_index"]) for r in reps] == list(range(1, N + 1))
    shared = pickle.load(open(os.path.join(ck, "shared.pkl"), "rb"))
    resumed = O.OraclePT(**okw)
    resumed.set_states(x=np.stack([r["state"] for r in reps]), chain=np.array([int(r["chain"]) - 1 for r in reps]),
                       rng=np.stack([r["rng"] for r in reps]).astype(np.uint64))
    resumed.set_schedule(shared.tempering.schedule.grids)
    if name == "automala":
        resumed.set_explorer_adaptation(shared.explorer.step_size, shared.explorer.estimated_target_std_deviations)
    resumed.set_round(shared.iterators.round)
    for _ in range(3):
        resumed.run_round()
    tol = dict(rtol=1e-9, atol=0)
    assert np.array_equal(resumed.index_process(), straight.index_process()) and resumed.round_trip() == straight.round_trip()
    np.testing.assert_allclose(resumed.schedule(), straight.schedule(), **tol)
    xa, ca, ga = straight.states(); xb, cb, gb = resumed.states()
    assert np.array_equal(ca, cb) and np.array_equal(ga, gb)
    np.testing.assert_allclose(xb, xa, rtol=1e-9, atol=1e-300)
    # ... and the engine resumed from the same folder agrees with both
    dev = P.pigeons(P.load_checkpoint(folder, n_rounds_increment=3))
    assert np.array_equal(dev.reduced_recorders.index_process, straight.index_process())
    xd, cd, gd = dev.replicas.states()
    assert np.array_equal(cd, ca) and np.array_equal(gd, ga)
    np.testing.assert_allclose(xd, xa, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(dev.shared.tempering.schedule.grids, straight.schedule(), **tol)


def test_checkpoint_without_a_folder_gets_the_reference_default(P, tmp_path, monkeypatch):
    """Inputs(checkpoint=True) with no exec folder: results/all/<time stamp> under the working directory, as the reference's
    next_exec_folder does -- not a silent no-op -- and an explicit folder passed to write_checkpoint leaves pt.exec_folder alone."""
    monkeypatch.chdir(tmp_path)
    pt = P.pigeons(target=P.toy_mvn_target(3), n_chains=4, n_rounds=2, show_report=False, checkpoint=True)
    assert pt.exec_folder.startswith(os.path.join("results", "all")) and P.latest_checkpoint_folder(pt.exec_folder) == 2
    assert os.path.realpath(os.path.join("results", "latest")) == os.path.realpath(pt.exec_folder)
    before = pt.exec_folder
    from pigeons_amd.checkpoint import write_checkpoint
    write_checkpoint(pt, str(tmp_path / "elsewhere"))
    assert pt.exec_folder == before and P.latest_checkpoint_folder(str(tmp_path / "elsewhere")) == 2
    again = P.pigeons(P.load_checkpoint(before, n_rounds_increment=1))
    assert again.shared.iterators.round == 3


@pytest.mark.parametrize("impl", [1, 0])
def test_slice_parameters_off_the_defaults(P, impl):
    """w, p, n_passes away from the defaults (small p / w exercise the doubling budget and the exact path).  Since round 4 the default
    kernel has two instantiations: the one for the default parameter range (3 < p <= 20, max_iter >= 9) and the generic one -- both are in
    the list, and a p between them (4: the FAST one with the doubling limit right behind the speculative budget)."""
    for w, p, n_passes in [(0.5, 1, 2), (2.0, 3, 1), (30.0, 20, 2), (0.3, 4, 2), (0.5, 25, 1)]:
        N, d, rounds = 5, 20, 4
        rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online]
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, explorer=P.SliceSampler(w=w, p=p, n_passes=n_passes),
                           record=rec, show_report=False), debug_kernel=impl)
        if impl == 0:
            assert pt.replicas.kernel_name() == ("k_explore_slice8" if 3 < p <= 20 else "k_explore_slice8_generic")
        ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_SLICE, record_online=1, slice_w=w, slice_p=p, slice_n_passes=n_passes)
        for _ in range(rounds):
            _check_round(P, pt, ref)


def test_slice_kernel_many_replicas_equals_sequential_kernel(P):
    """More than 2048 replicas on one GPU (two waves of the 512-draw kernel per SIMD) run the default SliceSampler kernel with the 256-draw window (10 KB of LDS, 16
    resident replicas per CU); same bits as the plain sequential kernel (which the oracle pins at small sizes)."""
    def run(impl):
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(70), n_chains=3000, n_rounds=3, seed=11, explorer=P.SliceSampler(),
                           record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False), debug_kernel=impl)
        assert pt.replicas.kernel_name() == ("k_explore_slice" if impl == 1 else "k_explore_slice8_lds10k")
        out = []
        for _ in range(3):
            P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
            out.append((red.index_process.copy(), red.swap_acceptance_pr[0].copy(), red.explorer_n_steps[0].copy()))
        return out, pt.replicas.states()
    a, sa = run(1)
    b, sb = run(0)
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_slice_special_states(P):
    """States a user can hand in through pte_set_state: all zeros (sum x^2 = 0, the margin of the filtered predicate has
    nothing to scale with), negative zeros, squares that underflow, large magnitudes -- same chain of decisions as the oracle."""
    N, d, rounds = 5, 70, 4
    pt, ref = _mk(P, N, d, rounds, "slice", seed=3)
    x = np.zeros((N, d))
    x[1] = -0.0
    x[2] = 1e-170 * np.arange(1, d + 1)
    x[3] = 1e3 * np.cos(np.arange(d))
    x[4, ::2] = 5e-324                              # denormals between zeros
    pt.replicas.set_states(x=x)
    ref.set_states(x=x)
    for _ in range(rounds):
        _check_round(P, pt, ref)


def test_slice_max_iter_error_is_raised_by_every_kernel(P):
    """slice_shrink!'s "Maximum number of iterations reached" (SliceSampler.jl:179-185) with max_iter below the speculation depth."""
    for impl in [1, 0]:
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(50), n_chains=4, n_rounds=6, explorer=P.SliceSampler(w=1000.0, max_iter=2),
                           record=[P.log_sum_ratio], show_report=False), debug_kernel=impl)
        with pytest.raises(P.PteError, match="Maximum number of iterations"):
            for _ in range(6):
                P.next_round(pt); P.run_one_round(pt)


@pytest.mark.parametrize("n_shards", [1, 5])
def test_extended_traces_parity_and_reference_shape(P, n_shards):
    """inputs.extended_traces (src/pt/pigeons.jl:116): every chain is traced.  Shape assertions of the reference's
    test/test_traces.jl:7-27 (toy_mvn_target(3), 10 chains, 2 rounds: (4, 3 + 1, 10)) plus parity with the oracle."""
    N, d, rounds = 10, 3, 2
    inp = P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=rounds, record=[P.traces, P.index_process], extended_traces=True,
                   show_report=False)
    pt = P.PT(inp, n_shards=n_shards, device_messages=n_shards > 1)
    ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_TOY, record_traces=2)
    for r in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); pt.reduced_recorders = red
        ref.run_round()
        np.testing.assert_allclose(red.traces, ref.traces(), rtol=1e-13, atol=1e-300)
    mtx = P.sample_array(pt)
    assert mtx.shape == (4, d + 1, N)
    assert np.array_equal(P.get_sample(pt, 3, 2), red.traces[1, 2, :])
    assert np.array_equal(P.get_sample(pt, N), red.traces[:, N - 1, :])


# ---------------------------------------------------------------------------------------------
# SURVEY.md 8(f) rank 4 (without GaussianReference): StabilizedPT / VariationalDEO -- two legs sharing the target,
# a reference at both ends (src/tempering/StabilizedPT.jl, src/swap/VariationalDEO.jl, src/swap/OddEven.jl:16-48).
# ---------------------------------------------------------------------------------------------
def test_two_references_double_the_restarts_kat(P):
    """reference test/test_variational.jl:44-57 (test_two_references_2): TestSwapper(0.5), 5 chains, 15 rounds, seed 1;
    with a second leg of 5 chains the tempered restarts double (|2 - ratio| <= 0.05)."""
    kw = dict(target=P.TestSwapper(0.5), record=[P.round_trip], n_chains=5, n_rounds=15, seed=1, show_report=False)
    pt = P.pigeons(**kw)
    pt2 = P.pigeons(n_chains_variational=5, variational=None, **kw)
    ratio = P.n_tempered_restarts(pt2) / P.n_tempered_restarts(pt)
    assert abs(2.0 - ratio) <= 0.05
    ref = O.OraclePT(target=O.TARGET_TEST_SWAPPER, p0=0.5, explorer=O.EXPLORER_NONE, seed=1, n_chains=5, n_chains_variational=5,
                     record_index_process=0)
    for _ in range(15):
        ref.run_round()
    assert pt2.reduced_recorders.round_trip == ref.round_trip()


@pytest.mark.parametrize("explorer,nf,nv,d,rounds", [("slice", 6, 5, 4, 8), ("slice", 4, 4, 70, 5), ("automala", 5, 5, 6, 7), ("toy", 3, 7, 9, 6)])
def test_two_leg_tempering_parity(P, explorer, nf, nv, d, rounds):
    ex = {"slice": P.SliceSampler(), "automala": P.AutoMALA(), "toy": P.ToyExplorer()}[explorer]
    oex = {"slice": O.EXPLORER_SLICE, "automala": O.EXPLORER_AUTOMALA, "toy": O.EXPLORER_TOY}[explorer]
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1, P.traces]
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=nf, n_chains_variational=nv, variational=None, n_rounds=rounds, explorer=ex,
                       record=rec, show_report=False))
    ref = O.OraclePT(n_chains=nf, n_chains_variational=nv, dim=d, explorer=oex, record_online=1, record_energy_ac1=1, record_traces=1,
                     am_preconditioner=2)
    N = nf + nv
    temp = pt.shared.tempering
    # reference test/test_two_legs.jl:29-40 (Issue #290): targets and references are disjoint and live on different legs
    tg = P.target_chains(pt); rf = [i for i in range(1, N + 1) if temp.is_reference(i)]
    assert not set(tg) & set(rf) and len({temp.leg_of(i) for i in tg}) == len(tg) == 2 and len({temp.leg_of(i) for i in rf}) == len(rf) == 2
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); pt.reduced_recorders = red
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        assert red.round_trip == ref.round_trip()
        m, n = red.swap_acceptance_pr; mr, nr = ref.swap_pr()
        assert np.array_equal(n, nr)
        np.testing.assert_allclose(m, mr, rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=RTOL)
        np.testing.assert_allclose(P.global_barrier(pt), ref.global_barrier(), rtol=RTOL)
        np.testing.assert_allclose(P.global_barrier_variational(pt), ref.global_barrier_variational(), rtol=RTOL)
        np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=RTOL)
        om, ov, on = red.online; omr, ovr, onr = ref.online()
        assert on == onr == 2 * 2 ** pt.shared.iterators.round
        np.testing.assert_allclose(om, omr, rtol=1e-9, atol=1e-12); np.testing.assert_allclose(ov, ovr, rtol=1e-9)
        assert red.traces.shape == (2 ** pt.shared.iterators.round, 2, d + 1)
        np.testing.assert_allclose(red.traces, ref.traces(), rtol=1e-9, atol=1e-300)
        cor, cn, mom = red.energy_ac1; corr, cnr, rawr = ref.energy_ac1()
        assert np.array_equal(cn, cnr)
        np.testing.assert_allclose(mom[:, :2], rawr[:, :2], rtol=1e-9)
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-9, atol=1e-15)


# ---------------------------------------------------------------------------------------------
# GaussianReference (src/variational/GaussianReference.jl) on the interpolated funnel path: refitted every round
# from the target chains' online statistics; the variational leg (or the only leg) starts at it.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nf,nv,d,explorer", [(5, 5, 4, "automala"), (6, 0, 6, "automala"), (4, 5, 70, "mala"), (0, 6, 5, "automala")])
def test_gaussian_reference_parity_funnel(P, nf, nv, d, explorer):
    rounds, first = 7, 3
    ex = P.AutoMALA() if explorer == "automala" else P.MALA(step_size=0.2)
    okw = dict(explorer=O.EXPLORER_AUTOMALA) if explorer == "automala" else dict(explorer=O.EXPLORER_MALA, am_step_size=0.2)
    if nf == 0:                                          # variational leg only: one leg (create_tempering, tempering.jl:66-68)
        n_chains, n_var = nv, 0
    else:
        n_chains, n_var = nf, nv
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.energy_ac1]
    pt = P.PT(P.Inputs(target=P.Funnel(d), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, d), n_chains=n_chains,
                       n_chains_variational=n_var, variational=P.GaussianReference(first_tuning_round=first), n_rounds=rounds,
                       explorer=ex, record=rec, show_report=False))
    ref = O.OraclePT(n_chains=n_chains, n_chains_variational=n_var, dim=d, target=O.TARGET_FUNNEL, p0=1.0 / 9.0, am_preconditioner=2,
                     variational_first_tuning_round=first, record_energy_ac1=1, **okw)
    for r in range(1, rounds + 1):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); pt.reduced_recorders = red
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process()), r
        assert red.round_trip == ref.round_trip()
        m, n = red.swap_acceptance_pr; mr, nr = ref.swap_pr()
        assert np.array_equal(n, nr)
        np.testing.assert_allclose(m, mr, rtol=1e-6, atol=1e-300)
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-6)
        np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=1e-6, atol=1e-9)
        v = ref.variational()
        temp = pt.shared.tempering
        leg = temp.variational_leg if n_var > 0 else temp
        if r >= first:
            assert isinstance(leg.path.ref, P.GaussianReference)            # test/test_variational.jl:41
            np.testing.assert_allclose(leg.path.ref.mean, v[0], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(leg.path.ref.standard_deviation, v[1], rtol=1e-6)
        else:
            assert v is None and not isinstance(leg.path, P.InterpolatingPath)
        cor, cn, mom = red.energy_ac1; corr, cnr, rawr = ref.energy_ac1()
        assert np.array_equal(cn, cnr)
        np.testing.assert_allclose(mom[:, :2], rawr[:, :2], rtol=1e-6, atol=1e-9)
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-6, atol=1e-9)


def test_gaussian_reference_survives_a_checkpoint(P, tmp_path):
    mk = lambda n: P.Inputs(target=P.Funnel(5), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 5), n_chains=5, n_chains_variational=4,
                            variational=P.GaussianReference(first_tuning_round=2), n_rounds=n, explorer=P.AutoMALA(),
                            record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False, checkpoint=True)
    import dataclasses
    straight = P.pigeons(P.PT(dataclasses.replace(mk(6), checkpoint=False)))
    folder = str(tmp_path / "exec")
    P.pigeons(P.PT(mk(3)), exec_folder=folder)
    resumed = P.pigeons(P.load_checkpoint(folder, n_rounds_increment=3))
    assert np.array_equal(straight.reduced_recorders.index_process, resumed.reduced_recorders.index_process)
    for a, b in zip(straight.replicas.states(), resumed.replicas.states()):
        assert np.array_equal(a, b)


def test_two_leg_tempering_ising_and_compose(P):
    """Two legs on the other device families: Ising (Bernoulli refresh at both ends) and Compose(SliceSampler, AutoMALA)."""
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    pt = P.PT(P.Inputs(target=P.IsingLogPotential(0.6, 32), n_chains=4, n_chains_variational=5, n_rounds=5, seed=3, record=rec, show_report=False))
    ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=32 * 32, p0=0.6, n_chains=4, n_chains_variational=5, seed=3, slice_n_passes=3)
    pc = P.PT(P.Inputs(target=P.toy_mvn_target(7), n_chains=5, n_chains_variational=4, n_rounds=5, explorer=P.Compose(P.SliceSampler(), P.AutoMALA()),
                       record=rec, show_report=False))
    rc = O.OraclePT(dim=7, n_chains=5, n_chains_variational=4, explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA, am_preconditioner=2)
    for dev, ora in ((pt, ref), (pc, rc)):
        for _ in range(5):
            assert P.next_round(dev)
            red = P.run_one_round(dev); P.adapt(dev, red)
            ora.run_round()
            assert np.array_equal(red.index_process, ora.index_process()) and red.round_trip == ora.round_trip()
            np.testing.assert_allclose(dev.shared.tempering.schedule.grids, ora.schedule(), rtol=RTOL)
            np.testing.assert_allclose(P.stepping_stone_pair(dev), ora.stepping_stone_pair(), rtol=RTOL)
        x, chain, rng = dev.replicas.states(); xr, cr, rr = ora.states()
        assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
        np.testing.assert_allclose(x, xr, rtol=1e-9, atol=1e-15)
