"""The ziggurat tables of Julia's randn / randexp (Random/src/normal.jl `ki, wi, fi, ke, we, fe`) -- the one artefact on this
path that every sampler draw goes through and that no test of the reference pins.

Three independent sources are held against each other, entry by entry (6 x 256):
  * the product's device header pigeons.jl_amd/csrc/zig_tables.h (tools/gen_ziggurat.py: 60-digit mpmath, Python);
  * the CPU oracle's own derivation, built when liboracle loads (oracle/pt_oracle.c: IEEE binary128, libquadmath, C) --
    the oracle never reads the product's header, so a wrong entry in either shows up here, not as a common-mode pass;
  * numpy's embedded tables (numpy/random/lib/libnpyrandom.a: the same 256-layer randmtzig construction, historical
    double-precision values, mantissas one bit wider) -- a third party that neither of the two above was derived from.
And, when tests/golden/reference_pigeons.json exists (tools/gen_golden.jl on a machine with Julia), Julia's literal tables.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
HEADER = os.path.join(ROOT, "pigeons.jl_amd", "csrc", "zig_tables.h")
NAMES = O.ZIG_NAMES


def _header_tables():
    import re
    src = open(HEADER).read()
    out = {}
    for name in NAMES:
        m = re.search(r"ZIG_%s\[256\] = \{(.*?)\};" % name.upper(), src, re.S)
        toks = [t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()]
        if name in ("ki", "ke"):
            out[name] = np.array([int(t.replace("ULL", ""), 16) for t in toks], dtype=np.uint64)
        else:
            out[name] = np.array([float.fromhex(t) for t in toks], dtype=np.float64).view(np.uint64)
        assert out[name].shape == (256,)
    return out


def _first_diff(a, b):
    i = int(np.nonzero(a != b)[0][0])
    return "first difference at index %d: %#x vs %#x" % (i, int(a[i]), int(b[i]))


@pytest.mark.parametrize("name", NAMES)
def test_device_header_equals_the_oracles_independent_derivation(name):
    """mpmath (60 digits, Python) and binary128 (libquadmath, C) agree on every one of the 256 entries, bit for bit."""
    dev, orc = _header_tables()[name], O.zig_table(name, derived=True)
    assert np.array_equal(dev, orc), _first_diff(dev, orc)


def test_recalled_julia_entries():
    """The four entries of Julia's tables recalled in SURVEY.md App. B (the only literal Julia values available here)."""
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_reference.json")))["ziggurat_pins"]
    for src in (_header_tables(), {n: O.zig_table(n, derived=True) for n in NAMES}):
        assert int(src["ki"][0]) == int(pins["ki0"], 16) and int(src["ke"][0]) == int(pins["ke0"], 16)
        assert src["wi"][:1].view(np.float64)[0] == pins["wi0"] and src["we"][:1].view(np.float64)[0] == pins["we0"]


def _numpy_tables():
    """ki/wi/fi/ke/we/fe _double of numpy's ziggurat (static data of distributions.c), read out of libnpyrandom.a."""
    import tempfile
    lib = os.path.join(os.path.dirname(np.__file__), "random", "lib", "libnpyrandom.a")
    if not os.path.exists(lib) or not all(shutil.which(t) for t in ("ar", "nm", "objcopy")):
        pytest.skip("numpy's static random library or binutils not available")
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["ar", "x", lib], cwd=td, check=True)
        objs = [f for f in os.listdir(td) if "distributions" in f and "random_" not in f and "logfactorial" not in f]
        if not objs:
            pytest.skip("distributions object not found in libnpyrandom.a")
        obj = os.path.join(td, objs[0])
        syms = {}
        for ln in subprocess.run(["nm", "-S", obj], capture_output=True, text=True, check=True).stdout.splitlines():
            f = ln.split()
            if len(f) == 4 and f[3].endswith("_double"):
                syms[f[3]] = (int(f[0], 16), int(f[1], 16))
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.rodata", obj, os.path.join(td, "ro.bin")], check=True)
        ro = open(os.path.join(td, "ro.bin"), "rb").read()
    out = {}
    for name in NAMES:
        off, size = syms[name + "_double"]
        assert size == 2048
        out[name] = np.frombuffer(ro[off:off + size], dtype=np.uint64).copy()
    return out


def test_numpy_cross_check_documents_the_last_bit_differences():
    """numpy carries the same construction with 52 / 53-bit mantissas (ki, ke doubled; wi, we halved) as historical
    double-precision values.  Exponential tables: identical to ours except a handful of last-bit entries -- listed here, so
    the test documents exactly which.  Normal tables: numpy used a slightly different section area (ki[0] differs by 6
    counts), so they agree to ~1e-14 relative only.  Neither is Julia's literal table: a 60-digit derivation and a
    double-precision historical table are NOT unique in the last bit, which is why bit-exactness of randn / randexp VALUES
    against Pigeons.jl stays unpinned until tools/gen_golden.jl has dumped Random.ki .. fe."""
    npy, ours = _numpy_tables(), _header_tables()
    f = lambda a: a.view(np.float64)
    # --- exponential: ke_numpy = 2 ke, we_numpy = we / 2, fe_numpy = fe
    ke_diff = np.nonzero(npy["ke"] != 2 * ours["ke"])[0]
    assert ke_diff.tolist() == [2, 10], ke_diff
    assert (npy["ke"][ke_diff].astype(np.int64) - 2 * ours["ke"][ke_diff].astype(np.int64)).tolist() == [-4, -2]
    we_ulps = npy["we"].astype(np.int64) - (f(ours["we"]) / 2).view(np.int64)
    assert np.nonzero(we_ulps)[0].tolist() == [1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 15, 18, 30, 33, 35, 55, 56, 66, 91, 141, 144, 179]
    assert we_ulps.min() == -4 and we_ulps.max() == 0
    fe_ulps = npy["fe"].astype(np.int64) - ours["fe"].astype(np.int64)
    assert np.nonzero(fe_ulps)[0].tolist() == [1, 2, 19, 22, 35, 49, 77, 81, 94, 100, 114, 126, 128]
    assert np.abs(fe_ulps).max() <= 2
    # --- normal: agreement to ~1e-14 relative, thresholds within a few hundred counts of 2^52 (different section area)
    np.testing.assert_allclose(f(npy["wi"]), f(ours["wi"]) / 2, rtol=5e-14)
    np.testing.assert_allclose(f(npy["fi"]), f(ours["fi"]), rtol=5e-14)
    dk = npy["ki"].astype(np.int64) - 2 * ours["ki"].astype(np.int64)
    assert int(dk[0]) == 6 and int(np.abs(dk).max()) < 64
    assert 100 < int((dk != 0).sum()) < 200                     # 154 of 256 at the numpy version of this image


# ------------------------------------------------------------------------------------- tier 3: Julia's literal tables
needs_reference = pytest.mark.skipif(not os.path.exists(O.REFERENCE_FIXTURE),
                                     reason="tests/golden/reference_pigeons.json not generated yet (tools/gen_golden.jl needs Julia): the tables stay unpinned")


@needs_reference
@pytest.mark.parametrize("name", NAMES)
def test_julia_tables_installed_everywhere(name):
    """After `julia tools/gen_golden.jl > tests/golden/reference_pigeons.json && python tools/import_tables.py` the device
    header and the oracle's active tables ARE Julia's; the message names the first entry where the derivation had differed."""
    ref = np.array([int(v) for v in json.load(open(O.REFERENCE_FIXTURE))["tables"][name]], dtype=np.uint64)
    assert np.array_equal(O.zig_table(name), ref), "oracle active table: " + _first_diff(O.zig_table(name), ref)
    dev = _header_tables()[name]
    assert np.array_equal(dev, ref), "device header (run tools/import_tables.py): " + _first_diff(dev, ref)
    der = O.zig_table(name, derived=True)
    if not np.array_equal(der, ref):
        print("note: the binary128 derivation differs from Julia's %s in %d entries; %s" % (name, int((der != ref).sum()), _first_diff(der, ref)))


# ------------------------------------------------------------------------------------- the RNG policy switches (CPU side)
@pytest.fixture
def default_policy():
    yield
    O.set_rng_policy(0)


def _draws(kind, seed, n, policy):
    O.set_rng_policy(policy)
    r = O.OracleRng(seed).split()
    f = {"randn": r.randn, "randexp": r.randexp, "bool": r.rand_bool}[kind]
    out = np.array([f() for _ in range(n)], dtype=np.float64)
    return out, r.state


@pytest.mark.parametrize("kind", ["randn", "randexp"])
def test_tail_policy_changes_only_tail_draws(kind, default_policy):
    """include/pte_rng_policy.h PTE_RNG_TAIL_LOG1P: -log(rand) vs -log1p(-rand) on the idx == 0 tail only."""
    n = 60000
    a, sa = _draws(kind, 7, n, 0)
    b, sb = _draws(kind, 7, n, O.RNG_TAIL_LOG1P)
    R = 3.6541528853610088 if kind == "randn" else 7.69711747013104972
    diff = np.nonzero(a != b)[0]
    assert diff.size >= 3
    assert abs(a[diff[0]]) > R and abs(b[diff[0]]) > R                          # the first difference is a tail draw
    if kind == "randexp":                                                       # one draw per tail visit under both formulas:
        assert sa == sb and diff.size <= 80                                     # same stream position, ~4.5e-4 of the draws differ,
        assert np.all(a[diff] > R) and np.all(b[diff] > R)                      # every one of them beyond R
    # (the normal tail is a rejection loop over pairs of draws: a changed value can change the number of draws consumed)


def test_bool_bit_policy(default_policy):
    a, sa = _draws("bool", 3, 4096, 0)
    raw = O.OracleRng(3).split()
    u = np.array([raw.next_u64() for _ in range(4096)], dtype=np.uint64)
    assert np.array_equal(a, (u & np.uint64(1)).astype(np.float64)) and sa == raw.state
    b, _ = _draws("bool", 3, 4096, O.rng_bool_bit(63))
    assert np.array_equal(b, (u >> np.uint64(63)).astype(np.float64))
    with pytest.raises(ValueError):
        O.set_rng_policy(1 << 20)


# ------------------------------------------------------------------------------------- ... and on the device
@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, O.RNG_TAIL_LOG1P, O.rng_bool_bit(63), O.RNG_TAIL_LOG1P | O.rng_bool_bit(17)])
def test_device_samplers_follow_the_policy(policy, default_policy):
    """Device and oracle read the same policy word (include/pte_rng_policy.h) and produce the same streams under each setting:
    identical draw counts, identical values except <= 1 ulp where libm / ocml log, log1p, exp differ on the slow paths."""
    from pigeons_amd.engine import test_rng_fill, set_rng_policy, get_rng_policy
    try:
        set_rng_policy(policy)
        assert get_rng_policy() == policy
        for kind, name in ((1, "randn"), (2, "randexp"), (3, "bool")):
            n = 60000 if kind != 3 else 5000
            want, st = _draws(name, 7, n, policy)
            r0 = O.OracleRng(7).split()
            got, st_dev = test_rng_fill(np.array(r0.state, dtype=np.uint64), kind, n)
            assert st_dev == st, name
            if kind == 3:
                assert np.array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=4e-16, atol=0)
                assert (got != want).sum() < 30
    finally:
        set_rng_policy(0)


@pytest.mark.gpu
def test_ising_refresh_follows_the_bool_bit_policy(default_policy):
    """The Bernoulli refresh of the Ising reference chain (examples/ising.jl:49-58) under another Bool bit: engine == oracle."""
    import pigeons_amd as P
    from pigeons_amd.engine import set_rng_policy
    pol = O.rng_bool_bit(63)
    try:
        set_rng_policy(pol); O.set_rng_policy(pol)
        for L in (8, 32):
            pt = P.PT(P.Inputs(target=P.IsingLogPotential(0.7, L), n_chains=4, n_rounds=3, show_report=False, seed=2,
                               record=[P.round_trip, P.index_process, P.log_sum_ratio]))
            ref = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=L * L, p0=0.7, n_chains=4, seed=2, slice_n_passes=3)
            for _ in range(3):
                P.next_round(pt); r = P.run_one_round(pt); P.adapt(pt, r); ref.run_round()
                assert np.array_equal(r.index_process, ref.index_process())
            xs, cs, rs = pt.replicas.states(); xr, cr, rr = ref.states()
            assert np.array_equal(xs, xr) and np.array_equal(cs, cr) and np.array_equal(rs, rr)
        # and the default bit gives a different lattice at the reference chain: the switch is live
        set_rng_policy(0); O.set_rng_policy(0)
        ref0 = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=64, p0=0.7, n_chains=4, seed=2, slice_n_passes=3)
        ref0.run_round()
        O.set_rng_policy(pol)
        ref1 = O.OraclePT(target=O.TARGET_ISING, explorer=O.EXPLORER_ISING, dim=64, p0=0.7, n_chains=4, seed=2, slice_n_passes=3)
        ref1.run_round()
        assert not np.array_equal(ref0.states()[0], ref1.states()[0])
    finally:
        set_rng_policy(0); O.set_rng_policy(0)
