"""CPU tests of the host-side mirror (schedule adaptation, barriers, stepping stone, API plumbing)
and of the C-ABI library's exports.  No GPU needed; no compute calls into libpte.so."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def P():
    import __graft_entry__ as g
    g.build_hip()
    import pigeons_amd
    return pigeons_amd


def test_library_exports_every_declared_symbol(P):
    from pigeons_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pte.h")).read()
    declared = set(re.findall(r"\b(pte_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pte_engine", "pte_config"}
    L = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_config_struct_matches_header(P):
    from pigeons_amd import _lib
    L = _lib.load()
    cfg = _lib.PteConfig()
    assert L.pte_default_config(C.byref(cfg)) == 0
    assert cfg.struct_size == C.sizeof(_lib.PteConfig)          # same layout on both sides of the ABI
    assert cfg.abi_version == _lib.ABI_VERSION
    assert (cfg.slice_w, cfg.slice_p, cfg.slice_n_passes, cfg.slice_max_iter) == (10.0, 20, 3, 1024)
    assert (cfg.n_chains, cfg.seed) == (10, 1)
    assert list(cfg.target_params)[:2] == [1.0, 10.0]


def test_enum_mirrors_match_the_header():
    """every enumerator of include/pte.h that the Python side (pigeons_amd/_lib.py) and the Julia glue (julia/PigeonsMI355X.jl) mirror by value"""
    import re
    from pigeons_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pte.h")).read()
    vals = {}
    for name, expr in re.findall(r"^\s*(PTE_[A-Z0-9_]+)\s*=\s*(1u\s*<<\s*\d+|0x[0-9a-fA-F]+|\d+)", hdr, re.M):
        m = re.fullmatch(r"1u\s*<<\s*(\d+)", expr)
        vals[name] = (1 << int(m.group(1))) if m else int(expr, 0)
    pairs = {"PTE_TARGET_MVN_SCALED_PRECISION": _lib.TARGET_MVN_SCALED_PRECISION, "PTE_TARGET_TEST_SWAPPER": _lib.TARGET_TEST_SWAPPER,
             "PTE_TARGET_FUNNEL": _lib.TARGET_FUNNEL, "PTE_TARGET_ISING": _lib.TARGET_ISING,
             "PTE_EXPLORER_NONE": _lib.EXPLORER_NONE, "PTE_EXPLORER_TOY": _lib.EXPLORER_TOY, "PTE_EXPLORER_SLICE": _lib.EXPLORER_SLICE,
             "PTE_EXPLORER_AUTOMALA": _lib.EXPLORER_AUTOMALA, "PTE_EXPLORER_MALA": _lib.EXPLORER_MALA,
             "PTE_RECORD_ROUND_TRIP": _lib.RECORD_ROUND_TRIP, "PTE_RECORD_INDEX_PROCESS": _lib.RECORD_INDEX_PROCESS, "PTE_RECORD_ONLINE": _lib.RECORD_ONLINE,
             "PTE_RECORD_TRACES": _lib.RECORD_TRACES, "PTE_RECORD_ENERGY_AC1": _lib.RECORD_ENERGY_AC1, "PTE_RECORD_TRACES_EXTENDED": _lib.RECORD_TRACES_EXTENDED,
             "PTE_RECORD_REFERENCE_REDUCTION": _lib.RECORD_REFERENCE_REDUCTION,
             "PTE_KERNEL_SLICE_SEQUENTIAL": _lib.KERNEL_SLICE_SEQUENTIAL, "PTE_KERNEL_ISING_BITS": _lib.KERNEL_ISING_BITS, "PTE_KERNEL_ISING_BYTES": _lib.KERNEL_ISING_BYTES,
             "PTE_KERNEL_TWO_LAUNCHES": _lib.KERNEL_TWO_LAUNCHES, "PTE_KERNEL_SCAN_LOOP_ONE_CHAIN": _lib.KERNEL_SCAN_LOOP_ONE_CHAIN, "PTE_KERNEL_FLAG_BITS": _lib.KERNEL_FLAG_BITS}
    for name, py in pairs.items():
        assert name in vals, name
        assert vals[name] == py, (name, vals[name], py)
    assert vals["PTE_KERNEL_FLAG_BITS"] == vals["PTE_KERNEL_TWO_LAUNCHES"] | vals["PTE_KERNEL_SCAN_LOOP_ONE_CHAIN"]
    jl = open(os.path.join(ROOT, "pigeons.jl_amd", "julia", "PigeonsMI355X.jl")).read()
    m = re.search(r"const RECORD_ROUND_TRIP, RECORD_INDEX_PROCESS, RECORD_ONLINE, RECORD_TRACES, RECORD_ENERGY_AC1, RECORD_TRACES_EXTENDED =\s*UInt32\.\(\(([^)]*)\)\)", jl)
    assert [int(x) for x in m.group(1).split(",")] == [vals[k] for k in ("PTE_RECORD_ROUND_TRIP", "PTE_RECORD_INDEX_PROCESS", "PTE_RECORD_ONLINE", "PTE_RECORD_TRACES",
                                                                           "PTE_RECORD_ENERGY_AC1", "PTE_RECORD_TRACES_EXTENDED")]
    assert re.search(r"const RECORD_REFERENCE_REDUCTION = UInt32\((\d+)\)", jl).group(1) == str(vals["PTE_RECORD_REFERENCE_REDUCTION"])


def test_no_cpu_fallback(P):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(P.PteError, match="no HIP device"):
        P.Engine(n_chains=4, dim=8)
    with pytest.raises(P.PteError):
        P.pigeons(target=P.toy_mvn_target(2), show_report=False)


def test_equally_spaced_schedule(P):
    g = P.equally_spaced_schedule(10).grids
    assert g[0] == 0.0 and g[-1] == 1.0 and len(g) == 10
    assert np.array_equal(g, O.OraclePT(n_chains=10).schedule())
    assert list(P.equally_spaced_schedule(1).grids) == [1.0]
    with pytest.raises(AssertionError):
        P.Schedule([0.0, 0.5, 0.5, 1.0])
    with pytest.raises(AssertionError):
        P.Schedule([0.1, 1.0])


@pytest.mark.parametrize("seed", range(6))
def test_fritsch_carlson_matches_oracle(P, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 30))
    x = np.concatenate([[0.0], np.cumsum(rng.random(n - 1) + 1e-3)]); x /= x[-1]
    y = np.concatenate([[0.0], np.cumsum(rng.random(n - 1) * (rng.random(n - 1) > 0.2))])
    f = P.FritschCarlsonMonotonicInterpolation(x, y)
    L = O.lib()
    m = np.zeros(n); c = np.zeros(n); d = np.zeros(n)
    L.po_fc_build(O._dp(x), O._dp(y), n, O._dp(m), O._dp(c), O._dp(d))
    ts = np.concatenate([x, rng.random(50)])
    got = np.array([f(t) for t in ts])
    ref = np.array([L.po_fc_eval(O._dp(x), O._dp(y), O._dp(m), O._dp(c), O._dp(d), n, float(t)) for t in ts])
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-15)
    # interpolates the knots and is monotone
    np.testing.assert_allclose(got[:n], y, rtol=1e-12, atol=1e-14)
    grid = np.linspace(0, 1, 400)
    vals = np.array([f(t) for t in grid])
    assert np.all(np.diff(vals) >= -1e-12)


def test_optimal_schedule_matches_oracle(P):
    """adapt_tempering (host mirror) vs the oracle's C restatement on the same recorders."""
    ref = O.OraclePT(dim=4, n_chains=12, explorer=O.EXPLORER_SLICE)
    old = ref.schedule()
    for _ in range(5):
        ref.run_round()
        m, n = ref.swap_pr()
        rej = P.rejections(m, n)
        new = P.optimal_schedule(rej, old, len(old))
        np.testing.assert_allclose(new, ref.schedule(), rtol=1e-12)
        cb = P.CommunicationBarriers(rej, old)
        assert math.isclose(cb.globalbarrier, ref.global_barrier(), rel_tol=1e-13)
        for b in (0.0, 0.13, 0.5, 0.99, 1.0):
            assert math.isclose(cb.cumulativebarrier(b), ref.cumulative_barrier(b), rel_tol=1e-12, abs_tol=1e-14)
        from pigeons_amd import tempering as T
        up, un, dn, dnn = ref.log_sum_ratio()
        np.testing.assert_allclose(T.stepping_stone_pair(up, un, dn, dnn), ref.stepping_stone_pair(), rtol=1e-13)
        old = new


def test_rejections_default_and_nudge(P):
    r = P.rejections(np.array([0.3, 0.0, 1.0]), np.array([4, 0, 2]))
    assert list(r) == [0.7, 0.5, 0.0]
    # zero intensities collapse knots -> the reference retries once with +1e-6 (adaptation.jl:76-79)
    sched = P.optimal_schedule(np.array([0.5, 0.0, 0.0, 0.5]), np.linspace(0, 1, 5), 5)
    assert sched[0] == 0.0 and sched[-1] == 1.0 and np.all(np.diff(sched) > 0)


def test_stepping_stone_rules(P):
    assert P.stepping_stone.__module__.endswith("pt")
    from pigeons_amd import tempering as T
    assert T.stepping_stone((-math.inf, 2.0)) == 2.0
    assert T.stepping_stone((3.0, math.inf)) == 3.0
    assert T.stepping_stone((1.0, 3.0)) == 2.0


def test_inputs_defaults_match_reference(P):
    i = P.Inputs(target=P.toy_mvn_target(3))
    assert (i.seed, i.n_rounds, i.n_chains, i.multithreaded, i.checkpoint) == (1, 10, 10, False, False)
    assert [b() for b in i.record] == ["log_sum_ratio", "timing_extrema", "allocation_extrema"]
    s = P.SliceSampler()
    assert (s.w, s.p, s.n_passes, s.max_iter) == (10.0, 20, 3, 1024)
    t = P.toy_mvn_target(7)
    assert (t.precision0, t.precision1, t.dim) == (1.0, 10.0, 7)
    assert math.isclose(P.analytic_lognormalization(t), -3.5 * math.log(10.0))


def test_the_product_ignores_pte_lib(monkeypatch):
    """$PTE_LIB used to swap the whole product library silently: the package does not read it any more; a development
    tool opts in through tools/_variant.py -> _lib.use_library(path)."""
    import subprocess, sys                              # a fresh interpreter: reloading _lib here would leave engine.py with the old PteConfig class
    prog = ("import os, sys; sys.path[:0] = [%r, %r]\n"
            "from pigeons_amd import _lib\n"
            "assert _lib.LIB_PATH.endswith(os.path.join('pigeons.jl_amd', 'lib', 'libpte.so')), _lib.LIB_PATH\n"
            "import _variant\n"
            "assert _variant.apply() == '/nonexistent/libpte_other.so' and _lib.LIB_PATH == '/nonexistent/libpte_other.so'\n"
            % (os.path.join(ROOT, "pigeons.jl_amd"), os.path.join(ROOT, "tools")))
    env = dict(os.environ, PTE_LIB="/nonexistent/libpte_other.so")
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
