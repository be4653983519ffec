"""ctypes binding of the CPU oracle (oracle/pt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libpt_oracle.so")

ZIG_NAMES = ("ki", "wi", "fi", "ke", "we", "fe")
TARGET_MVN, TARGET_TEST_SWAPPER, TARGET_FUNNEL, TARGET_ISING = 0, 1, 2, 3
EXPLORER_NONE, EXPLORER_TOY, EXPLORER_SLICE, EXPLORER_AUTOMALA, EXPLORER_ISING, EXPLORER_MALA = 0, 1, 2, 3, 4, 5


class Config(C.Structure):
    _fields_ = [
        ("n_chains", C.c_int64), ("dim", C.c_int64), ("seed", C.c_uint64),
        ("target", C.c_int32), ("explorer", C.c_int32),
        ("p0", C.c_double), ("p1", C.c_double),
        ("slice_w", C.c_double),
        ("slice_p", C.c_int32), ("slice_n_passes", C.c_int32), ("slice_max_iter", C.c_int32),
        ("am_base_n_refresh", C.c_int32),
        ("am_exponent_n_refresh", C.c_double), ("am_step_size", C.c_double),
        ("am_preconditioner", C.c_int32),
        ("am_p0", C.c_double), ("am_p1", C.c_double),
        ("record_round_trip", C.c_int32), ("record_index_process", C.c_int32),
        ("record_online", C.c_int32), ("n_threads", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32),
        ("record_traces", C.c_int32), ("record_energy_ac1", C.c_int32), ("explorer2", C.c_int32),
        ("n_chains_variational", C.c_int64), ("variational_first_tuning_round", C.c_int32),
    ]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("gamma", C.c_uint64)]


class SliceParams(C.Structure):      # SliceSampler's fields, src/explorers/SliceSampler.jl:8-20
    _fields_ = [("w", C.c_double), ("p", C.c_int32), ("n_passes", C.c_int32), ("max_iter", C.c_int32)]


class SliceStats(C.Structure):       # explorer_acceptance_pr (Mean), explorer_n_steps (Sum) of one replica
    _fields_ = [("acc_mean", C.c_double), ("acc_n", C.c_int64), ("steps_sum", C.c_double), ("steps_n", C.c_int64)]


LOGPOTENTIAL_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int64, C.c_void_p)
COORD_FLOAT64, COORD_INTEGER, COORD_BOOL = 0, 1, 2


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")] + [os.path.join(ROOT, "include", "pte_rng_policy.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return LIB_PATH


def build_native():
    """The oracle compiled for THIS machine's cores (-O3 -march=native; still -ffp-contract=off and no fast-math, so the
    results stay bit-identical): bench.py's cpu_baseline leg.  Built where it runs -- a native build must not travel to a
    box with other cores -- and named after the CPU it was built for."""
    import hashlib
    try:
        info = open("/proc/cpuinfo").read()
        key = "".join(ln for ln in info.splitlines(True) if ln.startswith(("model name", "flags")))[:20000]
    except OSError:
        key = "unknown"
    out = os.path.join(ORACLE_DIR, "_build", "libpt_oracle_native_%s.so" % hashlib.sha1(key.encode()).hexdigest()[:10])
    src = os.path.join(ORACLE_DIR, "pt_oracle.c")
    if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                        "-shared", "-o", out, src, "-lquadmath", "-lm"], check=True, capture_output=True, cwd=ORACLE_DIR)
    return out


_libs = {}


def lib(path=None):
    if path is None:
        build()
        path = LIB_PATH
    if path in _libs:
        return _libs[path]
    L = C.CDLL(path)
    dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    L.po_rng_new.restype = Rng
    L.po_rng_new.argtypes = [C.c_uint64]
    L.po_rng_next_u64.restype = C.c_uint64
    L.po_rng_next_u64.argtypes = [C.POINTER(Rng)]
    L.po_rng_split.restype = Rng
    L.po_rng_split.argtypes = [C.POINTER(Rng)]
    for f in ("po_rand", "po_randn", "po_randexp"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.POINTER(Rng)]
    L.po_rand_bool_pub.restype = C.c_int
    L.po_rand_bool_pub.argtypes = [C.POINTER(Rng)]
    L.po_zig_table.restype = None
    L.po_zig_table.argtypes = [C.c_int, C.c_void_p]
    L.po_zig_install.restype = None
    L.po_zig_install.argtypes = [C.c_int, C.c_void_p]
    L.po_set_rng_policy.restype = C.c_int
    L.po_set_rng_policy.argtypes = [C.c_uint32]
    L.po_get_rng_policy.restype = C.c_uint32
    L.po_sqr_norm.restype = C.c_double
    L.po_sqr_norm.argtypes = [dp, C.c_int64]
    L.po_logaddexp.restype = C.c_double
    L.po_logaddexp.argtypes = [C.c_double, C.c_double]
    L.po_fc_build.argtypes = [dp, dp, C.c_int64, dp, dp, dp]
    L.po_fc_eval.restype = C.c_double
    L.po_fc_eval.argtypes = [dp, dp, dp, dp, dp, C.c_int64, C.c_double]
    L.po_default_config.argtypes = [C.POINTER(Config)]
    L.po_create.restype = C.c_void_p
    L.po_create.argtypes = [C.POINTER(Config)]
    L.po_destroy.argtypes = [C.c_void_p]
    L.po_last_error.restype = C.c_char_p
    L.po_last_error.argtypes = [C.c_void_p]
    for f in ("po_run_round", "po_begin_round", "po_end_round"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p]
    L.po_run_scans.restype = C.c_int
    L.po_run_scans.argtypes = [C.c_void_p, C.c_int64]
    L.po_round.restype = C.c_int64
    L.po_round.argtypes = [C.c_void_p]
    L.po_set_round.restype = None
    L.po_set_round.argtypes = [C.c_void_p, C.c_int64]
    L.po_get_states.argtypes = [C.c_void_p, dp, ip, up]
    L.po_get_schedule.argtypes = [C.c_void_p, dp]
    L.po_set_schedule.argtypes = [C.c_void_p, dp]
    L.po_get_swap_pr.argtypes = [C.c_void_p, dp, ip]
    L.po_get_log_sum_ratio.argtypes = [C.c_void_p, dp, ip, dp, ip]
    L.po_get_round_trip.argtypes = [C.c_void_p, ip, ip]
    L.po_get_index_process.restype = C.c_int64
    L.po_get_index_process.argtypes = [C.c_void_p, ip]
    L.po_get_explorer_stats.argtypes = [C.c_void_p, dp, ip, dp, ip]
    L.po_get_am_stats.argtypes = [C.c_void_p, dp, ip, dp, ip]
    L.po_get_online.restype = C.c_int64
    L.po_get_online.argtypes = [C.c_void_p, dp, dp]
    L.po_get_online_lp.restype = None
    L.po_get_online_lp.argtypes = [C.c_void_p, dp, dp, ip]
    L.po_get_energy_ac1.restype = None
    L.po_get_energy_ac1.argtypes = [C.c_void_p, dp, ip, dp]
    L.po_get_traces.restype = C.c_int64
    L.po_get_traces.argtypes = [C.c_void_p, dp]
    L.po_get_stepping_stone.argtypes = [C.c_void_p, dp]
    L.po_get_global_barrier.restype = C.c_double
    L.po_get_global_barrier.argtypes = [C.c_void_p]
    L.po_cumulative_barrier.restype = C.c_double
    L.po_cumulative_barrier.argtypes = [C.c_void_p, C.c_double]
    L.po_get_step_size.restype = C.c_double
    L.po_get_step_size.argtypes = [C.c_void_p]
    L.po_get_target_std.restype = C.c_int64
    L.po_get_target_std.argtypes = [C.c_void_p, dp]
    L.po_set_explorer_adaptation.argtypes = [C.c_void_p, C.c_double, dp]
    i32p = C.POINTER(C.c_int32)
    L.po_shard_explore.restype = C.c_int
    L.po_shard_explore.argtypes = [C.c_void_p, C.c_int64]
    L.po_shard_swap_begin.restype = C.c_int
    L.po_shard_swap_begin.argtypes = [C.c_void_p, C.c_int64, dp, i32p]
    L.po_shard_swap_finish.restype = C.c_int
    L.po_shard_swap_finish.argtypes = [C.c_void_p, C.c_int64, dp, i32p]
    L.po_shard_payload_words.restype = C.c_int64
    L.po_shard_payload_words.argtypes = [C.c_void_p]
    L.po_shard_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.po_shard_import.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.po_shard_reduce.restype = C.c_int
    L.po_shard_reduce.argtypes = [C.c_void_p]
    L.po_shard_info.argtypes = [C.c_void_p, ip, ip, ip]
    L.po_shard_replica_ids.argtypes = [C.c_void_p, ip]
    L.po_shard_index_process.restype = C.c_int64
    L.po_shard_index_process.argtypes = [C.c_void_p, ip, ip]
    L.po_rand_range.restype = C.c_int64
    L.po_rand_range.argtypes = [C.POINTER(Rng), C.c_int64, C.c_int64]
    L.po_slice_step_mixed.restype = C.c_int
    L.po_slice_step_mixed.argtypes = [C.POINTER(Rng), dp, i32p, C.c_int64, C.POINTER(SliceParams), LOGPOTENTIAL_FN, C.c_void_p,
                                      C.POINTER(SliceStats), C.c_char_p, C.c_size_t]
    _libs[path] = L
    _install_reference_tables(L)
    return L


REFERENCE_FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_pigeons.json")


def _install_reference_tables(L):
    """Once tools/gen_golden.jl has produced the live-reference fixture, the oracle samples with Julia's OWN ziggurat tables
    (the product gets them through tools/import_tables.py); its binary128 derivation stays readable (zig_table(derived=True))."""
    if not os.path.exists(REFERENCE_FIXTURE):
        return False
    import json
    t = json.load(open(REFERENCE_FIXTURE)).get("tables")
    if not t:
        return False
    for i, name in enumerate(ZIG_NAMES):
        a = np.array([int(v) for v in t[name]], dtype=np.uint64)
        assert a.shape == (256,)
        L.po_zig_install(i, a.ctypes.data)
    return True


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


RNG_TAIL_LOG1P = 1


def rng_bool_bit(k):
    return (k & 63) << 8


def zig_table(name, derived=False):
    """256 uint64 bit patterns: the table the oracle samples with, or (derived=True) its own binary128 derivation --
    the same thing until the live-reference fixture installs Julia's tables."""
    out = np.zeros(256, dtype=np.uint64)
    lib().po_zig_table(ZIG_NAMES.index(name) + (8 if derived else 0), out.ctypes.data)
    return out


def set_libm_nudge(mode):
    """tests only: exp / log of the Langevin family as libm returns them (0), one ulp up (1), down (2), alternating per call (3)"""
    L = lib()
    L.po_set_libm_nudge.restype = C.c_int
    L.po_set_libm_nudge.argtypes = [C.c_int]
    if L.po_set_libm_nudge(int(mode)) != 0:
        raise ValueError("invalid libm nudge %r" % (mode,))


def set_rng_policy(policy):
    if lib().po_set_rng_policy(policy) != 0:
        raise ValueError("invalid rng policy %r" % (policy,))


class OracleRng:
    """SplittableRandom + Julia samplers (for RNG parity tests)."""

    def __init__(self, seed=None, state=None):
        self.L = lib()
        self.r = self.L.po_rng_new(seed) if state is None else Rng(*state)

    def next_u64(self):
        return self.L.po_rng_next_u64(C.byref(self.r))

    def split(self):
        c = self.L.po_rng_split(C.byref(self.r))
        return OracleRng(state=(c.seed, c.gamma))

    def rand(self):
        return self.L.po_rand(C.byref(self.r))

    def randn(self):
        return self.L.po_randn(C.byref(self.r))

    def randexp(self):
        return self.L.po_randexp(C.byref(self.r))

    def rand_bool(self):
        return self.L.po_rand_bool_pub(C.byref(self.r))

    def rand_range(self, a, b):
        """rand(rng, a:b) on Int64 (Random.SamplerRangeNDL)"""
        return self.L.po_rand_range(C.byref(self.r), a, b)

    @property
    def state(self):
        return (self.r.seed, self.r.gamma)


class MixedSliceSampler:
    """step!(::SliceSampler) on a state whose coordinates are Float64 / Integer / Bool (kinds: COORD_* per coordinate), behind a Python
    log potential  lp(state: np.ndarray) -> float.  Counts the density evaluations."""

    def __init__(self, log_potential, kinds, w=10.0, p=20, n_passes=3, max_iter=1024):
        self.L = lib()
        self.kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        self.h = SliceParams(w, p, n_passes, max_iter)
        self.stats = SliceStats(0.0, 0, 0.0, 0)
        self.n_evals = 0
        d = len(self.kinds)

        def cb(ptr, dd, _ctx):
            self.n_evals += 1
            return float(log_potential(np.ctypeslib.as_array(ptr, shape=(d,))))
        self._cb = LOGPOTENTIAL_FN(cb)

    def step(self, rng, state):
        """one exploration step in place (state: float64 array); raises RuntimeError with the oracle's message"""
        assert state.dtype == np.float64 and state.flags.c_contiguous and len(state) == len(self.kinds)
        err = C.create_string_buffer(256)
        rc = self.L.po_slice_step_mixed(C.byref(rng.r), _dp(state), self.kinds.ctypes.data_as(C.POINTER(C.c_int32)), len(state),
                                        C.byref(self.h), self._cb, None, C.byref(self.stats), err, 256)
        if rc:
            raise RuntimeError(err.value.decode())
        return state


class OraclePT:
    """The reference's `pigeons()` loop restated on the CPU."""

    def __init__(self, lib_path=None, **kw):
        self.L = lib(lib_path)
        self.cfg = Config()
        self.L.po_default_config(C.byref(self.cfg))
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        self.h = self.L.po_create(C.byref(self.cfg))
        self.N = int(self.cfg.n_chains) + max(int(self.cfg.n_chains_variational), 0)
        self.d = 0 if self.cfg.target == TARGET_TEST_SWAPPER else int(self.cfg.dim)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.po_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.po_last_error(self.h).decode())

    def run_round(self):
        self._chk(self.L.po_run_round(self.h))

    def begin_round(self):
        self._chk(self.L.po_begin_round(self.h))

    def run_scans(self, n):
        self._chk(self.L.po_run_scans(self.h, n))

    def end_round(self):
        self._chk(self.L.po_end_round(self.h))

    @property
    def round(self):
        return int(self.L.po_round(self.h))

    def set_round(self, r):
        """resume: shared.iterators.round of a checkpoint (src/pt/checkpoint.jl:19-54)"""
        self.L.po_set_round(self.h, r)

    def set_states(self, x=None, chain=None, rng=None):
        xa = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        ca = None if chain is None else np.ascontiguousarray(chain, dtype=np.int64)
        ra = None if rng is None else np.ascontiguousarray(rng, dtype=np.uint64)
        self.L.po_set_states.restype = None
        self.L.po_set_states.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.po_set_states(self.h, None if xa is None else xa.ctypes.data, None if ca is None else ca.ctypes.data,
                             None if ra is None else ra.ctypes.data)

    def states(self):
        x = np.zeros((self.N, max(self.d, 1)))
        chain = np.zeros(self.N, dtype=np.int64)
        rng = np.zeros((self.N, 2), dtype=np.uint64)
        self.L.po_get_states(self.h, _dp(x), _ip(chain), _up(rng))
        return x[:, :self.d], chain, rng

    def schedule(self):
        b = np.zeros(self.N)
        self.L.po_get_schedule(self.h, _dp(b))
        return b

    def set_schedule(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        self.L.po_set_schedule(self.h, _dp(b))

    def swap_pr(self):
        m = np.zeros(max(self.N - 1, 1)); n = np.zeros(max(self.N - 1, 1), dtype=np.int64)
        self.L.po_get_swap_pr(self.h, _dp(m), _ip(n))
        return m[:self.N - 1], n[:self.N - 1]

    def log_sum_ratio(self):
        k = max(self.N - 1, 1)
        up = np.zeros(k); dn = np.zeros(k)
        un = np.zeros(k, dtype=np.int64); dnn = np.zeros(k, dtype=np.int64)
        self.L.po_get_log_sum_ratio(self.h, _dp(up), _ip(un), _dp(dn), _ip(dnn))
        return up[:self.N - 1], un[:self.N - 1], dn[:self.N - 1], dnn[:self.N - 1]

    def round_trip(self):
        a = np.zeros(1, dtype=np.int64); b = np.zeros(1, dtype=np.int64)
        self.L.po_get_round_trip(self.h, _ip(a), _ip(b))
        return int(a[0]), int(b[0])   # (n_tempered_restarts, n_round_trips)

    def index_process(self):
        n = int(self.L.po_get_index_process(self.h, None))
        out = np.zeros((self.N, n), dtype=np.int64)
        self.L.po_get_index_process(self.h, _ip(out))
        return out

    def explorer_stats(self):
        am = np.zeros(self.N); sn = np.zeros(self.N)
        an = np.zeros(self.N, dtype=np.int64); ssn = np.zeros(self.N, dtype=np.int64)
        self.L.po_get_explorer_stats(self.h, _dp(am), _ip(an), _dp(sn), _ip(ssn))
        return am, an, sn, ssn

    def am_stats(self):
        fm = np.zeros(self.N); rm = np.zeros(self.N)
        fn = np.zeros(self.N, dtype=np.int64); rn = np.zeros(self.N, dtype=np.int64)
        self.L.po_get_am_stats(self.h, _dp(fm), _ip(fn), _dp(rm), _ip(rn))
        return fm, fn, rm, rn

    def online(self):
        m = np.zeros(max(self.d, 1)); v = np.zeros(max(self.d, 1))
        n = self.L.po_get_online(self.h, _dp(m), _dp(v))
        return m[:self.d], v[:self.d], int(n)

    def online_lp(self):
        m = np.zeros(1); v = np.zeros(1); n = np.zeros(1, dtype=np.int64)
        self.L.po_get_online_lp(self.h, _dp(m), _dp(v), _ip(n))
        return float(m[0]), float(v[0]), int(n[0])

    def energy_ac1(self):
        """(cor[N], n[N], raw[N,5] = b0, b1, A00, A01, A11)"""
        cor = np.zeros(self.N); n = np.zeros(self.N, dtype=np.int64); raw = np.zeros((self.N, 5))
        self.L.po_get_energy_ac1(self.h, _dp(cor), _ip(n), _dp(raw))
        return cor, n, raw

    def traces(self):
        """[scan][d+1] of the target chain, or [scan][chain][d+1] with record_traces == 2 (extended_traces)."""
        n = int(self.L.po_get_traces(self.h, None))
        ext = int(self.cfg.record_traces) == 2
        two = int(self.cfg.n_chains_variational) > 0          # [scan][the two target chains][d+1]
        out = np.zeros((n, self.N if ext else 2, self.d + 1)) if (ext or two) else np.zeros((n, self.d + 1))
        if n:
            self.L.po_get_traces(self.h, _dp(out))
        return out

    def stepping_stone_pair(self):
        p = np.zeros(2)
        self.L.po_get_stepping_stone(self.h, _dp(p))
        return float(p[0]), float(p[1])

    def global_barrier(self):
        return float(self.L.po_get_global_barrier(self.h))

    def variational(self):
        """(mean, std) of the active GaussianReference, or None"""
        m = np.zeros(max(self.d, 1)); sd = np.zeros(max(self.d, 1))
        self.L.po_get_variational.restype = C.c_int
        self.L.po_get_variational.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        return (m[:self.d], sd[:self.d]) if self.L.po_get_variational(self.h, _dp(m), _dp(sd)) else None

    def global_barrier_variational(self):
        self.L.po_get_global_barrier_variational.restype = C.c_double
        self.L.po_get_global_barrier_variational.argtypes = [C.c_void_p]
        return float(self.L.po_get_global_barrier_variational(self.h))

    def cumulative_barrier(self, beta):
        return float(self.L.po_cumulative_barrier(self.h, beta))

    def step_size(self):
        return float(self.L.po_get_step_size(self.h))

    def target_std(self):
        out = np.zeros(max(self.d, 1))
        n = self.L.po_get_target_std(self.h, _dp(out))
        return out[:self.d] if n else None

    def set_explorer_adaptation(self, step_size, target_std=None):
        s = None if target_std is None else np.ascontiguousarray(target_std, dtype=np.float64)
        self.L.po_set_explorer_adaptation(self.h, step_size, None if s is None else _dp(s))

    def automala_stats(self):
        sl = slice(getattr(self, "c0", 0), getattr(self, "c0", 0) + getattr(self, "K", self.N))
        return tuple(a[sl] for a in self.am_stats())


class OracleShard(OraclePT):
    """Oracle-backed chain shard with the Engine methods the sharded drivers use (CPU tests of the
    multi-rank protocol; pigeons_amd.sharded.LoopbackShards / DistShard)."""

    def __init__(self, rank=0, world_size=1, **kw):
        kw = dict(kw)
        # accept the pte_config spelling used by pigeons_amd.PT
        tp = kw.pop("target_params", None)
        if tp is not None:
            kw["p0"], kw["p1"] = tp[0], (tp[1] if len(tp) > 1 else 0.0)
        flags = kw.pop("record_flags", None)
        if flags is not None:
            kw["record_round_trip"] = 1 if flags & 1 else 0
            kw["record_index_process"] = 1 if flags & 2 else 0
            kw["record_online"] = 1 if flags & 4 else 0
            kw["record_traces"] = (2 if flags & 32 else 1) if flags & 8 else 0
            kw["record_energy_ac1"] = 1 if flags & 16 else 0
        kw.pop("device", None); kw.pop("max_scans_per_round", None)
        super().__init__(rank=rank, world_size=world_size, **kw)
        a = np.zeros(3, dtype=np.int64)
        self.L.po_shard_info(self.h, _ip(a[0:1]), _ip(a[1:2]), _ip(a[2:3]))
        self.c0, self.K, self.n_pairs = int(a[0]), int(a[1]), int(a[2])

    def explore(self, scan):
        self._chk(self.L.po_shard_explore(self.h, scan))

    def swap_begin(self, scan):
        stats = np.zeros(4); active = np.zeros(2, dtype=np.int32)
        self._chk(self.L.po_shard_swap_begin(self.h, scan, _dp(stats), active.ctypes.data_as(C.POINTER(C.c_int32))))
        return stats, active

    def swap_finish(self, scan, nbr):
        nbr = np.ascontiguousarray(nbr, dtype=np.float64); acc = np.zeros(2, dtype=np.int32)
        self._chk(self.L.po_shard_swap_finish(self.h, scan, _dp(nbr), acc.ctypes.data_as(C.POINTER(C.c_int32))))
        return acc

    def payload_words(self):
        return int(self.L.po_shard_payload_words(self.h))

    def boundary_export(self, side, ptr, is_device):
        assert not is_device
        self.L.po_shard_export(self.h, side, C.c_void_p(ptr))

    def boundary_import(self, side, ptr, is_device):
        assert not is_device
        self.L.po_shard_import(self.h, side, C.c_void_p(ptr))

    def reduce(self):
        self._chk(self.L.po_shard_reduce(self.h))

    def swap_acceptance(self):
        m, n = self.swap_pr()
        return m[self.c0:self.c0 + self.n_pairs], n[self.c0:self.c0 + self.n_pairs]

    def log_sum_ratio(self):
        sl = slice(self.c0, self.c0 + self.n_pairs)
        return tuple(a[sl] for a in super().log_sum_ratio())

    def explorer_stats(self):
        sl = slice(self.c0, self.c0 + self.K)
        return tuple(a[sl] for a in super().explorer_stats())

    def energy_ac1(self):
        sl = slice(self.c0, self.c0 + self.K)
        return tuple(a[sl] for a in super().energy_ac1())

    def online_log_density(self):
        return self.online_lp()[:2]

    def traces(self):
        t = super().traces()
        return t[:, self.c0:self.c0 + self.K, :] if t.ndim == 3 else t

    def index_process_shard(self):
        n = int(self.L.po_shard_index_process(self.h, None, None)) if False else None
        # sizes: query via a first call that does not reset (replica=None keeps the buffer)
        rep = np.zeros((4096, self.K), dtype=np.int64); ch = np.zeros((4096, self.K), dtype=np.int64)
        n = int(self.L.po_shard_index_process(self.h, _ip(rep), _ip(ch)))
        return rep[:n].copy(), ch[:n].copy()

    def replica_ids(self):
        out = np.zeros(self.K, dtype=np.int64)
        self.L.po_shard_replica_ids(self.h, _ip(out))
        return out

    def states(self):
        x = np.zeros((self.K, max(self.d, 1))); chain = np.zeros(self.K, dtype=np.int64)
        rng = np.zeros((self.K, 2), dtype=np.uint64)
        self.L.po_get_states(self.h, _dp(x), _ip(chain), _up(rng))
        return x[:, :self.d], chain, rng
