"""Device-resident replicas: thin object wrapper over the C ABI (include/pte.h).

Plays the role of the reference's `replicas` container (src/replicas/replicas.jl:11-40):
the engine owns every replica's state, chain, rng and recorders in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PteConfig, PteError


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Engine:
    def __init__(self, test_build=False, **kw):
        # test_build: True = libpte_test.so (every kernel generation); a path = that build (e.g. _lib.NRMCUT_LIB_PATH)
        self.L = _lib.load(test_build if isinstance(test_build, str) else (_lib.TEST_LIB_PATH if test_build else None))
        cfg = PteConfig()
        self.L.pte_default_config(C.byref(cfg))
        tp = kw.pop("target_params", None)
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise AttributeError("pte_config has no field %r" % k)
            setattr(cfg, k, v)
        if tp is not None:
            for i, v in enumerate(tp):
                cfg.target_params[i] = v
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.pte_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise PteError(self.L.pte_last_error(None).decode())
        self.h = h
        self.N = int(cfg.n_chains) + int(cfg.n_chains_variational)
        self.d = 0 if cfg.target == _lib.TARGET_TEST_SWAPPER else int(cfg.dim)
        a = np.zeros(3, dtype=np.int64)
        self.L.pte_shard_info(h, _ip(a[0:1]), _ip(a[1:2]), _ip(a[2:3]))
        self.c0, self.K, self.n_pairs = int(a[0]), int(a[1]), int(a[2])

    def close(self):
        if getattr(self, "h", None):
            self.L.pte_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PteError(self.L.pte_last_error(self.h).decode())

    # --- schedule / adaptation
    def set_schedule(self, betas):
        b = np.ascontiguousarray(betas, dtype=np.float64)
        self._chk(self.L.pte_set_schedule(self.h, _dp(b), len(b)))

    def schedule(self):
        b = np.zeros(self.N)
        self._chk(self.L.pte_get_schedule(self.h, _dp(b)))
        return b

    def set_explorer_adaptation(self, step_size, target_std=None):
        if target_std is None:
            self._chk(self.L.pte_set_explorer_adaptation(self.h, step_size, None, 0))
        else:
            s = np.ascontiguousarray(target_std, dtype=np.float64)
            self._chk(self.L.pte_set_explorer_adaptation(self.h, step_size, _dp(s), len(s)))

    def set_variational_reference(self, mean, std, uses):
        m = np.ascontiguousarray(mean, dtype=np.float64); sd = np.ascontiguousarray(std, dtype=np.float64)
        u = np.ascontiguousarray(uses, dtype=np.int32)
        self._chk(self.L.pte_set_variational_reference(self.h, _dp(m), _dp(sd), len(m), u.ctypes.data_as(C.POINTER(C.c_int32))))

    # --- hot path
    def explore(self, scan):
        self._chk(self.L.pte_explore(self.h, scan))

    def swap(self, scan):
        self._chk(self.L.pte_swap(self.h, scan))

    def run_scans(self, first_scan, n_scans):
        self._chk(self.L.pte_run_scans(self.h, first_scan, n_scans))

    def reduce(self):
        self._chk(self.L.pte_reduce(self.h))

    # --- reduced recorders
    def swap_acceptance(self):
        k = max(self.n_pairs, 1)
        m = np.zeros(k); n = np.zeros(k, dtype=np.int64)
        self._chk(self.L.pte_get_swap_acceptance(self.h, _dp(m), _ip(n)))
        return m[:self.n_pairs], n[:self.n_pairs]

    def log_sum_ratio(self):
        k = max(self.n_pairs, 1)
        up = np.zeros(k); dn = np.zeros(k)
        un = np.zeros(k, dtype=np.int64); dnn = np.zeros(k, dtype=np.int64)
        self._chk(self.L.pte_get_log_sum_ratio(self.h, _dp(up), _ip(un), _dp(dn), _ip(dnn)))
        return up[:self.n_pairs], un[:self.n_pairs], dn[:self.n_pairs], dnn[:self.n_pairs]

    def round_trip(self):
        a = np.zeros(1, dtype=np.int64); b = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_get_round_trip(self.h, _ip(a), _ip(b)))
        return int(a[0]), int(b[0])

    def index_process(self):
        n = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_get_index_process(self.h, None, _ip(n)))
        out = np.zeros((self.N, int(n[0])), dtype=np.int64)
        if out.size:
            self._chk(self.L.pte_get_index_process(self.h, _ip(out), _ip(n)))
        return out

    def index_process_shard(self):
        """(replica[scan][K], chain[scan][K]) of the local slots."""
        n = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_get_index_process_shard(self.h, None, None, _ip(n)))
        rep = np.zeros((int(n[0]), self.K), dtype=np.int64); ch = np.zeros((int(n[0]), self.K), dtype=np.int64)
        if rep.size:
            self._chk(self.L.pte_get_index_process_shard(self.h, _ip(rep), _ip(ch), _ip(n)))
        return rep, ch

    def replica_ids(self):
        out = np.zeros(self.K, dtype=np.int64)
        self._chk(self.L.pte_get_replica_ids(self.h, _ip(out)))
        return out

    # --- two-phase swap of chain-sharded engines
    def swap_begin(self, scan):
        stats = np.zeros(4); active = np.zeros(2, dtype=np.int32)
        self._chk(self.L.pte_swap_begin(self.h, scan, _dp(stats), active.ctypes.data_as(C.POINTER(C.c_int32))))
        return stats, active

    def swap_finish(self, scan, nbr_stats):
        nbr = np.ascontiguousarray(nbr_stats, dtype=np.float64)
        acc = np.zeros(2, dtype=np.int32)
        self._chk(self.L.pte_swap_finish(self.h, scan, _dp(nbr), acc.ctypes.data_as(C.POINTER(C.c_int32))))
        return acc

    def payload_words(self):
        return int(self.L.pte_boundary_payload_bytes(self.h)) // 8

    def boundary_export(self, side, ptr, is_device):
        self._chk(self.L.pte_boundary_export(self.h, side, C.c_void_p(ptr), 1 if is_device else 0))

    def boundary_import(self, side, ptr, is_device):
        self._chk(self.L.pte_boundary_import(self.h, side, C.c_void_p(ptr), 1 if is_device else 0))

    # --- device-resident, stream-ordered boundary exchange (include/pte.h: pte_shard_scan_*)
    def stream_ptr(self):
        return int(self.L.pte_get_stream(self.h) or 0)

    def message_words(self):
        return int(self.L.pte_shard_message_bytes(self.h)) // 8

    def shard_set_buffers(self, send_lo, recv_lo, send_hi, recv_hi):
        self._chk(self.L.pte_shard_set_buffers(self.h, C.c_void_p(send_lo), C.c_void_p(recv_lo), C.c_void_p(send_hi), C.c_void_p(recv_hi)))

    def shard_scan_begin(self, scan):
        active = np.zeros(2, dtype=np.int32)
        self._chk(self.L.pte_shard_scan_begin(self.h, scan, active.ctypes.data_as(C.POINTER(C.c_int32))))
        return active

    def shard_scan_finish(self, scan):
        self._chk(self.L.pte_shard_scan_finish(self.h, scan))

    def shard_sync(self):
        n = np.zeros(2, dtype=np.int64)
        self._chk(self.L.pte_shard_sync(self.h, _ip(n)))
        return n

    # --- transport behind the ABI (include/pte.h: pte_comm_*): RCCL send/recv enqueued by the library itself
    def comm_init(self, id_bytes):
        """Collective over the ranks: ncclCommInitRank with the id rank 0 obtained from comm_unique_id()."""
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        self._chk(self.L.pte_comm_init(self.h, buf))

    def comm_destroy(self):
        self._chk(self.L.pte_comm_destroy(self.h))

    def comm_info(self):
        """(kind, n_ranks_seen, boundary_swaps[2])"""
        k = np.zeros(1, dtype=np.int32); n = np.zeros(1, dtype=np.int32); b = np.zeros(2, dtype=np.int64)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.L.pte_comm_info(self.h, k.ctypes.data_as(i32p), n.ctypes.data_as(i32p), _ip(b)))
        return int(k[0]), int(n[0]), b

    def comm_barrier(self):
        self._chk(self.L.pte_comm_barrier(self.h))

    def comm_allreduce(self, values, op="max"):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.L.pte_comm_allreduce(self.h, _dp(v), v.size, 0 if op == "max" else 1))
        return v

    def comm_allgather_bytes(self, payload):
        """All ranks obtain [payload of rank 0, payload of rank 1, ...]; payloads may differ in length."""
        world = int(self.cfg.world_size)
        lens = np.zeros(world)
        lens[int(self.cfg.rank)] = len(payload)
        lens = self.comm_allreduce(lens, op="sum").astype(np.int64)
        m = int(lens.max())
        send = np.zeros(max(m, 1), dtype=np.uint8)
        send[:len(payload)] = np.frombuffer(payload, dtype=np.uint8)
        recv = np.zeros(max(m, 1) * world, dtype=np.uint8)
        self._chk(self.L.pte_comm_allgather(self.h, send.ctypes.data_as(C.c_void_p), max(m, 1), recv.ctypes.data_as(C.c_void_p)))
        return [recv[r * max(m, 1): r * max(m, 1) + int(lens[r])].tobytes() for r in range(world)]

    def kernel_name(self):
        return (self.L.pte_kernel_name(self.h) or b"").decode()

    def scan_loop_name(self):
        """"" = two launches per scan; else the ONE kernel that runs all the scans of a run_scans call (pte_scan_loop_name)"""
        return (self.L.pte_scan_loop_name(self.h) or b"").decode()

    def scan_loop_info(self):
        """(workgroups of the fused kernel the device holds at once, fused launches timed, scans inside them) since the last timing_reset"""
        a = np.zeros(1, dtype=np.int64); b = np.zeros(1, dtype=np.int64); c = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_scan_loop_info(self.h, _ip(a), _ip(b), _ip(c)))
        return int(a[0]), int(b[0]), int(c[0])


    def scan_loop_stats(self):
        """(fused_calls, gate_aborts, poisoned): run_scans calls that ran as ONE launch; launches whose residency gate found a workgroup
        missing (the call then ran as explore + swap launches, same results); whether a failure inside the one-launch loop has poisoned
        the engine (only set_state / close are accepted then) -- pte_scan_loop_stats"""
        a = np.zeros(1, dtype=np.int64); b = np.zeros(1, dtype=np.int64); c = np.zeros(1, dtype=np.int32)
        self._chk(self.L.pte_scan_loop_stats(self.h, _ip(a), _ip(b), c.ctypes.data_as(C.POINTER(C.c_int32))))
        return int(a[0]), int(b[0]), bool(c[0])

    def explorer_stats(self):
        am = np.zeros(self.K); ss = np.zeros(self.K)
        an = np.zeros(self.K, dtype=np.int64); sn = np.zeros(self.K, dtype=np.int64)
        self._chk(self.L.pte_get_explorer_stats(self.h, _dp(am), _ip(an), _dp(ss), _ip(sn)))
        return am, an, ss, sn

    def automala_stats(self):
        fm = np.zeros(self.K); rm = np.zeros(self.K)
        fn = np.zeros(self.K, dtype=np.int64); rn = np.zeros(self.K, dtype=np.int64)
        self._chk(self.L.pte_get_automala_stats(self.h, _dp(fm), _ip(fn), _dp(rm), _ip(rn)))
        return fm, fn, rm, rn

    def online(self):
        m = np.zeros(max(self.d, 1)); v = np.zeros(max(self.d, 1)); n = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_get_online(self.h, _dp(m), _dp(v), _ip(n)))
        return m[:self.d], v[:self.d], int(n[0])

    def online_log_density(self):
        """(mean, variance) of the last entry of the `online` sample [state; log density] (src/pt/state.jl:79)."""
        m = np.zeros(1); v = np.zeros(1)
        self._chk(self.L.pte_get_online_log_density(self.h, _dp(m), _dp(v)))
        return float(m[0]), float(v[0])

    def energy_ac1(self):
        """energy_ac1s of the local chains: (cor[K], n[K], moments[K,5])."""
        cor = np.zeros(self.K); n = np.zeros(self.K, dtype=np.int64); mom = np.zeros((self.K, 5))
        self._chk(self.L.pte_get_energy_ac1(self.h, _dp(cor), _ip(n), _dp(mom)))
        return cor, n, mom

    def traces(self):
        """[scan][d+1] = [state; log density] of the target chain over the last round (empty off the target shard)."""
        n = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_get_traces(self.h, None, _ip(n)))
        ext = bool(int(self.cfg.record_flags) & _lib.RECORD_TRACES_EXTENDED)     # [scan][local chain][d+1]
        two = int(self.cfg.n_chains_variational) > 0                             # [scan][the two target chains][d+1]
        out = np.zeros((int(n[0]), self.K if ext else 2, self.d + 1)) if (ext or two) else np.zeros((int(n[0]), self.d + 1))
        if n[0]:
            self._chk(self.L.pte_get_traces(self.h, _dp(out), _ip(n)))
        return out

    # --- replica fields
    def states(self):
        x = np.zeros((self.K, max(self.d, 1)))
        chain = np.zeros(self.K, dtype=np.int64)
        rng = np.zeros((self.K, 2), dtype=np.uint64)
        self._chk(self.L.pte_get_state(self.h, _dp(x) if self.d > 0 else None, _ip(chain), _up(rng)))
        return x[:, :self.d], chain, rng

    def set_states(self, x=None, chain=None, rng=None):
        xa = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        ca = None if chain is None else np.ascontiguousarray(chain, dtype=np.int64)
        ra = None if rng is None else np.ascontiguousarray(rng, dtype=np.uint64)
        self._chk(self.L.pte_set_state(self.h, None if xa is None else _dp(xa),
                                       None if ca is None else _ip(ca), None if ra is None else _up(ra)))

    # --- measurement
    def timing_reset(self, enable=True):
        """enable: False / True (every kernel) / 2 (explore kernels only)."""
        self._chk(self.L.pte_timing_reset(self.h, 2 if enable == 2 and enable is not True else (1 if enable else 0)))

    def timing(self, kernel):
        ms = np.zeros(1); n = np.zeros(1, dtype=np.int64)
        self._chk(self.L.pte_timing_get(self.h, kernel, _dp(ms), _ip(n)))
        return float(ms[0]), int(n[0])


def _timing_samples(self, kernel):
    """per-launch durations (ms) of kernel 0 = explore / 1 = swap since the last timing_reset"""
    n = np.zeros(1, dtype=np.int64)
    self._chk(self.L.pte_timing_get_samples(self.h, kernel, None, 0, _ip(n)))
    out = np.zeros(max(int(n[0]), 1))
    self._chk(self.L.pte_timing_get_samples(self.h, kernel, _dp(out), out.size, _ip(n)))
    return out[:int(n[0])]


Engine.timing_samples = _timing_samples


def set_rng_policy(policy, device=0, test_build=False):
    """include/pte_rng_policy.h: ziggurat tail formula / rand(rng, Bool) bit; one word per device and per loaded library."""
    L = _lib.load(_lib.TEST_LIB_PATH if test_build else None)
    if L.pte_set_rng_policy(device, policy) != 0:
        raise PteError(L.pte_last_error(None).decode())


def get_rng_policy(device=0):
    L = _lib.load()
    out = C.c_uint32(0)
    if L.pte_get_rng_policy(device, C.byref(out)) != 0:
        raise PteError(L.pte_last_error(None).decode())
    return int(out.value)


def comm_allow_library_override(allow=True):
    """Opt in to $PTE_RCCL_LIB (tests: an RCCL stand-in).  Without it a set variable makes every pte_comm_* call fail.  The flag is per
    loaded library (a static inside each build): it is applied to the default library, to every other build already mapped
    (Engine(test_build=...)) and to those mapped later."""
    _lib.load()
    _lib.comm_allow_library_override(allow)


def comm_library(test_build=False):
    """(file the transport's entry points come from, what its ncclGetVersion reports) -- of the default library, or of the test build"""
    L = _lib.load(_lib.TEST_LIB_PATH if test_build else None)
    buf = C.create_string_buffer(1024)
    ver = C.c_int32(0)
    if L.pte_comm_library(buf, 1024, C.byref(ver)) != 0:
        raise PteError(L.pte_last_error(None).decode())
    return buf.value.decode(), int(ver.value)


def comm_unique_id():
    """128 opaque bytes (ncclGetUniqueId) that rank 0 hands to every rank before Engine.comm_init."""
    L = _lib.load()
    buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    if L.pte_comm_unique_id(buf) != 0:
        raise PteError(L.pte_last_error(None).decode())
    return bytes(buf)


def group_run_scans(engines, first_scan, n_scans):
    """pte_group_run_scans: G shard engines of this process (engines[g] = rank g), messages by device copies."""
    L = engines[0].L
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    if L.pte_group_run_scans(arr, len(engines), first_scan, n_scans) != 0:
        raise PteError(L.pte_last_error(engines[0].h).decode())


def test_rng_fill(seed_gamma, kind, n, device=0):
    """n draws of kind (0 rand, 1 randn, 2 randexp) from stream (seed, gamma), on the device."""
    L = _lib.load()
    sg = np.array(seed_gamma, dtype=np.uint64)
    out = np.zeros(max(n, 1))
    if L.pte_test_rng_fill(device, _up(sg), kind, n, _dp(out)) != 0:
        raise PteError(L.pte_last_error(None).decode())
    return out[:n], (int(sg[0]), int(sg[1]))


def test_sqr_norm(x, device=0):
    L = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(x.shape[0])
    if L.pte_test_sqr_norm(device, _dp(x), x.shape[0], x.shape[1], _dp(out)) != 0:
        raise PteError(L.pte_last_error(None).decode())
    return out
