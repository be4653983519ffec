"""ctypes binding of libpte.so (the C ABI in include/pte.h).

The product path has NO CPU fallback: if the HIP library is missing or no HIP
device is present, every engine call fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # .../pigeons.jl_amd
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libpte.so")   # the product never reads the environment for this (round 5: $PTE_LIB used to swap the whole
                                                         # library silently); a development tool selects a tuning build with use_library(path)


def use_library(path):
    """tools/ only (tools/_variant.py: PTE_LIB=<path> python tools/<tool>.py): load this build of libpte instead of the in-tree product
    library, from the next load() on.  Call it before the first engine is made."""
    global LIB_PATH
    LIB_PATH = path

TARGET_MVN_SCALED_PRECISION, TARGET_TEST_SWAPPER, TARGET_FUNNEL, TARGET_ISING = 0, 1, 2, 3
EXPLORER_NONE, EXPLORER_TOY, EXPLORER_SLICE, EXPLORER_AUTOMALA, EXPLORER_ISING_METROPOLIS, EXPLORER_MALA = 0, 1, 2, 3, 4, 5
RECORD_ROUND_TRIP, RECORD_INDEX_PROCESS, RECORD_ONLINE, RECORD_TRACES, RECORD_ENERGY_AC1, RECORD_TRACES_EXTENDED = 1, 2, 4, 8, 16, 32
RECORD_REFERENCE_REDUCTION = 64      # swap_acceptance_pr / log_sum_ratio by per-replica Mean / LogSum fits and the binary-tree merge, replayed in pte_reduce (include/pte.h)
ABI_VERSION = 2
KERNEL_DEFAULT, KERNEL_SLICE_SEQUENTIAL, KERNEL_ISING_BITS, KERNEL_ISING_BYTES = 0, 1, 101, 102
KERNEL_SCAN_LOOP_ONE_CHAIN = 0x2000     # flag: the one-kernel scan loop with ONE chain per workgroup even where the form with several (LDS hand-shakes) exists
KERNEL_FLAG_BITS = 0x3000
KERNEL_TEST_DEAD_CHAIN, KERNEL_TEST_LATE_WORKGROUP, KERNEL_TEST_BITS = 0x4000, 0x8000, 0x1C000    # fault injection into the one-kernel scan loop: libpte_test.so only
KERNEL_TEST_LANGEVIN_ONE_WAVE = 0x10000   # libpte_test.so only: 512 < d <= 1024 on the one-wave Langevin kernel (A/B reference of k_explore_langevin_mw)
KERNEL_TWO_LAUNCHES = 0x1000            # flag: explore + swap launched per scan even where pte_run_scans could be one kernel (pte_scan_loop_name)
COMM_ID_BYTES = 128
RNG_TAIL_LOG1P = 1                      # include/pte_rng_policy.h


def rng_bool_bit(k):
    return (k & 63) << 8
TEST_LIB_PATH = os.path.join(PKG_ROOT, "lib", "libpte_test.so")    # -DPTE_TEST_KERNELS build: every kernel generation (parity tests, bisecting)
NRMCUT_LIB_PATH = os.path.join(PKG_ROOT, "lib", "libpte_nrmcut.so")   # -DNRM_MAX_EV_=2 build: the normal generator's cut-short / fallback paths run on most chunks


class PteConfig(C.Structure):
    """Mirror of `pte_config` (include/pte.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("device", C.c_int32), ("target", C.c_int32), ("explorer", C.c_int32),
        ("record_flags", C.c_uint32),
        ("n_chains", C.c_int64), ("dim", C.c_int64), ("seed", C.c_uint64),
        ("max_scans_per_round", C.c_int64),
        ("target_params", C.c_double * 4),
        ("slice_w", C.c_double),
        ("slice_p", C.c_int32), ("slice_n_passes", C.c_int32), ("slice_max_iter", C.c_int32),
        ("am_base_n_refresh", C.c_int32),
        ("am_exponent_n_refresh", C.c_double), ("am_step_size", C.c_double),
        ("am_p0", C.c_double), ("am_p1", C.c_double),
        ("am_preconditioner", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("explorer2", C.c_int32), ("n_chains_variational", C.c_int64),
        ("debug_kernel", C.c_int32), ("reserved0", C.c_int32),
    ]


class PteError(RuntimeError):
    pass


EXPORTS = [
    "pte_default_config", "pte_create", "pte_destroy", "pte_last_error",
    "pte_set_schedule", "pte_get_schedule", "pte_set_explorer_adaptation",
    "pte_explore", "pte_swap", "pte_run_scans", "pte_reduce",
    "pte_get_swap_acceptance", "pte_get_log_sum_ratio", "pte_get_round_trip",
    "pte_get_index_process", "pte_get_explorer_stats", "pte_get_automala_stats",
    "pte_get_online", "pte_get_state", "pte_set_state",
    "pte_timing_reset", "pte_timing_get", "pte_timing_get_samples", "pte_test_rng_fill", "pte_test_sqr_norm", "pte_test_quotient",
    "pte_shard_info", "pte_swap_begin", "pte_swap_finish", "pte_boundary_payload_bytes",
    "pte_boundary_export", "pte_boundary_import", "pte_get_index_process_shard", "pte_get_replica_ids",
    "pte_get_stream", "pte_shard_message_bytes", "pte_shard_set_buffers", "pte_shard_scan_begin",
    "pte_shard_scan_finish", "pte_shard_sync",
    "pte_get_online_log_density", "pte_get_energy_ac1", "pte_get_traces", "pte_set_variational_reference",
    "pte_comm_allow_library_override", "pte_comm_library", "pte_comm_unique_id", "pte_comm_init", "pte_comm_destroy", "pte_comm_info", "pte_comm_barrier",
    "pte_comm_allreduce", "pte_comm_allgather", "pte_group_run_scans", "pte_kernel_name",
    "pte_set_rng_policy", "pte_get_rng_policy", "pte_scan_loop_name", "pte_scan_loop_info", "pte_scan_loop_stats",
]

_libs = {}
_comm_override_allowed = False      # the opt-in to $PTE_RCCL_LIB lives in a static of EACH loaded library: applied to all of them, now and at load


def comm_allow_library_override(allow=True):
    """pte_comm_allow_library_override on every libpte build this process has mapped (libpte.so, the test builds, a $PTE_LIB build) and on
    every one it maps later: the flag is a function-local static of the library, so each build has its own copy."""
    global _comm_override_allowed
    _comm_override_allowed = bool(allow)
    for L in _libs.values():
        L.pte_comm_allow_library_override(1 if allow else 0)


def load(path=None):
    """Load libpte.so (or the build at `path`, e.g. TEST_LIB_PATH); raise PteError -- never fall back -- if it is missing."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise PteError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    try:
        # PyTorch-ROCm bundles its own libamdhip64; loading it FIRST makes libpte bind to the same HIP
        # runtime, so that torch device tensors / RCCL and the engine can share streams and pointers in
        # one process (two runtimes in one process: the second one sees no GPU).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    vp = C.c_void_p
    L.pte_default_config.argtypes = [C.POINTER(PteConfig)]
    L.pte_create.argtypes = [C.POINTER(PteConfig), C.POINTER(vp)]
    L.pte_destroy.argtypes = [vp]
    L.pte_last_error.restype = C.c_char_p
    L.pte_last_error.argtypes = [vp]
    L.pte_set_schedule.argtypes = [vp, dp, C.c_int64]
    L.pte_get_schedule.argtypes = [vp, dp]
    L.pte_set_explorer_adaptation.argtypes = [vp, C.c_double, dp, C.c_int64]
    L.pte_explore.argtypes = [vp, C.c_int64]
    L.pte_swap.argtypes = [vp, C.c_int64]
    L.pte_run_scans.argtypes = [vp, C.c_int64, C.c_int64]
    L.pte_reduce.argtypes = [vp]
    L.pte_get_swap_acceptance.argtypes = [vp, dp, ip]
    L.pte_get_log_sum_ratio.argtypes = [vp, dp, ip, dp, ip]
    L.pte_get_round_trip.argtypes = [vp, ip, ip]
    L.pte_get_index_process.argtypes = [vp, ip, ip]
    L.pte_get_explorer_stats.argtypes = [vp, dp, ip, dp, ip]
    L.pte_get_automala_stats.argtypes = [vp, dp, ip, dp, ip]
    L.pte_get_online.argtypes = [vp, dp, dp, ip]
    L.pte_get_state.argtypes = [vp, dp, ip, up]
    L.pte_set_state.argtypes = [vp, dp, ip, up]
    L.pte_timing_reset.argtypes = [vp, C.c_int]
    L.pte_timing_get.argtypes = [vp, C.c_int, dp, ip]
    L.pte_timing_get_samples.argtypes = [vp, C.c_int, dp, C.c_int64, ip]
    L.pte_test_rng_fill.argtypes = [C.c_int32, up, C.c_int32, C.c_int64, dp]
    L.pte_test_sqr_norm.argtypes = [C.c_int32, dp, C.c_int64, C.c_int64, dp]
    L.pte_test_quotient.argtypes = [C.c_int32, dp, dp, C.c_int64, dp, C.POINTER(C.c_int32)]
    i32p = C.POINTER(C.c_int32)
    L.pte_shard_info.argtypes = [vp, ip, ip, ip]
    L.pte_swap_begin.argtypes = [vp, C.c_int64, dp, i32p]
    L.pte_swap_finish.argtypes = [vp, C.c_int64, dp, i32p]
    L.pte_boundary_payload_bytes.argtypes = [vp]
    L.pte_boundary_export.argtypes = [vp, C.c_int, C.c_void_p, C.c_int]
    L.pte_boundary_import.argtypes = [vp, C.c_int, C.c_void_p, C.c_int]
    L.pte_get_index_process_shard.argtypes = [vp, ip, ip, ip]
    L.pte_get_replica_ids.argtypes = [vp, ip]
    L.pte_boundary_payload_bytes.restype = C.c_int64
    L.pte_get_online_log_density.argtypes = [vp, dp, dp]
    L.pte_get_energy_ac1.argtypes = [vp, dp, ip, dp]
    L.pte_get_traces.argtypes = [vp, dp, ip]
    L.pte_set_variational_reference.argtypes = [vp, dp, dp, C.c_int64, C.POINTER(C.c_int32)]
    L.pte_get_stream.argtypes = [vp]
    L.pte_get_stream.restype = C.c_void_p
    L.pte_shard_message_bytes.argtypes = [vp]
    L.pte_shard_message_bytes.restype = C.c_int64
    L.pte_shard_set_buffers.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pte_shard_scan_begin.argtypes = [vp, C.c_int64, i32p]
    L.pte_shard_scan_finish.argtypes = [vp, C.c_int64]
    L.pte_shard_sync.argtypes = [vp, ip]
    u8p = C.POINTER(C.c_uint8)
    L.pte_comm_allow_library_override.argtypes = [C.c_int32]
    L.pte_comm_library.argtypes = [C.c_char_p, C.c_int64, i32p]
    L.pte_comm_unique_id.argtypes = [u8p]
    L.pte_comm_init.argtypes = [vp, u8p]
    L.pte_comm_destroy.argtypes = [vp]
    L.pte_comm_info.argtypes = [vp, i32p, i32p, ip]
    L.pte_comm_barrier.argtypes = [vp]
    L.pte_comm_allreduce.argtypes = [vp, dp, C.c_int64, C.c_int32]
    L.pte_comm_allgather.argtypes = [vp, C.c_void_p, C.c_int64, C.c_void_p]
    L.pte_group_run_scans.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int64]
    L.pte_set_rng_policy.argtypes = [C.c_int32, C.c_uint32]
    L.pte_get_rng_policy.argtypes = [C.c_int32, C.POINTER(C.c_uint32)]
    L.pte_kernel_name.argtypes = [vp]
    L.pte_kernel_name.restype = C.c_char_p
    L.pte_scan_loop_name.argtypes = [vp]
    L.pte_scan_loop_name.restype = C.c_char_p
    L.pte_scan_loop_info.argtypes = [vp, ip, ip, ip]
    L.pte_scan_loop_stats.argtypes = [vp, ip, ip, i32p]
    for name in EXPORTS:
        if name not in ("pte_last_error", "pte_boundary_payload_bytes", "pte_get_stream", "pte_shard_message_bytes", "pte_kernel_name", "pte_scan_loop_name"):
            getattr(L, name).restype = C.c_int
    if _comm_override_allowed:
        L.pte_comm_allow_library_override(1)
    _libs[path] = L
    return L
