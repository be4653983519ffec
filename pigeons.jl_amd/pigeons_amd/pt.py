"""pigeons() / Inputs / PT: the reference's top-level surface (src/api.jl:8-19,
src/pt/Inputs.jl:9-131, src/pt/PT.jl:6-51, src/pt/pigeons.jl:12-55,152-162,
src/pt/Iterators.jl:9-49) driving the MI355X engine for the explore-then-swap loop.

Everything inside `while next_scan!` runs on the GPU (one C-ABI call per round);
what the reference runs a logarithmic number of times (adapt, report) stays host-side.
"""
import math
import time
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np

from . import _lib
from .engine import Engine
from . import tempering as T


# ---- targets (src/targets/toy_mvn_target.jl, src/paths/ScaledPrecisionNormalPath.jl) ---------
@dataclass
class ScaledPrecisionNormalPath:
    precision0: float = 1.0
    precision1: float = 10.0
    dim: int = 1

    def precision(self, beta):
        return (1.0 - beta) * self.precision0 + beta * self.precision1


def toy_mvn_target(dim):
    """src/targets/toy_mvn_target.jl:8"""
    return ScaledPrecisionNormalPath(1.0, 10.0, dim)


def analytic_lognormalization(path):
    """src/paths/ScaledPrecisionNormalPath.jl:66-71"""
    return 0.5 * path.dim * (math.log(path.precision0) - math.log(path.precision1))


def analytic_cumulativebarrier(path):
    """src/paths/ScaledPrecisionNormalPath.jl:56-64"""
    from scipy.special import beta as beta_fn
    b = beta_fn(path.dim / 2.0, path.dim / 2.0)

    def cumulativebarrier(beta):
        sigma0 = 1.0 / math.sqrt(path.precision0)
        sigmab = 1.0 / math.sqrt(path.precision(beta))
        return 2 ** (2.0 - path.dim) / b * math.log(sigma0 / sigmab)
    return cumulativebarrier


@dataclass
class ScaledPrecisionNormalLogPotential:
    """src/paths/ScaledPrecisionNormalPath.jl:14-20 (used as a `reference`)"""
    precision: float = 1.0
    dim: int = 1


@dataclass
class Funnel:
    """Neal's funnel as a LogDensityProblems target (reference test/supporting/dimensional-analysis.jl:33-48):
    z[1] ~ Normal(0, 3), z[i] ~ Normal(0, exp(z[1]/2)); initialization = zeros(dim).  Tempered through the
    default InterpolatingPath(reference, target) (src/targets/target.jl:72-75)."""
    dim: int = 2


@dataclass
class GaussianReference:
    """src/variational/GaussianReference.jl:4-17: mean-field Gaussian variational reference, refitted every round from
    the target chains' online mean / standard deviation once `first_tuning_round` is reached."""
    mean: Any = None
    standard_deviation: Any = None
    first_tuning_round: int = 6


@dataclass
class InterpolatingPath:
    """src/paths/InterpolatingPath.jl: (1 - beta) ref + beta target; what update_path_variational builds
    (src/variational/variational.jl:36-41)."""
    ref: Any = None
    target: Any = None


@dataclass
class IsingLogPotential:
    """2-D Ising model, p(state) ∝ exp(beta * sum of neighbour products) (reference examples/ising.jl:6-9,74).
    The reference distribution is IsingLogPotential(0.0, base_length) (ising.jl:77); states are
    base_length x base_length spin matrices, exposed here as 0/1 vectors in row-major order."""
    beta: float = 1.0
    base_length: int = 5


@dataclass
class IsingMetropolis:
    """examples/ising.jl:91-93"""
    n_steps: int = 3


@dataclass
class TestSwapper:
    """src/swap/pair_swapper.jl:100-149"""
    constant_swap_accept_pr: float = 1.0
    __test__ = False


# ---- explorers (src/explorers/*.jl) ------------------------------------------------------------
@dataclass
class SliceSampler:
    """src/explorers/SliceSampler.jl:8-20"""
    w: float = 10.0
    p: int = 20
    n_passes: int = 3
    max_iter: int = 1024


@dataclass
class ToyExplorer:
    """src/explorers/ToyExplorer.jl:5"""


@dataclass
class IdentityPreconditioner:
    """src/explorers/Preconditioner.jl:14"""


@dataclass
class DiagonalPreconditioner:
    """src/explorers/Preconditioner.jl:22"""


@dataclass
class MixDiagonalPreconditioner:
    """src/explorers/Preconditioner.jl:43-52 (default proportions 1//3, 1//3)"""
    p0: float = 1.0 / 3.0
    p1: float = 1.0 / 3.0


@dataclass
class AutoMALA:
    """src/explorers/AutoMALA.jl:29-68"""
    base_n_refresh: int = 3
    exponent_n_refresh: float = 0.35
    step_size: float = 1.0
    preconditioner: Any = field(default_factory=MixDiagonalPreconditioner)
    estimated_target_std_deviations: Any = None


@dataclass
class MALA:
    """src/explorers/MALA.jl:19-61 (the step size is NOT adapted; the preconditioner is)"""
    base_n_refresh: int = 3
    exponent_n_refresh: float = 0.35
    step_size: float = 1.0
    preconditioner: Any = field(default_factory=MixDiagonalPreconditioner)
    estimated_target_std_deviations: Any = None


@dataclass
class Compose:
    """src/explorers/Compose.jl:5-8: deterministic composition, e.g. Compose(SliceSampler(), AutoMALA())"""
    first: Any = None
    second: Any = None


def default_explorer(target):
    if isinstance(target, ScaledPrecisionNormalPath):
        return ToyExplorer()           # src/targets/toy_mvn_target.jl:13
    if isinstance(target, TestSwapper):
        return None                    # src/swap/pair_swapper.jl:139
    if isinstance(target, IsingLogPotential):
        return IsingMetropolis()       # examples/ising.jl:94
    return SliceSampler()              # src/targets/target.jl:20


# ---- recorder builders (src/recorders/recorder.jl) ----------------------------------------------
def log_sum_ratio(): return "log_sum_ratio"
def swap_acceptance_pr(): return "swap_acceptance_pr"
def round_trip(): return "round_trip"
def index_process(): return "index_process"
def online(): return "online"
def traces(): return "traces"
def energy_ac1(): return "energy_ac1"
def timing_extrema(): return "timing_extrema"
def allocation_extrema(): return "allocation_extrema"
def explorer_acceptance_pr(): return "explorer_acceptance_pr"
def explorer_n_steps(): return "explorer_n_steps"


def record_default():
    """src/pt/Inputs.jl:107-111"""
    return [log_sum_ratio, timing_extrema, allocation_extrema]


def record_online():
    """src/pt/Inputs.jl:116-123"""
    return [log_sum_ratio, timing_extrema, allocation_extrema, round_trip, energy_ac1, online]


@dataclass
class Inputs:
    """src/pt/Inputs.jl:9-102 (fields the hot path reads)."""
    target: Any = None
    seed: int = 1
    n_rounds: int = 10
    n_chains: int = 10
    n_chains_variational: int = 0
    reference: Any = None
    variational: Any = None
    checkpoint: bool = False
    record: List = field(default_factory=record_default)
    checked_round: int = 0
    multithreaded: bool = False
    explorer: Any = None
    extractor: Any = None
    show_report: bool = True
    extended_traces: bool = False
    device: int = 0                 # HIP device ordinal (not in the reference)


@dataclass
class Iterators:
    """src/pt/Iterators.jl:9-25"""
    round: int = 0
    scan: int = 0


def n_scans_in_round(iterators):
    return 2 ** iterators.round


@dataclass
class ReducedRecorders:
    """The reduced `recorders` NamedTuple of one round (src/recorders/recorders.jl:88-120)."""
    swap_acceptance_pr: Any = None      # (mean[N-1], n[N-1]) keyed (c, c+1)
    log_sum_ratio: Any = None           # (up[N-1], up_n, dn[N-1], dn_n)
    round_trip: Any = None              # (n_tempered_restarts, n_round_trips)
    index_process: Any = None           # int64 [replica][scan]
    explorer_acceptance_pr: Any = None  # (mean[N], n[N]) keyed by chain
    explorer_n_steps: Any = None        # (sum[N], n[N])
    am_factors: Any = None              # (mean[N], n[N])       src/explorers/AutoMALA.jl:277
    reversibility_rate: Any = None      # (mean[N], n[N])       src/explorers/AutoMALA.jl:294
    online: Any = None                  # (mean[d], var[d], n)
    online_log_density: Any = None      # (mean, var): entry d+1 of the online sample [state; log density]
    energy_ac1: Any = None              # (cor[N], n[N], moments[N,5]) keyed by chain      recorder.jl:113
    traces: Any = None                  # float64 [scan][d+1], target chain               recorder.jl:27
    timing_extrema: Any = None          # {"round": seconds}


@dataclass
class NonReversiblePT:
    """src/tempering/NonReversiblePT.jl:7-50"""
    path: Any
    schedule: T.Schedule
    communication_barriers: Optional[T.CommunicationBarriers] = None


@dataclass
class StabilizedPT:
    """src/tempering/StabilizedPT.jl:8-51 with inputs.variational == nothing (both legs keep the fixed reference).
    Global chain order (1-based i): i <= n_var -> variational leg chain i; i > n_var -> fixed leg chain
    n_fixed - (i - n_var) + 1 (create_replica_indexer, :86-104)."""
    fixed_leg: NonReversiblePT
    variational_leg: NonReversiblePT

    @property
    def n_fixed(self): return len(self.fixed_leg.schedule.grids)

    @property
    def n_var(self): return len(self.variational_leg.schedule.grids)

    @property
    def path(self): return self.fixed_leg.path

    @property
    def schedule(self):
        """per-chain grid in global chain order (concatenate_log_potentials, :67-69)"""
        return T.Schedule(np.concatenate([self.variational_leg.schedule.grids, self.fixed_leg.schedule.grids[::-1]]), check=False)

    @property
    def communication_barriers(self): return self.fixed_leg.communication_barriers      # global_barrier(::StabilizedPT), :115

    def is_reference(self, chain): return chain == 1 or chain == self.n_fixed + self.n_var           # VariationalDEO.jl:20
    def is_target(self, chain): return chain == self.n_var or chain == self.n_var + 1                # VariationalDEO.jl:21
    def leg_of(self, chain): return "variational" if chain <= self.n_var else "fixed"                 # indexer.i2t


@dataclass
class Shared:
    """src/pt/Shared.jl:12-48"""
    iterators: Iterators
    tempering: NonReversiblePT
    explorer: Any
    reports: list


class PT:
    """src/pt/PT.jl:6-51.  `replicas` is the device engine.

    Sharding (not in the reference's PT; replaces its MPI `EntangledReplicas`):
      n_shards=G           G chain-shards driven from this process (LoopbackShards; single-GPU tests;
                           device_messages=True: the device-resident exchange kernels of the RCCL path)
                           transport="group": the library's own pte_group_run_scans
      rank=r, world=G      this process owns shard r of G.  HIP engines: RcclShard -- the boundary exchange is RCCL
                           send/recv inside libpte (pte_comm_init; comm_id = the 128-byte id if the caller already
                           distributed it, else it is broadcast over torch.distributed).  transport="host" (and
                           engines without comm_init: the CPU tests' oracle shards): DistShard, host-driven over gloo.
    debug_kernel           pte_config.debug_kernel (0 = the default kernel)
    reference_reduction    PTE_RECORD_REFERENCE_REDUCTION: swap_acceptance_pr / log_sum_ratio from per-replica Mean / LogSum fits merged over
                           the replica-index tree (src/recorders/recorders.jl:88-130, src/mpi_utils/Entangler.jl:188-251) instead of the
                           device's chain-keyed sums -- the adapted schedule then equals the oracle's bit for bit (chain-shards replay their own pairs).
    """

    def __init__(self, inputs: Inputs, n_shards=1, rank=0, world=1, dist_device=None, engine_factory=None, device_messages=False,
                 transport=None, comm_id=None, debug_kernel=0, reference_reduction=False):
        self.inputs = inputs
        target = inputs.target
        explorer = inputs.explorer if inputs.explorer is not None else default_explorer(target)
        N = inputs.n_chains
        n_var = int(getattr(inputs, "n_chains_variational", 0) or 0)
        if n_var > 0 and N > 0:                         # create_tempering, src/tempering/tempering.jl:64-70
            tempering = StabilizedPT(NonReversiblePT(target, T.equally_spaced_schedule(N)),
                                     NonReversiblePT(target, T.equally_spaced_schedule(n_var)))
        else:
            tempering = NonReversiblePT(target, T.equally_spaced_schedule(N))
        self.shared = Shared(Iterators(), tempering, explorer, [])
        self.reduced_recorders = ReducedRecorders()
        names = {b() for b in inputs.record}
        flags = 0
        if "round_trip" in names:
            flags |= _lib.RECORD_ROUND_TRIP
        if "index_process" in names:
            flags |= _lib.RECORD_INDEX_PROCESS
        if "online" in names:
            flags |= _lib.RECORD_ONLINE
        if inputs.variational is not None:
            if not isinstance(inputs.variational, GaussianReference) or not isinstance(target, Funnel):
                raise NotImplementedError("the device engine has GaussianReference on the interpolated (funnel) path only")
            flags |= _lib.RECORD_ONLINE                 # variational_recorder_builders: _transformed_online
        if "traces" in names:
            flags |= _lib.RECORD_TRACES
            if getattr(inputs, "extended_traces", False):
                flags |= _lib.RECORD_TRACES_EXTENDED
        if "energy_ac1" in names:
            flags |= _lib.RECORD_ENERGY_AC1
        if reference_reduction:
            flags |= _lib.RECORD_REFERENCE_REDUCTION | _lib.RECORD_INDEX_PROCESS
        kw = dict(device=inputs.device, n_chains=N, n_chains_variational=n_var, seed=inputs.seed, record_flags=flags,
                  max_scans_per_round=2 ** inputs.n_rounds)
        if isinstance(target, ScaledPrecisionNormalPath):
            kw.update(target=_lib.TARGET_MVN_SCALED_PRECISION, dim=target.dim,
                      target_params=[target.precision0, target.precision1])
        elif isinstance(target, TestSwapper):
            kw.update(target=_lib.TARGET_TEST_SWAPPER, dim=1, target_params=[target.constant_swap_accept_pr])
        elif isinstance(target, IsingLogPotential):
            kw.update(target=_lib.TARGET_ISING, dim=target.base_length ** 2, target_params=[target.beta])
        elif isinstance(target, Funnel):
            ref = inputs.reference
            if not isinstance(ref, ScaledPrecisionNormalLogPotential) or ref.dim != target.dim:
                raise NotImplementedError("the device funnel path needs reference=ScaledPrecisionNormalLogPotential(prec, dim)")
            kw.update(target=_lib.TARGET_FUNNEL, dim=target.dim, target_params=[ref.precision])
        else:
            raise NotImplementedError(
                "target %r has no device log-potential; use the reference CPU path (Pigeons.jl)" % (target,))
        def explorer_kw(ex):
            if ex is None:
                return dict(explorer=_lib.EXPLORER_NONE)
            if isinstance(ex, ToyExplorer):
                return dict(explorer=_lib.EXPLORER_TOY)
            if isinstance(ex, SliceSampler):
                return dict(explorer=_lib.EXPLORER_SLICE, slice_w=ex.w, slice_p=ex.p, slice_n_passes=ex.n_passes,
                            slice_max_iter=ex.max_iter)
            if isinstance(ex, IsingMetropolis):
                return dict(explorer=_lib.EXPLORER_ISING_METROPOLIS, slice_n_passes=ex.n_steps)
            if isinstance(ex, (AutoMALA, MALA)):
                pc = ex.preconditioner
                kind = 0 if isinstance(pc, IdentityPreconditioner) else 1 if isinstance(pc, DiagonalPreconditioner) else 2
                return dict(explorer=_lib.EXPLORER_AUTOMALA if isinstance(ex, AutoMALA) else _lib.EXPLORER_MALA,
                            am_base_n_refresh=ex.base_n_refresh, am_exponent_n_refresh=ex.exponent_n_refresh,
                            am_step_size=ex.step_size, am_preconditioner=kind, am_p0=getattr(pc, "p0", 0.0),
                            am_p1=getattr(pc, "p1", 0.0))
            raise NotImplementedError("explorer %r is not available on the device" % (ex,))
        if isinstance(explorer, Compose):
            if isinstance(explorer.first, Compose) or isinstance(explorer.second, Compose):
                raise NotImplementedError("nested Compose is not available on the device")
            k1, k2 = explorer_kw(explorer.first), explorer_kw(explorer.second)
            shared_keys = (set(k1) & set(k2)) - {"explorer"}
            if any(k1[k] != k2[k] for k in shared_keys):
                raise NotImplementedError("Compose of two samplers of the same family needs identical parameters on the device")
            kw.update(k1); kw.update({k: v for k, v in k2.items() if k != "explorer"}); kw.update(explorer2=k2["explorer"])
        else:
            kw.update(explorer_kw(explorer))
        make = engine_factory or Engine
        if debug_kernel:
            kw.update(debug_kernel=debug_kernel)
            if (debug_kernel & ~(_lib.KERNEL_FLAG_BITS | _lib.KERNEL_TEST_BITS)) not in (0, _lib.KERNEL_SLICE_SEQUENTIAL, _lib.KERNEL_ISING_BYTES) or (debug_kernel & _lib.KERNEL_TEST_BITS):
                kw.update(test_build=True)              # the dominated generations live in libpte_test.so only
        self.shards = None
        if n_shards > 1:
            from .sharded import LoopbackShards
            self.shards = LoopbackShards([make(rank=g, world_size=n_shards, **kw) for g in range(n_shards)], device_messages=device_messages,
                                         transport=transport)
            self.replicas = self.shards.engines[0]
        elif world > 1:
            from .sharded import DistShard, RcclShard
            self.replicas = make(rank=rank, world_size=world, **kw)
            if transport != "host" and hasattr(self.replicas, "comm_init"):
                self.shards = RcclShard(self.replicas, rank, world, id_bytes=comm_id)
            else:
                self.shards = DistShard(self.replicas, rank, world, device=dist_device)
        else:
            self.replicas = make(**kw)


def next_round(pt):
    """src/pt/Iterators.jl:27-35"""
    it = pt.shared.iterators
    if it.round + 1 <= pt.inputs.n_rounds:
        it.round += 1
        return True
    return False


def run_one_round(pt):
    """src/pt/pigeons.jl:46-55: the scan loop runs fused on the device."""
    it = pt.shared.iterators
    eng = pt.shards if pt.shards is not None else pt.replicas
    n = n_scans_in_round(it)
    t0 = time.perf_counter()
    eng.run_scans(1, n)                  # explore!; communicate! for scan = 1..2^round (synchronous)
    elapsed = time.perf_counter() - t0
    it.scan = 0
    return reduce_recorders(pt, elapsed)


def reduce_recorders(pt, elapsed=None):
    """src/recorders/recorders.jl:88-120"""
    if pt.shards is not None:
        r = pt.shards.reduce()
        r.timing_extrema = {"round": elapsed}
        return r
    eng = pt.replicas
    eng.reduce()
    am, an, ss, sn = eng.explorer_stats()
    fm, fn, rm, rn = eng.automala_stats()
    r = ReducedRecorders(
        am_factors=(fm, fn), reversibility_rate=(rm, rn),
        swap_acceptance_pr=eng.swap_acceptance(),
        log_sum_ratio=eng.log_sum_ratio(),
        round_trip=eng.round_trip(),
        index_process=eng.index_process(),
        explorer_acceptance_pr=(am, an),
        explorer_n_steps=(ss, sn),
        online=eng.online(),
        online_log_density=eng.online_log_density(),
        energy_ac1=eng.energy_ac1(),
        traces=eng.traces(),
        timing_extrema={"round": elapsed},
    )
    return r


def adapt(pt, reduced):
    """src/pt/pigeons.jl:152-162 with adapt_tempering (src/tempering/NonReversiblePT.jl:52-66)."""
    temp = pt.shared.tempering
    pt.reduced_recorders = reduced
    adapt_explorer(pt, reduced)
    update_variational(pt, reduced)
    if isinstance(pt.inputs.target, TestSwapper) or len(temp.schedule.grids) == 1:
        return pt
    mean, n = reduced.swap_acceptance_pr
    rej = T.rejections(mean, n)
    if isinstance(temp, StabilizedPT):
        # adapt_tempering(::StabilizedPT) (StabilizedPT.jl:53-65): each leg from its own pairs; the fixed leg reads
        # (N-1,N), (N-2,N-1), ... from its reference towards the target
        nv, nf, Ntot = temp.n_var, temp.n_fixed, temp.n_var + temp.n_fixed
        legs = []
        for leg, r in ((temp.variational_leg, rej[:nv - 1]), (temp.fixed_leg, rej[Ntot - 2 - np.arange(nf - 1)] if nf > 1 else rej[:0])):
            old = leg.schedule.grids
            if len(old) == 1:
                legs.append(leg); continue
            legs.append(NonReversiblePT(leg.path, T.Schedule(T.optimal_schedule(r, old, len(old))), T.CommunicationBarriers(r, old)))
        pt.shared.tempering = StabilizedPT(fixed_leg=legs[1], variational_leg=legs[0])
        pt.replicas.set_schedule(pt.shared.tempering.schedule.grids)
        return pt
    old = temp.schedule.grids
    new_sched = T.Schedule(T.optimal_schedule(rej, old, len(old)))
    barriers = T.CommunicationBarriers(rej, old)
    pt.shared.tempering = NonReversiblePT(temp.path, new_sched, barriers)
    (pt.shards if pt.shards is not None else pt.replicas).set_schedule(new_sched.grids)   # discretize on the device
    return pt


def update_variational(pt, reduced):
    """update_path_if_needed (src/variational/variational.jl:28-41): from first_tuning_round on, refit the
    GaussianReference to the target chains' online statistics and put it at the reference end of the variational leg
    (of the only leg when there is one)."""
    var = pt.inputs.variational
    if not isinstance(var, GaussianReference) or pt.shared.iterators.round < var.first_tuning_round:
        return
    mean, variance, _n = reduced.online
    ref = GaussianReference(np.array(mean, dtype=np.float64), np.sqrt(np.asarray(variance, dtype=np.float64)), var.first_tuning_round)
    pt.inputs.variational = ref
    temp = pt.shared.tempering
    N = pt.replicas.N
    if isinstance(temp, StabilizedPT):
        temp.variational_leg.path = InterpolatingPath(ref, pt.inputs.target)
        uses = np.array([1 if c < temp.n_var else 0 for c in range(N)], dtype=np.int32)
    else:
        temp.path = InterpolatingPath(ref, pt.inputs.target)
        uses = np.ones(N, dtype=np.int32)
    pt.replicas.set_variational_reference(ref.mean, ref.standard_deviation, uses)


def adapt_explorer(pt, reduced):
    """adapt_explorer (src/explorers/AutoMALA.jl:70-79, MALA.jl:63-69, Compose.jl:10-14, Preconditioner.jl:54-55)."""
    def adapt_one(ex):
        if isinstance(ex, Compose):
            return Compose(adapt_one(ex.first), adapt_one(ex.second))
        if not isinstance(ex, (AutoMALA, MALA)) or reduced.am_factors is None:
            return ex
        std = None
        if not isinstance(ex.preconditioner, IdentityPreconditioner):
            std = np.sqrt(np.asarray(reduced.online[1], dtype=np.float64))
        new_step = ex.step_size
        if isinstance(ex, AutoMALA):
            fm, fn = reduced.am_factors
            present = np.asarray(fn) > 0
            if present.any():
                acc = 0.0                                # (a plain left-to-right sum, as the oracle's: np.mean sums in eight interleaved partial sums)
                for v in np.asarray(fm, dtype=np.float64)[present]:
                    acc += float(v)
                new_step = ex.step_size * (acc / float(int(present.sum())))
        adapted.append((new_step, std))
        return type(ex)(ex.base_n_refresh, ex.exponent_n_refresh, new_step, ex.preconditioner, std)
    adapted = []
    pt.shared.explorer = adapt_one(pt.shared.explorer)
    if not adapted:
        return
    new_step, std = adapted[-1]
    eng = pt.shards if pt.shards is not None else pt.replicas
    if hasattr(eng, "engines"):
        for e in eng.engines:
            e.set_explorer_adaptation(new_step, std)
    else:
        pt.replicas.set_explorer_adaptation(new_step, std)


def stepping_stone_pair(pt):
    up, un, dn, dnn = pt.reduced_recorders.log_sum_ratio
    temp = pt.shared.tempering
    if isinstance(temp, StabilizedPT):                 # only the variational leg's keys (stepping_stone.jl:53-65)
        k = temp.n_var - 1
        up, un, dn, dnn = up[:k], un[:k], dn[:k], dnn[:k]
    return T.stepping_stone_pair(up, un, dn, dnn)


def stepping_stone(pt):
    return T.stepping_stone(stepping_stone_pair(pt))


def energy_ac1s(pt, skip_reference=False):
    """src/recorders/recorder.jl:156-173: autocorrelation of the log density before / after an exploration
    step, one entry per chain."""
    cor = np.asarray(pt.reduced_recorders.energy_ac1[0])
    if not skip_reference or len(cor) <= 1:
        return cor
    return cor[1:-1] if isinstance(pt.shared.tempering, StabilizedPT) else cor[1:]     # is_reference: chain 1 (and N with two legs)


def sample_names(pt):
    """src/pt/state.jl:60-63,91"""
    d = pt.replicas.d
    return ["param_%d" % (i + 1) for i in range(d)] + ["log_density"]


def sample_array(pt):
    """src/pt/process_sample.jl:19-32: [iteration, variable, target chain] of the last round's traces."""
    tr = pt.reduced_recorders.traces
    if tr is None or tr.size == 0:
        raise ValueError("no traces recorded: pass record=[traces]")
    if tr.ndim == 3:                                   # extended_traces: [scan][chain][var] -> [scan, var, chain]
        return np.transpose(tr, (0, 2, 1)).copy()
    return tr[:, :, None].copy()


def get_sample(pt, chain=None, scan=None):
    """src/pt/process_sample.jl get_sample(pt, chain[, scan]) for the target chain (1-based scan)."""
    tr = pt.reduced_recorders.traces
    if tr.ndim == 3:                                   # extended_traces
        tr = tr[:, (pt.inputs.n_chains if chain is None else chain) - 1, :]
    elif chain is not None and chain != pt.inputs.n_chains:
        raise ValueError("traces were recorded for the target chain only: pass extended_traces=True")
    return tr if scan is None else tr[scan - 1]


def mean(pt):
    """Statistics.mean(pt) (src/recorders/OnlineStateRecorder.jl:16): online mean of [state; log density]."""
    r = pt.reduced_recorders
    return np.concatenate([r.online[0], [r.online_log_density[0]]])


def var(pt):
    """Statistics.var(pt) (src/recorders/OnlineStateRecorder.jl:21)."""
    r = pt.reduced_recorders
    return np.concatenate([r.online[1], [r.online_log_density[1]]])


def n_round_trips(pt):
    return pt.reduced_recorders.round_trip[1]


def n_tempered_restarts(pt):
    return pt.reduced_recorders.round_trip[0]


def global_barrier(pt):
    return pt.shared.tempering.communication_barriers.globalbarrier


def global_barrier_variational(pt):
    """src/tempering/StabilizedPT.jl:117"""
    return pt.shared.tempering.variational_leg.communication_barriers.globalbarrier


def target_chains(pt):
    """src/pt/process_sample.jl:41-44 (1-based chain indices)"""
    temp = pt.shared.tempering
    n = pt.replicas.N
    return [i for i in range(1, n + 1) if (temp.is_target(i) if isinstance(temp, StabilizedPT) else i == n)]


def last_round_max_time(pt):
    """src/recorders/recorder.jl:137"""
    return pt.reduced_recorders.timing_extrema["round"]


def report(pt):
    """One line of the reference's report table (all_reports(), src/pt/report.jl:8-26); an item whose recorder is
    missing is skipped, as there."""
    it = pt.shared.iterators
    red = pt.reduced_recorders
    row = {"scans": n_scans_in_round(it)}
    if red.round_trip is not None:
        row["restarts"] = red.round_trip[0]
    n_total = pt.replicas.N
    if not isinstance(pt.inputs.target, TestSwapper) and n_total > 1:
        m, n = red.swap_acceptance_pr
        row["Λ"] = global_barrier(pt)
        if isinstance(pt.shared.tempering, StabilizedPT) and pt.shared.tempering.variational_leg.communication_barriers is not None:
            row["Λ_var"] = global_barrier_variational(pt)
        row.update({"time(s)": last_round_max_time(pt), "log(Z₁/Z₀)": stepping_stone(pt),
                    "min(α)": float(np.min(m)), "mean(α)": float(np.mean(m))})
        names = {b() for b in pt.inputs.record}
        if "energy_ac1" in names and red.energy_ac1 is not None:
            rho = np.abs(energy_ac1s(pt, True)); rho = rho[np.isfinite(rho)]
            if rho.size:
                row["max|ρ|"], row["mean|ρ|"] = float(rho.max()), float(rho.mean())
        am, an = red.explorer_acceptance_pr if red.explorer_acceptance_pr is not None else (None, None)
        if am is not None and np.any(np.asarray(an) > 0):
            a = np.asarray(am)[np.asarray(an) > 0]
            row["min(αₑ)"], row["mean(αₑ)"] = float(a.min()), float(a.mean())
        if red.reversibility_rate is not None and np.any(np.asarray(red.reversibility_rate[1]) > 0):
            rr = np.asarray(red.reversibility_rate[0])[np.asarray(red.reversibility_rate[1]) > 0]
            row["min(RR)"], row["mean(RR)"] = float(rr.min()), float(rr.mean())
    else:
        row["time(s)"] = last_round_max_time(pt)
    if red.round_trip is not None:
        row["round trips"] = red.round_trip[1]
    pt.shared.reports.append(row)
    if pt.inputs.show_report:
        print("  ".join("%s=%s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in row.items()))


def pigeons(pt_or_none=None, **kwargs):
    """pigeons(; target, seed, n_rounds, n_chains, explorer, record, ...) -> PT  (src/api.jl:8-19)."""
    exec_folder = kwargs.pop("exec_folder", None)
    pt = pt_or_none if pt_or_none is not None else PT(Inputs(**kwargs))
    if exec_folder is not None:
        pt.exec_folder = exec_folder
    elif pt.inputs.checkpoint and getattr(pt, "exec_folder", None) is None:
        from .checkpoint import next_exec_folder           # the reference always has an exec folder when checkpoint = true
        pt.exec_folder = next_exec_folder()                # (results/all/<time stamp>, src/pt/pigeons.jl:20, checkpoint.jl:110-113)
    while next_round(pt):
        reduced = run_one_round(pt)
        pt = adapt(pt, reduced)
        report(pt)
        if pt.inputs.checkpoint and getattr(pt, "exec_folder", None):     # src/pt/pigeons.jl:20, checkpoint.jl:110-113
            from .checkpoint import write_checkpoint
            write_checkpoint(pt)
    return pt
