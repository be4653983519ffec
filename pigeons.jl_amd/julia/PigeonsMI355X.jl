# PigeonsMI355X.jl -- the reference-side binding a Pigeons.jl maintainer would add (a package extension or a small package).
#
# INERT IN THE BUILD IMAGE (no Julia toolchain exists there; never executed).  It binds the dispatch hooks that bracket the
# explore-then-swap hot path of Pigeons.jl v0.4.10, plus construction / adaptation / recorders, to the C ABI of include/pte.h.
# Conventions follow the reference's only FFI precedent, ext/PigeonsBridgeStanExt/interface.jl:118-183 (Cint return code,
# message fetched on error).
#
# How a run reaches the device -- nothing in Pigeons.jl is edited, every hook is reached by ordinary dispatch:
#
#     pt = pigeons(target = on_mi355x(toy_mvn_target(1024)), n_chains = 1024, explorer = SliceSampler(),
#                  record = [round_trip, log_sum_ratio])
#
#   PT(inputs)  (src/pt/PT.jl:49-54)
#     Shared(inputs)                      create_path / default_explorer / default_reference forward to the wrapped target (below)
#     create_replicas(inputs, shared)     `inputs::Inputs{<:OnDevice}`: dispatch on the FIRST type parameter of Inputs (its `target`
#                                          field, src/pt/Inputs.jl:9-11) -- the method PT(inputs) itself calls (src/replicas/replicas.jl:65-68)
#   run_one_round!(pt)  (src/pt/pigeons.jl:46-55), unmodified:
#     explore!(pt, explorer, ::Val)       loops `for replica in locals(pt.replicas)`: locals(::DeviceReplicas) is ONE DeviceBatch, and
#                                          explore!(pt, ::DeviceBatch, explorer) is one ccall for all replicas (dispatch on the replica)
#     communicate!(pt) -> swap!(pair_swapper, pt.replicas, swap_graph)       dispatch on the replicas container
#     reduce_recorders!(pt, pt.replicas)                                     dispatch on the replicas container
#   adapt(pt, reduced_recorders)  (src/pt/pigeons.jl:152-162), unmodified: adapt_tempering runs on the host as always; the new
#     schedule / explorer parameters reach the device from `after_adapt!` (called from the fused run_one_round! below, or by hand).
# The FUSED scan loop (one ccall per round instead of two per scan) needs a method of run_one_round! itself; it dispatches on
# PT's first type parameter (the `inputs` field, src/pt/PT.jl:6-12) and is optional.
module PigeonsMI355X

using Pigeons
using Pigeons: Inputs, Shared, Replica, PT, SliceSampler, AutoMALA, MALA, Compose, ToyExplorer, ScaledPrecisionNormalPath,
               InterpolatingPath, ScaledPrecisionNormalLogPotential, GroupBy, Mean, Sum, Variance, CovMatrix, Group, LogSum,
               RoundTripRecorder, OnlineStateRecorder, EqualWeight
using Random: AbstractRNG

const libpte = "libpte.so"

# ---- include/pte.h mirrored ------------------------------------------------------------------------------------------------
const TARGET_MVN, TARGET_TEST_SWAPPER, TARGET_FUNNEL, TARGET_ISING = Int32(0), Int32(1), Int32(2), Int32(3)
const EXPLORER_NONE, EXPLORER_TOY, EXPLORER_SLICE, EXPLORER_AUTOMALA, EXPLORER_ISING, EXPLORER_MALA = Int32.((0, 1, 2, 3, 4, 5))
const RECORD_ROUND_TRIP, RECORD_INDEX_PROCESS, RECORD_ONLINE, RECORD_TRACES, RECORD_ENERGY_AC1, RECORD_TRACES_EXTENDED =
    UInt32.((1, 2, 4, 8, 16, 32))
const RECORD_REFERENCE_REDUCTION = UInt32(64)   # swap recorders reduced by per-replica Mean / LogSum fits + tree merge, replayed in pte_reduce (include/pte.h)

# mirror of `pte_config` -- field order and types must match; pte_create checks struct_size and abi_version, so a layout
# mismatch fails loudly instead of corrupting memory
Base.@kwdef mutable struct PteConfig
    struct_size::UInt32 = 0
    abi_version::UInt32 = 2
    device::Int32 = 0
    target::Int32 = TARGET_MVN
    explorer::Int32 = EXPLORER_TOY
    record_flags::UInt32 = 3
    n_chains::Int64 = 10
    dim::Int64 = 1
    seed::UInt64 = 1
    max_scans_per_round::Int64 = 1024
    target_params::NTuple{4,Float64} = (1.0, 10.0, 0.0, 0.0)
    slice_w::Float64 = 10.0
    slice_p::Int32 = 20
    slice_n_passes::Int32 = 3
    slice_max_iter::Int32 = 1024
    am_base_n_refresh::Int32 = 3
    am_exponent_n_refresh::Float64 = 0.35
    am_step_size::Float64 = 1.0
    am_p0::Float64 = 1/3
    am_p1::Float64 = 1/3
    am_preconditioner::Int32 = 2
    rank::Int32 = 0
    world_size::Int32 = 1
    explorer2::Int32 = EXPLORER_NONE      # Compose(explorer, explorer2)
    n_chains_variational::Int64 = 0       # StabilizedPT with variational == nothing
    debug_kernel::Int32 = 0               # PTE_KERNEL_*: 0 = the default kernel of the explorer (never read from the environment)
    reserved0::Int32 = 0
end

# ---- targets on the device -------------------------------------------------------------------------------------------------
"""`on_mi355x(target; device = 0, rank = 0, world_size = 1, reference_reduction = false)`: run the explore-then-swap loop of `target` on an MI355X.
`reference_reduction = true` (PTE_RECORD_REFERENCE_REDUCTION, include/pte.h): the swap recorders are reduced with the reference's own arithmetic --
per-replica Mean / LogSum fits merged over the binary tree on the replica index -- instead of chain-keyed sums; forces :index_process, logs 16 B per
chain and scan.  An explicit option, like `PT(reference_reduction = ...)` on the Python side: nothing here reads the environment (ADVICE r05)."""
struct OnDevice{T}
    target::T
    device::Int
    rank::Int
    world_size::Int
    reference_reduction::Bool
end
on_mi355x(target; device = 0, rank = 0, world_size = 1, reference_reduction = false) = OnDevice(target, device, rank, world_size, reference_reduction)

# everything Shared(inputs) / preflight ask of a target goes to the wrapped one (src/targets/target.jl informal interface)
Pigeons.create_path(t::OnDevice, inputs::Inputs) = Pigeons.create_path(t.target, inputs)
Pigeons.default_explorer(t::OnDevice) = Pigeons.default_explorer(t.target)
Pigeons.default_reference(t::OnDevice) = Pigeons.default_reference(t.target)
Pigeons.initialization(t::OnDevice, rng::AbstractRNG, i::Int) = Pigeons.initialization(t.target, rng, i)
Pigeons.sample_iid!(t::OnDevice, replica, shared) = Pigeons.sample_iid!(t.target, replica, shared)
Pigeons.sample_names(state, t::OnDevice) = Pigeons.sample_names(state, t.target)

"""Neal's funnel of test/supporting/dimensional-analysis.jl:34-52 (a `LogDensity` defined in user code there): the device family
PTE_TARGET_FUNNEL.  Used as `Inputs(target = on_mi355x(DeviceFunnel(128)), reference = ScaledPrecisionNormalLogPotential(1/9, 128))`."""
struct DeviceFunnel; dim::Int; end
"""The Ising model of examples/ising.jl:13-37 (`IsingLogPotential(beta, base_length)`, defined in user code there):
PTE_TARGET_ISING, explorer = the Metropolis sweeps of examples/ising.jl:91-116 (n_sweeps passes)."""
struct DeviceIsing; beta::Float64; base_length::Int; n_sweeps::Int; end

# (target code, dim, target_params, reference precision check) of a wrapped target
device_family(t::ScaledPrecisionNormalPath, inputs) = (TARGET_MVN, t.dim, (t.precision0, t.precision1, 0.0, 0.0))
device_family(t::Pigeons.TestSwapper, inputs) = (TARGET_TEST_SWAPPER, 0, (t.constant_swap_accept_pr, 0.0, 0.0, 0.0))
function device_family(t::DeviceFunnel, inputs)
    ref = inputs.reference
    ref isa ScaledPrecisionNormalLogPotential && ref.dim == t.dim ||
        error("the device funnel path needs reference = ScaledPrecisionNormalLogPotential(precision, $(t.dim)); keep the CPU path otherwise")
    return (TARGET_FUNNEL, t.dim, (ref.precision, 0.0, 0.0, 0.0))
end
device_family(t::DeviceIsing, inputs) = (TARGET_ISING, t.base_length^2, (t.beta, 0.0, 0.0, 0.0))
device_family(t, inputs) = error("target $(typeof(t)) has no device log-potential family (closed set: include/pte.h PTE_TARGET_*); keep the CPU path")

# explorer structs -> pte_config fields (SliceSampler.jl:8-20, AutoMALA.jl:29-68, MALA.jl:19-40, Compose.jl:5-9)
preconditioner_code(::Pigeons.IdentityPreconditioner) = Int32(0)
preconditioner_code(::Pigeons.DiagonalPreconditioner) = Int32(1)
preconditioner_code(p::Pigeons.MixDiagonalPreconditioner) = Int32(2)
function set_explorer!(cfg::PteConfig, ex::SliceSampler, slot)
    cfg.slice_w = ex.w; cfg.slice_p = ex.p; cfg.slice_n_passes = ex.n_passes; cfg.slice_max_iter = ex.max_iter
    return EXPLORER_SLICE
end
set_explorer!(cfg::PteConfig, ::ToyExplorer, slot) = EXPLORER_TOY
function set_explorer!(cfg::PteConfig, ex::AutoMALA, slot)
    cfg.am_base_n_refresh = ex.base_n_refresh; cfg.am_exponent_n_refresh = ex.exponent_n_refresh; cfg.am_step_size = ex.step_size
    cfg.am_preconditioner = preconditioner_code(ex.preconditioner)
    if ex.preconditioner isa Pigeons.MixDiagonalPreconditioner
        cfg.am_p0 = ex.preconditioner.p0; cfg.am_p1 = ex.preconditioner.p1
    end
    return EXPLORER_AUTOMALA
end
function set_explorer!(cfg::PteConfig, ex::MALA, slot)
    cfg.am_base_n_refresh = ex.base_n_refresh; cfg.am_exponent_n_refresh = ex.exponent_n_refresh; cfg.am_step_size = ex.step_size
    cfg.am_preconditioner = preconditioner_code(ex.preconditioner)
    return EXPLORER_MALA
end
set_explorer!(cfg::PteConfig, ::Nothing, slot) = EXPLORER_NONE
set_explorer!(cfg::PteConfig, ex, slot) = error("explorer $(typeof(ex)) has no device kernel; keep the CPU path")

# Inputs.record (recorder builder functions, src/pt/Inputs.jl:57-62) -> PTE_RECORD_*
function record_flags(inputs::Inputs, shared::Shared)
    names = Set(Symbol(b) for b in Pigeons.recorder_builders(inputs, shared))
    f = UInt32(0)
    :round_trip in names && (f |= RECORD_ROUND_TRIP)
    :index_process in names && (f |= RECORD_INDEX_PROCESS)
    (:online in names || :_transformed_online in names) && (f |= RECORD_ONLINE)
    :traces in names && (f |= RECORD_TRACES)
    (:traces in names && inputs.extended_traces) && (f |= RECORD_TRACES_EXTENDED)
    :energy_ac1 in names && (f |= RECORD_ENERGY_AC1)
    :disk in names && error("the disk recorder is not served by the device path (SURVEY.md 2: out of scope)")
    # on_mi355x(target; reference_reduction = true): the reference's own reduction arithmetic for the swap recorders (and :online when :traces is recorded)
    (inputs.target isa OnDevice && inputs.target.reference_reduction) && (f |= RECORD_REFERENCE_REDUCTION | RECORD_INDEX_PROCESS)
    return f
end

# ---- replicas (informal interface src/replicas/replicas.jl:11-40) ------------------------------------------------------------
"""Device-resident `replicas`.  `host_recorders` receives what the host itself records (timing, allocations)."""
mutable struct DeviceReplicas
    handle::Ptr{Cvoid}
    n_chains::Int          # all chains (both legs of a StabilizedPT)
    dim::Int
    first_chain::Int       # 0-based, this rank's shard
    n_local::Int
    world_size::Int
    record_flags::UInt32
    host_recorders
end
"""What `locals(replicas)` iterates over: ALL local replicas as one unit of work (explore! is one launch, not a loop)."""
struct DeviceBatch
    replicas::DeviceReplicas
end
Base.getproperty(b::DeviceBatch, s::Symbol) =          # the two fields generic code reads off `locals(pt.replicas)[1]`
    s === :recorders ? getfield(b, :replicas).host_recorders :
    s === :state ? replica_state(getfield(b, :replicas), 1) : getfield(b, s)

Pigeons.locals(r::DeviceReplicas) = (DeviceBatch(r),)
Pigeons.load(r::DeviceReplicas) = Pigeons.single_process_load(r.n_chains)
Pigeons.entangler(r::DeviceReplicas) = Pigeons.Entangler(r.n_chains; parent_communicator = nothing)   # (only `.load` is read off it)

function check(r::DeviceReplicas, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:pte_last_error, libpte), Cstring, (Ptr{Cvoid},), r.handle))
    error(msg)    # same wording class as the reference's exceptions (SliceSampler.jl:52-60,179-185)
end

# create_replicas(inputs, shared, source)  (src/replicas/replicas.jl:65-98) -- the method PT(inputs) calls
function Pigeons.create_replicas(inputs::Inputs{<:OnDevice}, shared::Shared, source = nothing)
    t = inputs.target
    code, dim, params = device_family(t.target, inputs)
    N = inputs.n_chains
    cfg = PteConfig(device = t.device, target = code, n_chains = N, dim = dim, seed = inputs.seed,
                    max_scans_per_round = 2^inputs.n_rounds, target_params = params,
                    n_chains_variational = inputs.n_chains_variational, rank = t.rank, world_size = t.world_size)
    cfg.struct_size = sizeof(PteConfig)
    ex = shared.explorer
    if t.target isa DeviceIsing
        cfg.explorer = EXPLORER_ISING; cfg.slice_n_passes = t.target.n_sweeps
    elseif ex isa Compose
        cfg.explorer = set_explorer!(cfg, ex.first, 1); cfg.explorer2 = set_explorer!(cfg, ex.second, 2)
    else
        cfg.explorer = set_explorer!(cfg, ex, 1)
    end
    cfg.record_flags = record_flags(inputs, shared)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:pte_create, libpte), Cint, (Ref{PteConfig}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || error(unsafe_string(ccall((:pte_last_error, libpte), Cstring, (Ptr{Cvoid},), C_NULL)))
    info = zeros(Int64, 3)
    ccall((:pte_shard_info, libpte), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), h[], pointer(info, 1), pointer(info, 2), pointer(info, 3))
    r = DeviceReplicas(h[], N + inputs.n_chains_variational, dim, info[1], info[2], t.world_size, cfg.record_flags,
                       Pigeons.create_recorders(inputs, shared))
    finalizer(x -> ccall((:pte_destroy, libpte), Cint, (Ptr{Cvoid},), x.handle), r)
    source === nothing || restore!(r, source)          # FromCheckpoint: pte_set_state from the deserialised Replica structs
    return r
end

# checkpoint fields of Replica (src/pt/checkpoint.jl:110-145): state, chain (1-based on this side), rng words
function replica_states(r::DeviceReplicas)
    x = zeros(r.dim, r.n_local); chain = zeros(Int64, r.n_local); rng = zeros(UInt64, 2, r.n_local)
    check(r, ccall((:pte_get_state, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{UInt64}), r.handle, x, chain, rng))
    return x, chain .+ 1, rng
end
replica_state(r::DeviceReplicas, i::Int) = replica_states(r)[1][:, i]
function restore!(r::DeviceReplicas, source::Pigeons.FromCheckpoint)
    reps = [Pigeons.deserialize("$(source.checkpoint_folder)/replica=$i.jls") for i in (r.first_chain + 1):(r.first_chain + r.n_local)]
    x = reduce(hcat, [Float64.(rep.state) for rep in reps]); chain = Int64[rep.chain - 1 for rep in reps]
    rng = reduce(hcat, [UInt64[rep.rng.seed, rep.rng.gamma] for rep in reps])
    check(r, ccall((:pte_set_state, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{UInt64}), r.handle, x, chain, rng))
end

# ---- the hot path ----------------------------------------------------------------------------------------------------------
# explore!(pt, replica, explorer)  (src/pt/pigeons.jl:101-143): one ccall for ALL local replicas; the bookkeeping around the
# explorer (energy_ac1, online, traces) is fused into the kernels (record_flags)
function Pigeons.explore!(pt, b::DeviceBatch, explorer)
    pt.shared.iterators.scan == 1 && after_adapt!(pt)   # first scan of a round: what adapt() decided reaches the device here
    check(b.replicas, ccall((:pte_explore, libpte), Cint, (Ptr{Cvoid}, Int64), b.replicas.handle, pt.shared.iterators.scan))
end

# swap!(pair_swapper, replicas, swap_graph)  (src/swap/swap.jl:6-39): the DEO parity is iseven(scan) (src/swap/DEO.jl:12);
# OddEven and VariationalOddEven (src/swap/OddEven.jl) both carry it in `.even`
Pigeons.swap!(pair_swapper, r::DeviceReplicas, swap_graph) =
    check(r, ccall((:pte_swap, libpte), Cint, (Ptr{Cvoid}, Int64), r.handle, swap_graph.even ? 2 : 1))

# run_one_round!: the fused `while next_scan!` loop (src/pt/pigeons.jl:46-55), one ccall per round.  Optional: without this method
# the reference's own loop runs with the two hooks above.  PT's first type parameter is its `inputs` field (src/pt/PT.jl:6-12).
@assert fieldnames(PT)[1] === :inputs "PT's field order changed: re-derive the dispatch of the fused run_one_round!"
function Pigeons.run_one_round!(pt::PT{<:Inputs{<:OnDevice}})
    r = pt.replicas::DeviceReplicas
    n = Pigeons.n_scans_in_round(pt.shared.iterators)
    after_adapt!(pt)                                    # the schedule / explorer parameters of this round
    timed = @timed check(r, ccall((:pte_run_scans, libpte), Cint, (Ptr{Cvoid}, Int64, Int64), r.handle, 1, n))
    pt.shared.iterators.scan = 0
    Pigeons.record_timed_if_requested!(r.host_recorders, :round, timed)
    return Pigeons.reduce_recorders!(pt, r)
end

# Which form pte_run_scans takes on this engine (round 5): "" = an explore and a swap launch per scan; otherwise the ONE kernel that runs all the
# scans of the call ("k_scans_slice8", "k_scans_automala": workgroup c keeps chain c, the DEO swap of a pair is a hand-shake between its two
# waves; "k_scans_automala_wg": four consecutive chains per workgroup, inner pairs shake hands through LDS).  Results are bit-identical
# either way; `debug_kernel = PTE_KERNEL_TWO_LAUNCHES` (0x1000) in the config forces the per-scan loop, `PTE_KERNEL_SCAN_LOOP_ONE_CHAIN`
# (0x2000) the one-chain-per-workgroup form of the one-kernel loop.
scan_loop_name(r::DeviceReplicas) = unsafe_string(ccall((:pte_scan_loop_name, libpte), Cstring, (Ptr{Cvoid},), r.handle))
# Round 6: the one kernel cannot hang (residency gate inside the launch, fallback to the launch-per-scan loop: include/pte.h, pte_scan_loop_stats).
# (fused_calls, gate_aborts, poisoned): calls that ran as one launch; launches that found a workgroup missing and fell back (same results); whether
# a failure inside the kernel has poisoned the handle -- then only `pte_set_state` (a checkpoint's state, chain, rng) or `pte_destroy` are accepted.
function scan_loop_stats(r::DeviceReplicas)
    f = Ref{Int64}(0); a = Ref{Int64}(0); p = Ref{Int32}(0)
    check(r, ccall((:pte_scan_loop_stats, libpte), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int32}), r.handle, f, a, p))
    return (fused_calls = f[], gate_aborts = a[], poisoned = p[] != 0)
end

# adapt(pt, reduced_recorders) ran on the host (adapt_tempering: src/tempering/NonReversiblePT.jl:46-66, StabilizedPT.jl:53-65;
# adapt_explorer: AutoMALA.jl:70-79, MALA.jl:63-69, Compose.jl:10-14; update_reference!: GaussianReference.jl:24-31): push the results
function after_adapt!(pt)
    r = pt.replicas::DeviceReplicas
    tempering = pt.shared.tempering
    betas = tempering isa Pigeons.StabilizedPT ?
        vcat(tempering.variational_leg.schedule.grids, reverse(tempering.fixed_leg.schedule.grids)) :   # concatenate_log_potentials order
        tempering.schedule.grids
    check(r, ccall((:pte_set_schedule, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), r.handle, betas, length(betas)))
    for ex in explorers(pt.shared.explorer)
        ex isa Union{AutoMALA, MALA} || continue
        std = ex.estimated_target_std_deviations
        check(r, ccall((:pte_set_explorer_adaptation, libpte), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int64), r.handle, ex.step_size,
                       std === nothing ? C_NULL : pointer(std), std === nothing ? 0 : length(std)))
    end
    v = pt.inputs.variational
    if v isa Pigeons.GaussianReference && !isempty(v.mean)
        m = v.mean[:singleton_variable]; s = v.standard_deviation[:singleton_variable]
        uses = Int32[c <= pt.inputs.n_chains_variational || pt.inputs.n_chains_variational == 0 ? 1 : 0 for c in 1:r.n_chains]
        check(r, ccall((:pte_set_variational_reference, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}),
                       r.handle, m, s, length(m), uses))
    end
end
explorers(e::Compose) = (e.first, e.second)
explorers(e) = (e,)

# ---- reduce_recorders!(pt, replicas)  (src/recorders/recorders.jl:88-130): rebuild every recorder from the flat arrays ----------
# Keys are 1-based chains / replica indices on this side.  The device keeps per-pair / per-chain sums where the reference merges
# per-replica OnlineStats in a tree (src/mpi_utils/Entangler.jl:188-251): integers are identical, floats agree to 1e-9.
function Pigeons.reduce_recorders!(pt, r::DeviceReplicas)
    check(r, ccall((:pte_reduce, libpte), Cint, (Ptr{Cvoid},), r.handle))
    rec = Pigeons.create_recorders(pt.inputs, pt.shared)
    N, K, c0 = r.n_chains, r.n_local, r.first_chain
    has(name) = haskey(rec, name)
    # swap_acceptance_pr: GroupBy(Tuple{Int,Int}, Mean()); log_sum_ratio: GroupBy(Tuple{Int,Int}, LogSum()) (recorder.jl:54,84)
    np = r.world_size == 1 ? N - 1 : K
    mean = zeros(np); n = zeros(Int64, np)
    check(r, ccall((:pte_get_swap_acceptance, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), r.handle, mean, n))
    up = zeros(np); dn = zeros(np); un = zeros(Int64, np); dnn = zeros(Int64, np)
    check(r, ccall((:pte_get_log_sum_ratio, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}), r.handle, up, un, dn, dnn))
    for i in 1:np
        c = c0 + i                                       # lower chain of the pair, 1-based
        if has(:swap_acceptance_pr) && n[i] > 0
            put!(rec.swap_acceptance_pr, (c, c + 1), Mean(mean[i], EqualWeight(), n[i]), n[i])
        end
        if has(:log_sum_ratio) && un[i] > 0
            put!(rec.log_sum_ratio, (c, c + 1), LogSum(up[i], un[i]), un[i])
            put!(rec.log_sum_ratio, (c + 1, c), LogSum(dn[i], dnn[i]), dnn[i])
        end
    end
    # round_trip (RoundTripRecorder.jl:4-19): the per-replica state machines live on the device, the counts are already summed
    if has(:round_trip)
        a = Ref{Int64}(0); b = Ref{Int64}(0)
        check(r, ccall((:pte_get_round_trip, libpte), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), r.handle, a, b))
        rec.round_trip.n_tempered_restarts = a[]; rec.round_trip.n_round_trips = b[]
    end
    # index_process: Dict{Int, Vector{Int}}, replica index => chain at every scan (recorder.jl:81; swap.jl:115)
    if has(:index_process)
        S = Pigeons.n_scans_in_round(pt.shared.iterators)
        ip = zeros(Int64, S, N); ns = Ref{Int64}(0)      # C layout [replica][scan] == column-major (scan, replica)
        check(r, ccall((:pte_get_index_process, libpte), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), r.handle, ip, ns))
        for i in 1:N
            rec.index_process[i] = ip[1:ns[], i] .+ 1
        end
    end
    # explorer_acceptance_pr: GroupBy(Int, Mean()); explorer_n_steps: GroupBy(Int, Sum()) (recorder.jl:61,68)
    am = zeros(K); an = zeros(Int64, K); ss = zeros(K); sn = zeros(Int64, K)
    check(r, ccall((:pte_get_explorer_stats, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}), r.handle, am, an, ss, sn))
    for i in 1:K
        has(:explorer_acceptance_pr) && an[i] > 0 && put!(rec.explorer_acceptance_pr, c0 + i, Mean(am[i], EqualWeight(), an[i]), an[i])
        has(:explorer_n_steps) && sn[i] > 0 && put!(rec.explorer_n_steps, c0 + i, Sum(ss[i], sn[i]), sn[i])
    end
    # am_factors, reversibility_rate: GroupBy(Int, Mean()) (AutoMALA.jl:277,294)
    if has(:am_factors) || has(:reversibility_rate)
        fm = zeros(K); fn = zeros(Int64, K); rm = zeros(K); rn = zeros(Int64, K)
        check(r, ccall((:pte_get_automala_stats, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}), r.handle, fm, fn, rm, rn))
        for i in 1:K
            has(:am_factors) && fn[i] > 0 && put!(rec.am_factors, c0 + i, Mean(fm[i], EqualWeight(), fn[i]), fn[i])
            has(:reversibility_rate) && rn[i] > 0 && put!(rec.reversibility_rate, c0 + i, Mean(rm[i], EqualWeight(), rn[i]), rn[i])
        end
    end
    # online / _transformed_online: OnlineStateRecorder, stats[(:singleton_variable => Mean / Variance)] = Group of d + 1 stats of
    # extract_sample(state, lp) = [state; lp]  (OnlineStateRecorder.jl:95-110, src/pt/state.jl:79)
    for name in (:online, :_transformed_online)
        has(name) || continue
        m = zeros(r.dim); v = zeros(r.dim); cnt = Ref{Int64}(0); lm = Ref{Float64}(0.0); lv = Ref{Float64}(0.0)
        check(r, ccall((:pte_get_online, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}), r.handle, m, v, cnt))
        cnt[] > 0 || continue
        check(r, ccall((:pte_get_online_log_density, libpte), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), r.handle, lm, lv))
        ms = name === :online ? vcat(m, lm[]) : m; vs = name === :online ? vcat(v, lv[]) : v
        stats = getproperty(rec, name).stats
        stats[Pair(:singleton_variable, Mean)] = Group([Mean(ms[i], EqualWeight(), cnt[]) for i in eachindex(ms)])
        stats[Pair(:singleton_variable, Variance)] = Group([Variance(vs[i], ms[i], EqualWeight(), cnt[]) for i in eachindex(ms)])
    end
    # energy_ac1: GroupBy(Int, CovMatrix(2)) of (lp before, lp after) per chain (recorder.jl:113; pigeons.jl:134-143)
    if has(:energy_ac1)
        cor = zeros(K); en = zeros(Int64, K); mom = zeros(5, K)
        check(r, ccall((:pte_get_energy_ac1, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}), r.handle, cor, en, mom))
        for i in 1:K
            en[i] > 0 || continue
            mb, ma, cbb, cba, caa = mom[:, i]            # running means and co-moment sums (Welford)
            o = CovMatrix(2)
            o.b .= (mb, ma); o.n = en[i]                 # OnlineStatsBase 1.x: A = running mean of x x', b = running mean
            o.A .= [cbb / en[i] + mb^2  cba / en[i] + mb * ma; cba / en[i] + mb * ma  caa / en[i] + ma^2]
            put!(rec.energy_ac1, c0 + i, o, en[i])
        end
    end
    # traces: Dict{Pair{Int,Int}, Any}, (chain => scan) => [state; lp] (recorder.jl:27,39-43)
    if has(:traces)
        rows = (r.record_flags & RECORD_TRACES_EXTENDED) != 0 ? K : (pt.inputs.n_chains_variational > 0 ? 2 : 1)
        S = Pigeons.n_scans_in_round(pt.shared.iterators)
        buf = zeros(r.dim + 1, rows, S); ns = Ref{Int64}(0)
        check(r, ccall((:pte_get_traces, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}), r.handle, buf, ns))
        chains = rows == K ? collect((c0 + 1):(c0 + K)) : target_chains(pt)
        for s in 1:ns[], (j, c) in enumerate(chains)
            rec.traces[c => s] = buf[:, j, s]
        end
    end
    # what the host itself recorded this round (timing_extrema, allocation_extrema): merge, then reset (recorders.jl:121-125)
    for name in (:timing_extrema, :allocation_extrema)
        has(name) && haskey(r.host_recorders, name) || continue
        rec = merge(rec, NamedTuple{(name,)}((merge(getproperty(rec, name), getproperty(r.host_recorders, name)),)))
        empty!(getproperty(r.host_recorders, name))
    end
    return rec
end
# a GroupBy{T,S} keeps `value::OrderedDict{T,S}` and the total count `n`
function put!(g::GroupBy, key, stat, n)
    g.value[key] = stat
    g.n += n
end
target_chains(pt) = [c for c in 1:Pigeons.n_chains(pt.inputs) if Pigeons.is_target(pt.shared.tempering.swap_graphs, c)]

# ---- distributed runs: one Julia process per GPU (the reference's MPI layout, docs/src/distributed.md), chains sharded ----------
# The reference's transport (src/mpi_utils/Entangler.jl:118-180 `transmit!`, swap! over EntangledReplicas src/swap/swap.jl:79-102)
# is replaced by RCCL send/recv that libpte enqueues itself; Julia only hands the 128-byte communicator id around once:
#
#     id = zeros(UInt8, 128)
#     MPI.Comm_rank(comm) == 0 && ccall((:pte_comm_unique_id, libpte), Cint, (Ptr{UInt8},), id)
#     MPI.Bcast!(id, 0, comm)
#     pt = PT(Inputs(target = on_mi355x(toy_mvn_target(4096); device = local_rank, rank = MPI.Comm_rank(comm), world_size = MPI.Comm_size(comm)), ...))
#     comm_init!(pt.replicas, id)                        # collective: ncclCommInitRank inside libpte
#
# and from then on run_one_round! above is unchanged: pte_run_scans on a sharded engine performs explore, the two swap phases and
# the boundary exchange (one {SwapStat, payload} message per active side and scan) on the engine's HIP stream.  The per-rank
# recorder slices (keyed by chain / pair) are concatenated in rank order -- that IS the deterministic reduction of
# all_reduce_deterministically -- with `allgather` below or MPI.Allgatherv.
comm_init!(r::DeviceReplicas, id::Vector{UInt8}) =
    check(r, ccall((:pte_comm_init, libpte), Cint, (Ptr{Cvoid}, Ptr{UInt8}), r.handle, id))
function allgather(r::DeviceReplicas, mine::Vector{UInt8})
    out = zeros(UInt8, length(mine) * r.world_size)
    check(r, ccall((:pte_comm_allgather, libpte), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64, Ptr{UInt8}), r.handle, mine, length(mine), out))
    return out
end
barrier(r::DeviceReplicas) = check(r, ccall((:pte_comm_barrier, libpte), Cint, (Ptr{Cvoid},), r.handle))

# One Julia process driving G GPUs (no MPI at all): G engines with rank = g - 1, world_size = G, device = g - 1
function run_scans_group!(rs::Vector{DeviceReplicas}, first_scan, n_scans)
    hs = [r.handle for r in rs]
    check(rs[1], ccall((:pte_group_run_scans, libpte), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int64, Int64), hs, length(hs), first_scan, n_scans))
end

end # module
