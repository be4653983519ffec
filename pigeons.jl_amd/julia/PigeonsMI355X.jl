# PigeonsMI355X.jl -- the reference-side binding a Pigeons.jl maintainer would add.
#
# INERT IN THE BUILD IMAGE (no Julia toolchain; never executed there).  It shows how the three
# dispatch hooks that bracket the hot path, plus construction / adaptation, bind to the C ABI of
# include/pte.h.  Conventions follow the reference's only FFI precedent,
# ext/PigeonsBridgeStanExt/interface.jl:118-183 (Cint return code, message fetched on error).
module PigeonsMI355X

using Pigeons
using Pigeons: Inputs, Shared, Replica, PT, SliceSampler, ToyExplorer, ScaledPrecisionNormalPath

const libpte = "libpte.so"

# mirror of `pte_config` (include/pte.h) -- field order and types must match
Base.@kwdef mutable struct PteConfig
    struct_size::UInt32 = 0
    abi_version::UInt32 = 2
    device::Int32 = 0
    target::Int32 = 0
    explorer::Int32 = 1
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
    explorer2::Int32 = 0        # Compose(explorer, explorer2)
    n_chains_variational::Int64 = 0   # StabilizedPT with variational == nothing
    debug_kernel::Int32 = 0     # PTE_KERNEL_*: 0 = the default kernel of the explorer (never read from the environment)
    reserved0::Int32 = 0
end

"""Device-resident `replicas` (informal interface src/replicas/replicas.jl:11-40)."""
mutable struct DeviceReplicas
    handle::Ptr{Cvoid}
    n_chains::Int
    dim::Int
end

function check(r::DeviceReplicas, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:pte_last_error, libpte), Cstring, (Ptr{Cvoid},), r.handle))
    error(msg)    # same wording class as the reference's exceptions (SliceSampler.jl:52-60,179-185)
end

# create_replicas(inputs, shared, source)  (src/replicas/replicas.jl:65-98)
function Pigeons.create_replicas(inputs::Inputs{<:ScaledPrecisionNormalPath}, shared::Shared, ::Val{:mi355x})
    cfg = PteConfig(n_chains = inputs.n_chains, dim = inputs.target.dim, seed = inputs.seed,
                    max_scans_per_round = 2^inputs.n_rounds,
                    target_params = (inputs.target.precision0, inputs.target.precision1, 0.0, 0.0))
    cfg.struct_size = sizeof(PteConfig)
    ex = shared.explorer
    if ex isa SliceSampler
        cfg.explorer = 2; cfg.slice_w = ex.w; cfg.slice_p = ex.p
        cfg.slice_n_passes = ex.n_passes; cfg.slice_max_iter = ex.max_iter
    elseif ex isa ToyExplorer
        cfg.explorer = 1
    else
        error("explorer $(typeof(ex)) has no device kernel; use the CPU path")
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:pte_create, libpte), Cint, (Ref{PteConfig}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || error(unsafe_string(ccall((:pte_last_error, libpte), Cstring, (Ptr{Cvoid},), C_NULL)))
    r = DeviceReplicas(h[], inputs.n_chains, inputs.target.dim)
    finalizer(x -> ccall((:pte_destroy, libpte), Cint, (Ptr{Cvoid},), x.handle), r)
    return r
end

# explore!(pt, explorer, ::Val)  (src/pt/pigeons.jl:82-97): one ccall for ALL local replicas
Pigeons.explore!(pt::PT{<:Any,DeviceReplicas}, explorer, ::Val) =
    check(pt.replicas, ccall((:pte_explore, libpte), Cint, (Ptr{Cvoid}, Int64), pt.replicas.handle, pt.shared.iterators.scan))

# swap!(pair_swapper, replicas, swap_graph)  (src/swap/swap.jl:6): graph parity = iseven(scan)
Pigeons.swap!(pair_swapper, r::DeviceReplicas, swap_graph::Pigeons.OddEven) =
    check(r, ccall((:pte_swap, libpte), Cint, (Ptr{Cvoid}, Int64), r.handle, swap_graph.even ? 2 : 1))

# run_one_round!: the fused `while next_scan!` loop (src/pt/pigeons.jl:46-55)
function Pigeons.run_one_round!(pt::PT{<:Any,DeviceReplicas})
    n = Pigeons.n_scans_in_round(pt.shared.iterators)
    timed = @timed check(pt.replicas, ccall((:pte_run_scans, libpte), Cint, (Ptr{Cvoid}, Int64, Int64), pt.replicas.handle, 1, n))
    pt.shared.iterators.scan = 0
    return reduce_recorders!(pt, pt.replicas, timed)
end

# reduce_recorders!(pt, replicas)  (src/recorders/recorders.jl:88-120): rebuild the GroupBy recorders
function reduce_recorders!(pt, r::DeviceReplicas, timed)
    check(r, ccall((:pte_reduce, libpte), Cint, (Ptr{Cvoid},), r.handle))
    N = r.n_chains
    mean = zeros(N - 1); n = zeros(Int64, N - 1)
    check(r, ccall((:pte_get_swap_acceptance, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), r.handle, mean, n))
    up = zeros(N - 1); dn = zeros(N - 1); un = zeros(Int64, N - 1); dnn = zeros(Int64, N - 1)
    check(r, ccall((:pte_get_log_sum_ratio, libpte), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}), r.handle, up, un, dn, dnn))
    recorders = Pigeons.create_recorders(pt.inputs, pt.shared)
    for i in 1:(N-1)                       # 1-based chains on the Julia side
        n[i] > 0 || continue
        recorders.swap_acceptance_pr.value[(i, i + 1)] = Pigeons.Mean(mean[i], Pigeons.EqualWeight(), n[i])
        recorders.log_sum_ratio.value[(i, i + 1)] = Pigeons.LogSum(up[i], un[i])
        recorders.log_sum_ratio.value[(i + 1, i)] = Pigeons.LogSum(dn[i], dnn[i])
    end
    # round_trip, index_process, explorer_* are filled the same way from pte_get_round_trip,
    # pte_get_index_process (+1 for 1-based chains), pte_get_explorer_stats.
    Pigeons.record_timed_if_requested!(recorders, :round, timed)
    return recorders
end

# ---- distributed runs: one Julia process per GPU (the reference's MPI layout, docs/src/distributed.md), chains sharded ----
# The reference's transport (src/mpi_utils/Entangler.jl:118-180 `transmit!`, swap! over EntangledReplicas src/swap/swap.jl:79-102)
# is replaced by RCCL send/recv that libpte enqueues itself; Julia only hands the 128-byte communicator id around once.
# With MPI.jl (what Pigeons already depends on for its distributed mode):
#
#     id = zeros(UInt8, 128)
#     MPI.Comm_rank(comm) == 0 && check0(ccall((:pte_comm_unique_id, libpte), Cint, (Ptr{UInt8},), id))
#     MPI.Bcast!(id, 0, comm)
#     cfg.rank = MPI.Comm_rank(comm); cfg.world_size = MPI.Comm_size(comm); cfg.device = local_rank
#     r = create_replicas(...)                                   # pte_create: this rank's chains [rank*N/G, (rank+1)*N/G)
#     comm_init!(r, id)                                          # collective: ncclCommInitRank inside libpte
#
# and from then on run_one_round! above is unchanged: pte_run_scans on a sharded engine performs explore, the two swap
# phases and the boundary exchange (one {SwapStat, payload} message per active side and scan) on the engine's HIP stream.
comm_init!(r::DeviceReplicas, id::Vector{UInt8}) =
    check(r, ccall((:pte_comm_init, libpte), Cint, (Ptr{Cvoid}, Ptr{UInt8}), r.handle, id))

# reduce_recorders!(pt, ::EntangledReplicas) (src/recorders/recorders.jl:86): recorders are keyed by chain / pair, so the
# reduction is a concatenation in rank order -- pte_comm_allgather moves the per-rank slices without MPI
function allgather(r::DeviceReplicas, mine::Vector{UInt8}, world::Int)
    out = zeros(UInt8, length(mine) * world)
    check(r, ccall((:pte_comm_allgather, libpte), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64, Ptr{UInt8}), r.handle, mine, length(mine), out))
    return out
end
barrier(r::DeviceReplicas) = check(r, ccall((:pte_comm_barrier, libpte), Cint, (Ptr{Cvoid},), r.handle))

# One Julia process driving G GPUs (no MPI at all): G engines with cfg.rank = g - 1, cfg.world_size = G, cfg.device = g - 1
function run_scans_group!(rs::Vector{DeviceReplicas}, first_scan, n_scans)
    hs = [r.handle for r in rs]
    check(rs[1], ccall((:pte_group_run_scans, libpte), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int64, Int64), hs, length(hs), first_scan, n_scans))
end

# adapt_tempering -> new Schedule -> discretize on the device (src/tempering/NonReversiblePT.jl:46-66)
set_schedule!(r::DeviceReplicas, schedule::Pigeons.Schedule) =
    check(r, ccall((:pte_set_schedule, libpte), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), r.handle, schedule.grids, length(schedule.grids)))

# record = [...] symbols -> pte_config.record_flags (include/pte.h PTE_RECORD_*)
record_flags(names) = UInt32(sum((
    (:round_trip in names) * 1, (:index_process in names) * 2, (:online in names || :_transformed_online in names) * 4,
    (:traces in names) * 8, (:energy_ac1 in names) * 16)))

# Compose(first, second)  (src/explorers/Compose.jl:5-19) -> pte_config.explorer / explorer2
explorer_code(::SliceSampler) = Int32(2)
explorer_code(::Pigeons.AutoMALA) = Int32(3)
explorer_code(::Pigeons.MALA) = Int32(5)
explorer_codes(e::Pigeons.Compose) = (explorer_code(e.first), explorer_code(e.second))
explorer_codes(e) = (explorer_code(e), Int32(0))

# update_reference!(reduced_recorders, ::GaussianReference, state)  (src/variational/GaussianReference.jl:24-31)
# followed by update_path_variational (src/variational/variational.jl:36-41): the variational leg's chains start at it
function update_reference_on_device!(r::DeviceReplicas, variational::Pigeons.GaussianReference, uses::Vector{Int32})
    m = variational.mean[:singleton_variable]; s = variational.standard_deviation[:singleton_variable]
    check(r, ccall((:pte_set_variational_reference, libpte), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}), r.handle, m, s, length(m), uses))
end

end # module
