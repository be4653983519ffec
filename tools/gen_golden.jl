# gen_golden.jl -- INERT in the build image (no Julia).  ONE command on any machine with Julia >= 1.8, Pigeons.jl and a
# checkout of this repo:
#
#     julia --project=<env with Pigeons, SplittableRandoms, Distributions, LogDensityProblems, ForwardDiff> \
#           tools/gen_golden.jl > tests/golden/reference_pigeons.json && python tools/import_tables.py
#
# It records what the CPU oracle and the HIP engine are compared against bit for bit (SURVEY.md 8(c), tier 3):
#   * Julia's own ziggurat tables Random.ki / wi / fi / ke / we / fe, all 6 x 256 entries as bit patterns
#     (tools/import_tables.py installs them into pigeons.jl_amd/csrc/zig_tables.h; tests/oracle.py installs them into the oracle);
#   * raw streams of a replica's SplittableRandom: 4096 rand / randn / randexp per seed (several slow-path and tail draws each)
#     and 1024 rand(rng, Bool) -- these decide the conventions named in include/pte_rng_policy.h;
#   * rand(rng, a:b) on Int64 ranges and SliceSampler steps on Integer / Bool / mixed states (SliceSampler.jl:65-86, 136-142, 189), which only
#     the oracle restates (po_rand_range, po_slice_step_mixed): tests/test_oracle_slice_mixed.py::test_*_against_live_reference;
#   * seeded runs of the real reference: C1 with SliceSampler / ToyExplorer, AutoMALA on the MVN path, a two-leg TestSwapper
#     run, a C5-shaped Ising run (examples/ising.jl, base_length 8) and a C3-shaped funnel run with AutoMALA.
# Once the file is committed, tests/test_golden.py::test_*_against_live_reference and tests/test_zig_tables.py stop skipping
# and either go green or name the first differing table entry / draw.
using Pigeons, SplittableRandoms, Random

bits(x::Float64) = string(reinterpret(UInt64, x))          # exact, as a decimal string
bits(x::UInt64) = string(x)
bits(v::AbstractVector) = [bits(x) for x in v]

tables() = Dict("ki" => bits(collect(UInt64, Random.ki)), "wi" => bits(collect(Float64, Random.wi)), "fi" => bits(collect(Float64, Random.fi)),
                "ke" => bits(collect(UInt64, Random.ke)), "we" => bits(collect(Float64, Random.we)), "fe" => bits(collect(Float64, Random.fe)),
                "ziggurat_nor_r" => bits(Float64(Random.ziggurat_nor_r)), "ziggurat_exp_r" => bits(Float64(Random.ziggurat_exp_r)))

function rng_streams(seed; n = 4096, nbool = 1024)
    master = SplittableRandom(seed)
    r = split(master)                                       # replica 1's stream (src/utils/misc.jl:21-31)
    a = deepcopy(r); b = deepcopy(r); c = deepcopy(r); d = deepcopy(r); e = deepcopy(r)
    Dict("seed" => seed,
         "stream" => [string(r.seed), string(r.gamma)],
         "u64" => [string(rand(e, UInt64)) for _ in 1:16],
         "rand" => bits([rand(a) for _ in 1:n]),
         "randn" => bits([randn(b) for _ in 1:n]),
         "randexp" => bits([randexp(c) for _ in 1:n]),
         "rand_bool" => [rand(d, Bool) ? 1 : 0 for _ in 1:nbool])
end

function run(; state_of = r -> r.state, kwargs...)
    pt = pigeons(; seed = 1, show_report = false,
                 record = [index_process, round_trip, swap_acceptance_pr, log_sum_ratio, Pigeons.explorer_n_steps], kwargs...)
    n = Pigeons.n_chains(pt.inputs)
    ip = pt.reduced_recorders.index_process                 # Dict replica -> Vector{chain} of the last round
    sw = Pigeons.value(pt.reduced_recorders.swap_acceptance_pr)
    reps = Pigeons.locals(pt.replicas)
    Dict("n_chains" => n,
         "index_process_last_round" => [ip[i] for i in 1:n],           # 1-based chains
         "schedule" => bits(pt.shared.tempering.schedule.grids),
         "swap_acceptance_mean" => bits([Pigeons.value(sw[(i, i + 1)]) for i in 1:(n - 1)]),
         "stepping_stone_pair" => bits(collect(Float64, Pigeons.stepping_stone_pair(pt))),
         "round_trip" => [Pigeons.n_tempered_restarts(pt), Pigeons.n_round_trips(pt)],
         "final_states" => [bits(collect(Float64, vec(state_of(r)))) for r in reps],
         "final_chains" => [r.chain for r in reps],
         "final_rng" => [[string(r.rng.seed), string(r.rng.gamma)] for r in reps])
end

out = Dict{String,Any}(
    "pigeons_version" => string(pkgversion(Pigeons)), "julia_version" => string(VERSION),
    "tables" => tables(),
    "rng" => [rng_streams(s) for s in 1:3],
    "c1_slice" => run(target = toy_mvn_target(2), n_chains = 10, n_rounds = 5, explorer = SliceSampler()),
    "c1_toy" => run(target = toy_mvn_target(2), n_chains = 10, n_rounds = 5),
    "mvn10_automala" => run(target = toy_mvn_target(10), n_chains = 6, n_rounds = 6, explorer = AutoMALA()),
    "test_swapper_two_legs" => run(target = Pigeons.TestSwapper(0.5), n_chains = 5, n_chains_variational = 5, n_rounds = 8),
)

# C5-shaped: the Ising example of the reference (examples/ising.jl), base_length 8, IsingMetropolis(3)
try
    include(joinpath(pkgdir(Pigeons), "examples", "ising.jl"))
    out["ising8"] = run(target = IsingLogPotential(1.0, 8), n_chains = 6, n_rounds = 5,
                        state_of = r -> Float64.(permutedims(r.state.matrix)))      # row-major matrix[i,j] -> state[i*L + j]
catch err
    out["ising8_error"] = sprint(showerror, err)
end

# C3-shaped: Neal's funnel (test/supporting/dimensional-analysis.jl:33-48) from a normal reference of precision 1/9, AutoMALA
try
    @eval using Distributions, LogDensityProblems, ForwardDiff
    @eval begin
        struct GoldenFunnel; dim::Int; end
        (p::GoldenFunnel)(x) = LogDensityProblems.logdensity(p, x)
        LogDensityProblems.dimension(p::GoldenFunnel) = p.dim
        LogDensityProblems.capabilities(::Type{GoldenFunnel}) = LogDensityProblems.LogDensityOrder{0}()
        function LogDensityProblems.logdensity(p::GoldenFunnel, z)
            s = 0.0
            y = z[1]
            s += logpdf(Normal(0.0, 3.0), y)
            sigma = exp(y / 2.0)
            for i in 2:p.dim
                s += logpdf(Normal(0.0, sigma), z[i])
            end
            return s
        end
        Pigeons.initialization(p::GoldenFunnel, ::AbstractRNG, ::Int64) = zeros(p.dim)
    end
    out["funnel8_automala"] = Base.invokelatest(() -> run(target = GoldenFunnel(8),
        reference = Pigeons.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 8), n_chains = 6, n_rounds = 6, explorer = AutoMALA()))
catch err
    out["funnel8_error"] = sprint(showerror, err)
end

# rand(rng, a:b) (Random.SamplerRangeNDL over rand(rng, UInt64)) and SliceSampler's Integer / Bool / mixed-state methods: oracle only
const GOLDEN_RANGES = [(0, 10), (-5, 5), (0, 1), (3, 1023), (-(2^62), 2^62), (typemin(Int64), typemax(Int64)), (7, 7 + 2^62 + 2^61), (-17, 2^33 + 5)]
function range_streams(seed; n = 256)
    r = split(SplittableRandom(seed))
    Dict("seed" => seed, "ranges" => [[string(a), string(b)] for (a, b) in GOLDEN_RANGES],
         "draws" => [[string(rand(r, a:b)) for _ in 1:n] for (a, b) in GOLDEN_RANGES],          # ONE stream, range after range
         "final_rng" => [string(r.seed), string(r.gamma)])
end
golden_lp_int(x)   = -0.125 * ((x[1] - 3)^2 + (x[2] + 2)^2)
golden_lp_bool(x)  = (x[1] ? 0.5 : 0.0) - (x[2] ? 1.25 : 0.0) + ((x[1] && x[3]) ? 0.75 : 0.0)
golden_lp_mixed(x) = -0.5 * x[1]^2 - 0.125 * (x[2] - 3)^2 + (x[3] ? 0.75 : 0.0)
function slice_mixed(state, lp; steps = 64, seed = 1)
    rng = split(SplittableRandom(seed))
    replica = Pigeons.Replica(state, 1, rng, (;), 1)
    h = SliceSampler()
    trace = Vector{Vector{String}}()
    for _ in 1:steps
        cached = -Inf
        for _ in 1:h.n_passes                                  # step!(::SliceSampler), SliceSampler.jl:24-30
            cached = Pigeons.slice_sample!(h, replica.state, lp, cached, replica)
        end
        push!(trace, bits(Float64[Float64(v) for v in replica.state]))
    end
    Dict("seed" => seed, "states" => trace, "final_rng" => [string(rng.seed), string(rng.gamma)])
end
try
    out["rand_range"] = [range_streams(s) for s in 1:3]
    out["slice_integer"] = slice_mixed([0, 0], golden_lp_int)
    out["slice_bool"] = slice_mixed([false, true, false], golden_lp_bool)
    out["slice_mixed"] = slice_mixed(Real[0.25, 3, true], golden_lp_mixed)
catch err
    out["slice_mixed_error"] = sprint(showerror, err)
end

# minimal JSON writer (no extra dependency)
json(x::AbstractString) = "\"" * replace(x, "\\" => "\\\\", "\"" => "\\\"", "\n" => "\\n") * "\""
json(x::Bool) = x ? "true" : "false"
json(x::Number) = string(x)
json(x::AbstractVector) = "[" * join(json.(x), ",") * "]"
json(x::AbstractDict) = "{" * join([json(string(k)) * ":" * json(v) for (k, v) in x], ",") * "}"
println(json(out))
