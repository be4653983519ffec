# gen_golden.jl -- INERT in the build image (no Julia).  Run on any machine with Julia >= 1.8 and Pigeons.jl:
#
#     julia --project=. tools/gen_golden.jl > tests/golden/reference_pigeons.json
#
# It records what the CPU oracle and the HIP engine are compared against bit for bit (SURVEY.md 8(c), tier 3):
# raw RNG streams of the replicas' SplittableRandoms and a few seeded runs of the real reference.  Once the file is
# committed, tests/test_golden.py::test_*_against_live_reference stop skipping and the "parity unpinned" notes in
# oracle/pt_oracle.h, DESIGN.md and tests/golden can go.
using Pigeons, SplittableRandoms, Random

bits(x::Float64) = string(reinterpret(UInt64, x))          # exact, as a decimal string
bits(v::AbstractVector{Float64}) = [bits(x) for x in v]

function rng_streams(seed)
    master = SplittableRandom(seed)
    r = split(master)                                       # replica 1's stream (src/utils/misc.jl:21-31)
    a = deepcopy(r); b = deepcopy(r); c = deepcopy(r); d = deepcopy(r)
    Dict("seed" => seed,
         "rand" => bits([rand(a) for _ in 1:64]),
         "randn" => bits([randn(b) for _ in 1:64]),
         "randexp" => bits([randexp(c) for _ in 1:64]),
         "rand_bool" => [rand(d, Bool) for _ in 1:64])
end

function run(; kwargs...)
    pt = pigeons(; seed = 1, show_report = false,
                 record = [index_process, round_trip, swap_acceptance_pr, log_sum_ratio, Pigeons.explorer_n_steps], kwargs...)
    n = Pigeons.n_chains(pt.inputs)
    ip = pt.reduced_recorders.index_process                 # Dict replica -> Vector{chain} of the last round
    sw = Pigeons.value(pt.reduced_recorders.swap_acceptance_pr)
    Dict("n_chains" => n,
         "index_process_last_round" => [ip[i] for i in 1:n],           # 1-based chains
         "schedule" => bits(pt.shared.tempering.schedule.grids),
         "swap_acceptance_mean" => bits([Pigeons.value(sw[(i, i + 1)]) for i in 1:(n - 1)]),
         "stepping_stone_pair" => bits(collect(Pigeons.stepping_stone_pair(pt))),
         "round_trip" => [Pigeons.n_tempered_restarts(pt), Pigeons.n_round_trips(pt)],
         "final_states" => [bits(r.state) for r in Pigeons.locals(pt.replicas)],
         "final_chains" => [r.chain for r in Pigeons.locals(pt.replicas)])
end

out = Dict(
    "pigeons_version" => string(pkgversion(Pigeons)), "julia_version" => string(VERSION),
    "rng" => [rng_streams(s) for s in 1:3],
    "c1_slice" => run(target = toy_mvn_target(2), n_chains = 10, n_rounds = 5, explorer = SliceSampler()),
    "c1_toy" => run(target = toy_mvn_target(2), n_chains = 10, n_rounds = 5),
    "mvn10_automala" => run(target = toy_mvn_target(10), n_chains = 6, n_rounds = 6, explorer = AutoMALA()),
    "test_swapper_two_legs" => run(target = Pigeons.TestSwapper(0.5), n_chains = 5, n_chains_variational = 5, n_rounds = 8),
)

# minimal JSON writer (no extra dependency)
json(x::AbstractString) = "\"" * x * "\""
json(x::Bool) = x ? "true" : "false"
json(x::Number) = string(x)
json(x::AbstractVector) = "[" * join(json.(x), ",") * "]"
json(x::AbstractDict) = "{" * join([json(string(k)) * ":" * json(v) for (k, v) in x], ",") * "}"
println(json(out))
