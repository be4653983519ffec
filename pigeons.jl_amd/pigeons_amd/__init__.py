"""pigeons_amd: host-side mirror of the Pigeons.jl surface for the explore-then-swap hot path,
driving the MI355X-native engine (libpte.so, C ABI in include/pte.h) through ctypes.

No Julia toolchain exists in the build image, so this Python mirror is the runnable host side;
the Julia `ccall` glue a Pigeons.jl maintainer would add is in INTEGRATION.md and
pigeons.jl_amd/julia/PigeonsMI355X.jl.
"""
from ._lib import PteError, LIB_PATH
from .engine import Engine
from .pt import (Inputs, PT, pigeons, toy_mvn_target, ScaledPrecisionNormalPath, TestSwapper,
                 SliceSampler, ToyExplorer, AutoMALA, MALA, Compose, GaussianReference, InterpolatingPath, Funnel, IsingLogPotential, IsingMetropolis, ScaledPrecisionNormalLogPotential,
                 IdentityPreconditioner, DiagonalPreconditioner, MixDiagonalPreconditioner,
                 record_default, record_online,
                 log_sum_ratio, swap_acceptance_pr, round_trip, index_process, online, traces, energy_ac1,
                 energy_ac1s, sample_array, sample_names, get_sample, mean, var,
                 timing_extrema, allocation_extrema, explorer_acceptance_pr, explorer_n_steps,
                 stepping_stone, stepping_stone_pair, n_round_trips, n_tempered_restarts,
                 global_barrier, global_barrier_variational, target_chains, StabilizedPT, last_round_max_time, report, analytic_lognormalization,
                 analytic_cumulativebarrier, run_one_round, adapt, next_round, n_scans_in_round)
from .checkpoint import write_checkpoint, load_checkpoint, latest_checkpoint_folder, increment_n_rounds
from .tempering import (Schedule, equally_spaced_schedule, optimal_schedule,
                        FritschCarlsonMonotonicInterpolation, CommunicationBarriers, rejections)
