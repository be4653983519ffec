"""Regenerates tests/golden/*.json.

Two kinds of fixtures, kept apart:
  * kat_reference.json  -- known answers that do NOT come from this repo: the public SplitMix64 vector (== Java
    SplittableRandom.nextLong, which SplittableRandoms.jl reproduces), the four ziggurat table entries of Julia's
    Random recalled in SURVEY.md App. B, and answers the reference's own tests state in closed form
    (test/test_round_trips.jl:1-14; the DEO index process with every swap accepted is RNG-free, src/swap/DEO.jl:12,
    src/swap/OddEven.jl:23-31).  The generator only re-derives the closed forms; it does not call the oracle for them.
  * oracle_runs.json    -- seeded runs of the CPU oracle (oracle/pt_oracle.c) on small hot-path configs.  The reference
    cannot run in the build image (no Julia), so these are REGRESSION vectors of the restated algorithm ("parity
    unpinned"): they pin the oracle against drift and give the GPU tests a committed target that does not depend on
    rebuilding the oracle.
Run:  python tools/gen_golden.py
"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle as O

GOLD = os.path.join(ROOT, "tests", "golden")


def deo_all_accept_index_process(n_chains, n_rounds):
    """chain of every replica at every scan of the LAST round when all proposed swaps are accepted (TestSwapper(1.0))."""
    chain = list(range(n_chains))                       # chain[replica], 0-based
    last = None
    for r in range(1, n_rounds + 1):
        rows = []
        for scan in range(1, 2 ** r + 1):
            rows.append(list(chain))                    # index_process is recorded in swap!, before the exchange
            even = (scan % 2 == 0)
            new = list(chain)
            for rep, c in enumerate(chain):
                chain_even = ((c + 1) % 2 == 0)
                proposed = (c + 1) + (1 if chain_even == even else -1)
                partner = 0 if proposed == 0 else (n_chains - 1 if proposed == n_chains + 1 else proposed - 1)
                new[rep] = partner
            chain = new
        last = rows
    return np.array(last).T.tolist()                    # [replica][scan]


def kat_reference():
    n_chains, n_rounds = 4, 5
    return {
        "_sources": {
            "splitmix64": "public SplitMix64 / java.util.SplittableRandom vector, seed 1234567, golden gamma",
            "ziggurat_pins": "Julia Random ki[1], wi[1], ke[1], we[1] (SURVEY.md App. B)",
            "test_swapper": "reference test/test_round_trips.jl:1-14; src/swap/DEO.jl:12; src/swap/OddEven.jl:23-31",
        },
        "splitmix64": {"seed": 1234567, "next_u64": [6457827717110365317, 3203168211198807973, 9817491932198370423,
                                                    4593380528125082431, 16408922859458223821]},
        "ziggurat_pins": {"ki0": "0x0007799ec012f7b2", "wi0": 1.7367254121602630e-15,
                          "ke0": "0x000e290a13924be3", "we0": 1.9311480126418366e-15},
        "test_swapper": {"n_chains": n_chains, "n_rounds": n_rounds,
                         "n_round_trips": sum(math.floor(max(2 ** n_rounds - i, 0) / n_chains / 2) for i in range(n_chains)),
                         "index_process_last_round": deo_all_accept_index_process(n_chains, n_rounds)},
    }


RUNS = {
    "slice_mvn": dict(n_chains=6, dim=10, seed=1, explorer=O.EXPLORER_SLICE, rounds=5),
    "slice_mvn_ragged": dict(n_chains=4, dim=65, seed=3, explorer=O.EXPLORER_SLICE, rounds=4),
    "toy_mvn": dict(n_chains=7, dim=33, seed=2, explorer=O.EXPLORER_TOY, rounds=5),
    "automala_mvn": dict(n_chains=5, dim=12, seed=1, explorer=O.EXPLORER_AUTOMALA, am_preconditioner=2, rounds=6),
    "automala_funnel": dict(n_chains=6, dim=8, seed=1, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL, p0=1.0 / 9.0,
                            am_preconditioner=2, rounds=6),
    "mala_mvn": dict(n_chains=5, dim=10, seed=1, explorer=O.EXPLORER_MALA, am_step_size=0.25, am_preconditioner=2, rounds=6),
    "compose_slice_automala": dict(n_chains=4, dim=1, seed=1, explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA,
                                   am_preconditioner=2, rounds=7),
    "two_leg_slice": dict(n_chains=5, n_chains_variational=4, dim=6, seed=1, explorer=O.EXPLORER_SLICE, rounds=6),
    "variational_funnel": dict(n_chains=5, n_chains_variational=4, dim=5, seed=2, explorer=O.EXPLORER_AUTOMALA, target=O.TARGET_FUNNEL,
                               p0=1.0 / 9.0, am_preconditioner=2, variational_first_tuning_round=3, rounds=6),
    "ising": dict(n_chains=6, dim=64, seed=2, explorer=O.EXPLORER_ISING, target=O.TARGET_ISING, p0=0.6, slice_n_passes=3, rounds=6),
}


def oracle_runs():
    out = {"_note": "oracle-generated regression vectors (parity vs a live Pigeons.jl is UNPINNED); see tools/gen_golden.py"}
    for name, kw in RUNS.items():
        kw = dict(kw); rounds = kw.pop("rounds")
        pt = O.OraclePT(record_energy_ac1=1, **kw)
        for _ in range(rounds):
            pt.run_round()
        x, chain, rng = pt.states()
        m, n = pt.swap_pr()
        cor, cn, raw = pt.energy_ac1()
        out[name] = {
            "config": {k: v for k, v in dict(RUNS[name]).items()},
            "index_process_last_round": pt.index_process().tolist(),
            "round_trip": list(pt.round_trip()),
            "chain": chain.tolist(),
            "rng": [[str(int(a)), str(int(b))] for a, b in rng],
            "swap_acceptance_mean": [float.hex(float(v)) for v in m],
            "swap_acceptance_n": n.tolist(),
            "schedule": [float.hex(float(v)) for v in pt.schedule()],
            "stepping_stone_pair": [float.hex(float(v)) for v in pt.stepping_stone_pair()],
            "state_first_row": [float.hex(float(v)) for v in x[0][:8]],
            "state_sum_abs": float.hex(float(np.abs(x).sum())),
            "energy_ac1_mean_after": [float.hex(float(v)) for v in raw[:, 1]],
        }
    return out


def cabi_c1():
    """BASELINE configs[0] (toy_mvn_target(2), n_chains=10, n_rounds=5, SliceSampler, seed 1) round by round, in a line format
    a plain C program reads with fscanf (tests/test_cabi.c): the schedule IN FORCE during each round (so that the C side needs no
    adaptation code), that round's swap acceptance and index process, and the final replicas."""
    N, d, R = 10, 2, 5
    pt = O.OraclePT(n_chains=N, dim=d, seed=1, explorer=O.EXPLORER_SLICE)
    hx = lambda a: " ".join(float.hex(float(v)) for v in np.ravel(a))
    it = lambda a: " ".join(str(int(v)) for v in np.ravel(a))
    lines = ["config %d %d 1 %d" % (N, d, R)]
    for r in range(1, R + 1):
        lines.append("round %d" % r)
        lines.append("schedule " + hx(pt.schedule()))
        pt.run_round()
        m, n = pt.swap_pr()
        lines.append("swap_mean " + hx(m))
        lines.append("swap_n " + it(n))
        lines.append("index_process " + it(pt.index_process()))          # [replica][scan]
    x, chain, rng = pt.states()
    lines += ["final_chain " + it(chain), "final_rng " + it(rng), "final_state " + hx(x), "end"]
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "cabi_c1.txt"), "w") as f:
        f.write(cabi_c1())
    with open(os.path.join(GOLD, "kat_reference.json"), "w") as f:
        json.dump(kat_reference(), f, indent=1)
    with open(os.path.join(GOLD, "oracle_runs.json"), "w") as f:
        json.dump(oracle_runs(), f, indent=1)
    print("wrote", sorted(os.listdir(GOLD)))
