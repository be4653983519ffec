"""The PRODUCTION multi-rank transport with a real peer: RcclShard -> pte_comm_init -> pte_run_scans (ncclGroupStart /
ncclSend / ncclRecv / ncclGroupEnd enqueued on the engine's stream) run by >= 2 fresh processes, bit-identical to the single
engine -- what the reference gets from 2 MPI processes in test/test_parallelism_invariance.jl:5-46 (src/swap/swap.jl:79-102).

The box has ONE GPU and RCCL refuses two ranks on one device, so the ranks load tests/fakerccl/libfakerccl.so through libpte's
documented override $PTE_RCCL_LIB: the 12 entry points libpte resolves, stream-ordered, over POSIX shm.  Everything above those
entry points is the shipped code path.  No torch.distributed anywhere: the 128-byte id travels through a file.
Also `bench.py --gpus 2 --same-device`: the driver's launcher, rank set-up, agreed transport, timing gather and teardown."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fakerccl")
FAKE = os.path.join(FAKE_DIR, "libfakerccl.so")


def build_fakerccl():
    src = os.path.join(FAKE_DIR, "fakerccl.cpp")
    if os.path.exists(FAKE) and os.path.getmtime(FAKE) >= os.path.getmtime(src):
        return FAKE
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-o", FAKE, src, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return FAKE


WORKER = r'''
import os, sys, json, time
import numpy as np
root, cfg, rank, world, idfile = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.path[:0] = [root, root + "/pigeons.jl_amd", root + "/tests"]
import pigeons_amd as P
from pigeons_amd.engine import comm_unique_id, comm_allow_library_override, comm_library
comm_allow_library_override(True)          # the stand-in is honoured only after this opt-in (a stale $PTE_RCCL_LIB alone is refused)
assert "fakerccl" in comm_library()[0], comm_library()

def explorer():
    k = cfg["explorer"]
    if k == "slice": return P.SliceSampler()
    if k == "toy": return P.ToyExplorer()
    if k == "automala": return P.AutoMALA()
    if k == "slice+automala": return P.Compose(P.SliceSampler(), P.AutoMALA())
    if k == "ising": return None
    raise SystemExit("unknown explorer " + k)

def mk():
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    if cfg["explorer"] == "ising":
        return P.Inputs(target=P.IsingLogPotential(cfg.get("beta", 0.6), cfg["L"]), n_chains=cfg["N"], n_rounds=cfg["rounds"],
                        seed=cfg.get("seed", 1), record=rec, show_report=False)
    return P.Inputs(target=P.toy_mvn_target(cfg["d"]), n_chains=cfg["N"], n_rounds=cfg["rounds"], explorer=explorer(),
                    seed=cfg.get("seed", 1), record=rec, show_report=False)

if rank == 0:
    cid = comm_unique_id()
    with open(idfile + ".tmp", "wb") as f: f.write(bytes(cid))
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120: raise SystemExit("rank %d: no communicator id after 120 s" % rank)
        time.sleep(0.01)
    cid = open(idfile, "rb").read()
pt = P.PT(mk(), rank=rank, world=world, comm_id=cid)            # RcclShard: pte_comm_init (collective) inside
assert type(pt.shards).__name__ == "RcclShard", type(pt.shards).__name__
one = P.PT(mk()) if rank == 0 else None
ok = pt.shards.n_ranks_seen == world and pt.replicas.comm_info()[0] == 1
why = []
for r in range(cfg["rounds"]):
    P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
    if rank == 0:
        P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
        checks = {"index_process": np.array_equal(ra.index_process, red.index_process), "round_trip": ra.round_trip == red.round_trip,
                  "swap_pr": np.array_equal(ra.swap_acceptance_pr[0], red.swap_acceptance_pr[0]),
                  "log_sum_ratio": np.array_equal(ra.log_sum_ratio[0], red.log_sum_ratio[0]),
                  "steps": np.array_equal(ra.explorer_n_steps[0], red.explorer_n_steps[0]),
                  "schedule": np.array_equal(one.shared.tempering.schedule.grids, pt.shared.tempering.schedule.grids)}
        for k, v in checks.items():
            if not v: ok = False; why.append("round %d: %s" % (r + 1, k))
x, chain, rng = pt.shards.states()                                # all-gather over the communicator
if rank == 0:
    xa, ca, ga = one.replicas.states()
    if not (np.array_equal(x, xa) and np.array_equal(chain, ca) and np.array_equal(rng, ga)): ok = False; why.append("final states")
pt.shards.barrier()
mx = float(pt.shards.allreduce_max([float(rank)])[0])
if mx != world - 1: ok = False; why.append("allreduce max %r" % mx)
swaps = [int(v) for v in pt.replicas.comm_info()[2]]
pt.replicas.comm_destroy()
print(json.dumps({"ok": bool(ok), "why": why, "rank": rank, "boundary_swaps": swaps, "kernel": pt.replicas.kernel_name()}))
'''


def _run_world(tmp_path, cfg, world):
    fake = build_fakerccl()
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    idfile = str(tmp_path / "comm_id.bin")
    env = dict(os.environ, PTE_RCCL_LIB=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, json.dumps(cfg), str(r), str(world), idfile], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # exactly the PIDs started above
    res = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        assert lines, (so[-1500:], se[-1500:])
        res.append(json.loads(lines[-1]))
    return res


@pytest.mark.parametrize("name,cfg,world", [
    ("K even", dict(explorer="slice", N=8, d=70, rounds=5, seed=5), 2),
    ("K odd", dict(explorer="slice", N=6, d=33, rounds=5, seed=2), 2),
    ("K = 1, three ranks", dict(explorer="slice", N=3, d=20, rounds=5, seed=3), 3),
    ("toy, four ranks", dict(explorer="toy", N=12, d=130, rounds=5, seed=4), 4),
    ("Compose", dict(explorer="slice+automala", N=8, d=40, rounds=4, seed=6), 2),
    ("Ising, bit-packed payloads", dict(explorer="ising", N=8, L=32, rounds=4, seed=7, beta=0.5), 2),
    ("C4 payload size (32 KiB)", dict(explorer="slice", N=4, d=4096, rounds=2, seed=8, no_swaps_expected=True), 2),   # (4 chains at d = 4096: no swap is ever accepted)
])
def test_rccl_shard_with_a_peer_equals_single_engine(tmp_path, name, cfg, world):
    res = _run_world(tmp_path, cfg, world)
    assert all(r["ok"] for r in res), res
    if not cfg.get("no_swaps_expected"):
        assert sum(sum(r["boundary_swaps"]) for r in res) > 0, res       # replicas did cross the rank boundary


def test_bench_two_ranks_on_one_device(tmp_path):
    """bench.py --gpus 2 end to end (self-launch, RcclShard, agreed transport, timed region, gathers, teardown)."""
    fake = build_fakerccl()
    env = dict(os.environ, PTE_RCCL_LIB=fake)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--steps", "6", "--warmup", "2",
                        "--round-trip-rounds", "4", "--no-cpu-baseline", "--timeout-s", "400"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 2 and c["n_ranks_seen"] == 2, c
    assert "RCCL send/recv enqueued by libpte" in c["sharding"] and "FALLBACK" not in c["sharding"], c
    assert len(c["boundary_swaps_per_rank"]) == 2 and sum(c["boundary_swaps_per_rank"]) > 0, c
    assert len(c["ms_per_step_per_rank"]) == 2 and "same_device_test_run" in c
    assert line["value"] > 0 and line["roofline"]["kernel"] == "k_explore_slice8"
    # round 4: the first line of a multi-GPU run says which library carried the messages, that the G-rank run is bit-identical to one
    # engine on this transport, what a boundary exchange costs on the stream, and which environment overrides were in effect
    assert "fakerccl" in c["transport_library"]["path"] and c["env_overrides"]["PTE_RCCL_LIB"] == fake
    assert c["parallelism_invariant"] is True
    be = c["boundary_exchange"]
    assert be["samples"] >= 4 and be["us_min_median_max"][0] > 0 and be["message_bytes_per_side"] == 8 * (1024 + 8)


def test_bench_strong_scaling_two_ranks_on_one_device():
    """bench.py --scaling strong (BASELINE configs[3]'s partitioning: a fixed ladder cut over the ranks) through the same 2-rank path,
    at a size that keeps the test short: 8192 chains of dimension 128 -> 4096 per rank = the many-replica kernel on every rank."""
    fake = build_fakerccl()
    env = dict(os.environ, PTE_RCCL_LIB=fake)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--scaling", "strong", "--chains", "8192",
                        "--dim", "128", "--steps", "4", "--warmup", "2", "--round-trip-rounds", "0", "--no-cpu-baseline", "--timeout-s", "400"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["config"]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and c["n_ranks_seen"] == 2 and c["chains_per_gpu"] == 4096
    assert line["roofline"]["kernel"] == "k_explore_slice8_lds10k" and "FALLBACK" not in c["sharding"]
    assert abs(line["value"] - 8192 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]


def test_rccl_override_needs_the_opt_in():
    """$PTE_RCCL_LIB alone must not re-route the transport: without pte_comm_allow_library_override(1) every pte_comm_* call fails loudly."""
    fake = build_fakerccl()
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from pigeons_amd.engine import comm_unique_id, comm_library, comm_allow_library_override\n"
            "from pigeons_amd import PteError\n"
            "try:\n    comm_unique_id(); print('NOT REFUSED')\n"
            "except PteError as e:\n    print('refused:', e)\n"
            "comm_allow_library_override(True); print('lib', comm_library())\n") % (ROOT, os.path.join(ROOT, "pigeons.jl_amd"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTE_RCCL_LIB=fake), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "refused:" in p.stdout and "did not opt in" in p.stdout and "NOT REFUSED" not in p.stdout, p.stdout
    assert "fakerccl" in p.stdout.split("lib", 1)[1], p.stdout
    assert "transport OVERRIDE" in p.stderr                      # ... and an override in effect says so on stderr
