"""The REAL librccl with real peers, the day `pytest -m gpu` meets a node with >= 2 GPUs (VERDICT r04 "next round" item 4).

The reference runs its distributed swap under 2 MPI processes in CI (test/test_parallelism_invariance.jl:5-37; src/swap/swap.jl:79-102).
Here every multi-rank test of tests/test_gpu_rccl_peer.py forces the stand-in (the build boxes have ONE GPU and RCCL refuses two ranks on
one device), so until now a multi-GPU box would have run the whole suite without one real ncclSend.  This module runs the production path
-- RcclShard -> pte_comm_init -> pte_run_scans, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd enqueued by libpte on the engine's
stream -- over the REAL library, one fresh process per GPU, G = 2, 4, 8 (as many as the node shows), with BASELINE configs[3]'s payload
(d = 4096: 32 KiB + 64 B per message) and configs[4]'s (256 x 256 spins bit-packed: 8 KiB), and asserts

  * index process, round trips, swap / log-sum recorders, explorer step counts, schedule after every round and the final states, RNG
    counters and chains bit-identical to ONE engine holding the whole ladder (rank 0 runs it after the sharded run's last collective),
  * n_ranks_seen == G, replicas did cross rank boundaries,
  * pte_comm_library names a file that is not the stand-in (and $PTE_RCCL_LIB is not set in the ranks),
  * and prints the boundary exchange's median duration on the engine's stream.

On a 1-GPU box the module is collected and SKIPPED (torch.cuda.device_count() does not initialise HIP on this image; the launcher process
never touches the GPU; the ranks are fresh children that exit non-zero on their own 300 s watchdog -- no re-exec anywhere).
tools/scale8.sh runs this module before it times anything.  (The file name sorts it behind every other module: on the first multi-GPU box
`pytest -x` should have run the whole single-GPU suite before it meets a transport nobody has been able to run yet.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    try:
        import torch
        return int(torch.cuda.device_count())          # counts devices without creating a HIP context (this image)
    except Exception:
        return 0


N_GPUS = _n_gpus()

WORKER = r'''
import os, sys, json, time, threading
root, cfg, rank, world, idfile = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
def _watchdog():
    sys.stderr.write("rank %d: still running after 300 s (a peer died or a collective hangs)\n" % rank); sys.stderr.flush(); os._exit(124)
_t = threading.Timer(300.0, _watchdog); _t.daemon = True; _t.start()
import numpy as np
sys.path[:0] = [root, root + "/pigeons.jl_amd", root + "/tests"]
import pigeons_amd as P
from pigeons_amd.engine import comm_unique_id, comm_library
STAND_IN = bool(cfg.get("stand_in"))            # only the 1-GPU self-test of THIS script sets it (every rank on device 0, tests/fakerccl)
if STAND_IN:
    from pigeons_amd.engine import comm_allow_library_override
    comm_allow_library_override(True)
else:
    assert not os.environ.get("PTE_RCCL_LIB"), "a real-RCCL run takes no transport override"
lib_path, lib_ver = comm_library()
assert ("fakerccl" in lib_path) == STAND_IN and lib_ver > 0, (lib_path, lib_ver)
DEVICE = 0 if STAND_IN else rank

def mk():
    rec = [P.round_trip, P.index_process, P.log_sum_ratio]
    if cfg["explorer"] == "ising":
        return P.Inputs(target=P.IsingLogPotential(cfg.get("beta", 1.0), cfg["L"]), n_chains=cfg["N"], n_rounds=cfg["rounds"],
                        seed=cfg.get("seed", 1), record=rec, show_report=False, device=DEVICE)
    expl = P.SliceSampler() if cfg["explorer"] == "slice" else P.ToyExplorer()
    return P.Inputs(target=P.toy_mvn_target(cfg["d"]), n_chains=cfg["N"], n_rounds=cfg["rounds"], explorer=expl,
                    seed=cfg.get("seed", 1), record=rec, show_report=False, device=DEVICE)

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
REFRED = bool(cfg.get("refred"))                                  # PTE_RECORD_REFERENCE_REDUCTION: every rank replays the pairs whose lower chain it owns
pt = P.PT(mk(), rank=rank, world=world, comm_id=cid, reference_reduction=REFRED)            # RcclShard: pte_comm_init (collective) inside
assert type(pt.shards).__name__ == "RcclShard", type(pt.shards).__name__
ok = pt.shards.n_ranks_seen == world and pt.replicas.comm_info()[0] == 1
why = [] if ok else ["n_ranks_seen %r" % pt.shards.n_ranks_seen]
sharded = []
for r in range(cfg["rounds"]):
    P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
    sharded.append((red.index_process.copy(), red.round_trip, red.swap_acceptance_pr[0].copy(), red.log_sum_ratio[0].copy(),
                    red.explorer_n_steps[0].copy(), np.array(pt.shared.tempering.schedule.grids).copy()))
x, chain, rng = pt.shards.states()                                # all-gather over the communicator
# what a boundary exchange costs on the engine's stream (HIP events around the grouped send / recv, one sample per even scan)
e = pt.replicas
n_t = min(16, 2 ** cfg["rounds"])                   # (the index-process buffer holds one round of the run's last round: 2^rounds scans)
e.timing_reset(True); pt.shards.run_scans(1, n_t); bs = np.sort(e.timing_samples(3)); e.timing_reset(False)
pt.shards.reduce()
pt.shards.barrier()
mx = float(pt.shards.allreduce_max([float(rank)])[0])
if mx != world - 1: ok = False; why.append("allreduce max %r" % mx)
swaps = [int(v) for v in pt.replicas.comm_info()[2]]
kernel = pt.replicas.kernel_name()
pt.replicas.comm_destroy()
# the single engine on rank 0: after the sharded run's last collective, so that a failure here cannot leave a peer inside one
if rank == 0:
    one = P.PT(mk(), reference_reduction=REFRED)
    for r in range(cfg["rounds"]):
        P.next_round(one); ra = P.run_one_round(one); P.adapt(one, ra)
        ip, rt, sw, ls, st, gr = sharded[r]
        checks = {"index_process": np.array_equal(ra.index_process, ip), "round_trip": ra.round_trip == rt,
                  "swap_pr": np.array_equal(ra.swap_acceptance_pr[0], sw), "log_sum_ratio": np.array_equal(ra.log_sum_ratio[0], ls),
                  "steps": np.array_equal(ra.explorer_n_steps[0], st), "schedule": np.array_equal(one.shared.tempering.schedule.grids, gr)}
        for k, v in checks.items():
            if not v: ok = False; why.append("round %d: %s" % (r + 1, k))
    xa, ca, ga = one.replicas.states()
    if not (np.array_equal(x, xa) and np.array_equal(chain, ca) and np.array_equal(rng, ga)): ok = False; why.append("final states")
print(json.dumps({"ok": bool(ok), "why": why, "rank": rank, "boundary_swaps": swaps, "kernel": kernel, "transport_library": [lib_path, lib_ver],
                  "n_ranks_seen": int(pt.shards.n_ranks_seen),
                  "boundary_exchange_us_min_median_max": [float(bs[0]) * 1e3, float(bs[len(bs) // 2]) * 1e3, float(bs[-1]) * 1e3] if len(bs) else None}))
sys.stdout.flush()
_t.cancel()
'''


def _run_world(tmp_path, cfg, world):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    import uuid
    idfile = str(tmp_path / ("comm_id_%s.bin" % uuid.uuid4().hex))      # one id file per world: a second world in the same tmp_path must not read the first one's
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("PTE_RCCL_LIB", "PTE_LIB"):
        env.pop(k, None)
    if cfg.get("stand_in"):
        env["PTE_RCCL_LIB"] = cfg["stand_in"]
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


def _check(res, world, kernel, stand_in=False, require_swaps=True):
    assert all(r["ok"] for r in res), res
    assert all(r["n_ranks_seen"] == world for r in res), res
    assert all(("fakerccl" in r["transport_library"][0]) == stand_in for r in res), res
    assert all(r["kernel"].startswith(kernel) for r in res), res
    if require_swaps:
        assert sum(sum(r["boundary_swaps"]) for r in res) > 0, res   # replicas did cross rank boundaries
    be = res[0]["boundary_exchange_us_min_median_max"]
    assert be and be[0] > 0, res[0]
    print("real RCCL, G = %d: %s v%s; boundary exchange min / median / max %.1f / %.1f / %.1f us on rank 0's stream; boundary swaps per rank %s"
          % (world, res[0]["transport_library"][0], res[0]["transport_library"][1], be[0], be[1], be[2], [r["boundary_swaps"] for r in res]))


@pytest.mark.skipif(N_GPUS < 2, reason="needs >= 2 GPUs on the node (%d visible): the real librccl refuses two ranks on one device" % N_GPUS)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_rccl_config4_payload_equals_single_engine(tmp_path, world):
    """BASELINE configs[3]'s message (d = 4096: 8 (d + 8) B per boundary side and even scan), 256 chains per GPU, SliceSampler, rounds 1-3"""
    if world > N_GPUS:
        pytest.skip("%d GPUs visible" % N_GPUS)
    res = _run_world(tmp_path, dict(explorer="slice", d=4096, N=256 * world, rounds=3, seed=3), world)
    _check(res, world, "k_explore_slice8")


@pytest.mark.skipif(N_GPUS < 2, reason="needs >= 2 GPUs on the node (%d visible): the real librccl refuses two ranks on one device" % N_GPUS)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_rccl_config5_payload_equals_single_engine(tmp_path, world):
    """BASELINE configs[4]'s message (256 x 256 spins bit-packed: 8 KiB), 32 chains per GPU, IsingMetropolis(3), rounds 1-3"""
    if world > N_GPUS:
        pytest.skip("%d GPUs visible" % N_GPUS)
    res = _run_world(tmp_path, dict(explorer="ising", L=256, beta=1.0, N=32 * world, rounds=3, seed=5), world)
    _check(res, world, "k_explore_ising_spec")


def test_the_rank_script_of_this_module_runs(tmp_path):
    """So that the script above is not first executed the day a multi-GPU node appears: the same ranks, the same checks, two ranks on
    device 0 over the stand-in (the only thing this run does NOT exercise is librccl itself).  Runs on every GPU box."""
    from test_gpu_rccl_peer import build_fakerccl
    fake = build_fakerccl()
    # (the stand-in moves at most 256 KiB per message, so the states' all-gather bounds chains x dimension per rank here; the real tests above
    # run 256 chains per GPU at d = 4096)
    res = _run_world(tmp_path, dict(explorer="slice", d=256, N=128, rounds=3, seed=3, stand_in=fake), 2)
    _check(res, 2, "k_explore_slice8", stand_in=True)
    res = _run_world(tmp_path, dict(explorer="slice", d=4096, N=8, rounds=3, seed=3, stand_in=fake), 2)        # C4's message size (8 chains: no swap is ever accepted)
    _check(res, 2, "k_explore_slice8", stand_in=True, require_swaps=False)
    res = _run_world(tmp_path, dict(explorer="ising", L=256, beta=1.0, N=8, rounds=3, seed=5, stand_in=fake), 2)
    _check(res, 2, "k_explore_ising_spec", stand_in=True, require_swaps=False)
    res = _run_world(tmp_path, dict(explorer="slice", d=64, N=32, rounds=5, seed=2, refred=True, stand_in=fake), 2)      # the reference's reduction replayed per rank
    _check(res, 2, "k_explore_slice8", stand_in=True)


@pytest.mark.skipif(N_GPUS < 2, reason="needs >= 2 GPUs on the node (%d visible)" % N_GPUS)
def test_real_rccl_bench_line_two_gpus():
    """bench.py --gpus 2 over the real library: the line names librccl, says the cut ladder is bit-identical to one engine on this
    transport and carries the boundary exchange's duration"""
    env = dict(os.environ)
    for k in ("PTE_RCCL_LIB", "PTE_LIB", "PTE_BENCH_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4", "--round-trip-rounds", "0",
                        "--no-cpu-baseline", "--timeout-s", "600"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 2 and c["n_ranks_seen"] == 2 and "FALLBACK" not in c["sharding"], c
    assert "fakerccl" not in c["transport_library"]["path"] and c["transport_library"]["nccl_version"] > 0, c
    assert c["parallelism_invariant"] is True and c["env_overrides"] == {}, c
    assert c["boundary_exchange"]["samples"] >= 4, c
