"""CPU tests of the multi-rank (chain-sharded) path: the same drivers that move boundary replicas
between GPUs (pigeons_amd.sharded) run here over oracle-backed shards -- in one process
(LoopbackShards) and as two gloo ranks (DistShard over torch.distributed, world_size 2).
The output must equal the unsharded oracle: integers exactly, floats to rounding (the shard-local
recorders merge in a different order)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unsharded(N, d, explorer, scans_per_round):
    ref = O.OraclePT(n_chains=N, dim=d, explorer=explorer, record_online=1, record_traces=1, record_energy_ac1=1)
    out = []
    for n in scans_per_round:
        ref.begin_round(); ref.run_scans(n); 
        ref.L.po_end_round(ref.h)
        out.append(dict(ip=ref.index_process(), rt=ref.round_trip(), swap=ref.swap_pr(), lsr=ref.log_sum_ratio(),
                        expl=ref.explorer_stats(), sched=ref.schedule(), online=ref.online(),
                        eac=ref.energy_ac1(), traces=ref.traces(), online_lp=ref.online_lp()))
    return ref, out


@pytest.mark.parametrize("N,d,G,explorer", [(8, 6, 2, O.EXPLORER_SLICE), (9, 4, 3, O.EXPLORER_TOY), (6, 3, 6, O.EXPLORER_SLICE)])
def test_loopback_shards_equal_unsharded_oracle(N, d, G, explorer):
    from pigeons_amd.sharded import LoopbackShards
    from pigeons_amd import tempering as T
    scans = [2, 4, 8, 16]
    ref, want = _unsharded(N, d, explorer, scans)
    shards = LoopbackShards([O.OracleShard(rank=g, world_size=G, n_chains=N, dim=d, explorer=explorer, record_online=1,
                                           record_traces=1, record_energy_ac1=1) for g in range(G)])
    for r, n in enumerate(scans):
        shards.run_scans(1, n)
        red = shards.reduce()
        w = want[r]
        assert np.array_equal(red.index_process, w["ip"])
        assert red.round_trip == w["rt"]
        assert np.array_equal(red.swap_acceptance_pr[1], w["swap"][1])
        np.testing.assert_allclose(red.swap_acceptance_pr[0], w["swap"][0], rtol=1e-12)
        np.testing.assert_allclose(red.log_sum_ratio[0], w["lsr"][0], rtol=1e-12)
        np.testing.assert_allclose(red.log_sum_ratio[2], w["lsr"][2], rtol=1e-12)
        assert np.array_equal(red.explorer_n_steps[0], w["expl"][2])
        assert np.array_equal(red.traces, w["traces"])                          # target-chain samples: the last shard's
        assert np.array_equal(red.energy_ac1[1], w["eac"][1])
        np.testing.assert_allclose(red.energy_ac1[0], w["eac"][0], rtol=1e-9)   # merge order differs
        np.testing.assert_allclose(red.online_log_density, w["online_lp"][:2], rtol=1e-12)
        rej = T.rejections(*red.swap_acceptance_pr)
        old = shards.engines[0].schedule()
        new = T.optimal_schedule(rej, old, N)
        np.testing.assert_allclose(new, w["sched"], rtol=1e-11)
        shards.set_schedule(w["sched"])               # keep both runs on the identical ladder
    x, chain, rng = shards.states()
    xr, cr, rr = ref.states()
    assert np.array_equal(x, xr) and np.array_equal(chain, cr) and np.array_equal(rng, rr)
    assert shards.n_boundary_swaps > 0


WORKER = r'''
import os, sys, json
import numpy as np
sys.path[:0] = [%(root)r, %(root)r + "/pigeons.jl_amd", %(root)r + "/tests"]
import torch, torch.distributed as dist
import oracle as O
from pigeons_amd.sharded import DistShard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N, d = 10, 5
eng = O.OracleShard(rank=rank, world_size=world, n_chains=N, dim=d, explorer=O.EXPLORER_SLICE, record_online=1)
sh = DistShard(eng, rank, world)
ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_SLICE, record_online=1)
ok = True
for n in (2, 4, 8, 16, 32):
    sh.run_scans(1, n)
    red = sh.reduce()
    ref.begin_round(); ref.run_scans(n); ref.L.po_end_round(ref.h)
    ok &= bool(np.array_equal(red.index_process, ref.index_process()))
    ok &= red.round_trip == ref.round_trip()
    ok &= bool(np.allclose(red.swap_acceptance_pr[0], ref.swap_pr()[0], rtol=1e-12, atol=0))
    sh.set_schedule(ref.schedule())
x, chain, rng = sh.states()
xr, cr, rr = ref.states()
ok &= bool(np.array_equal(x, xr) and np.array_equal(chain, cr) and np.array_equal(rng, rr))
flags = [None] * world
dist.all_gather_object(flags, (ok, sh.n_boundary_swaps))
if rank == 0:
    print(json.dumps({"ok": all(f[0] for f in flags), "boundary_swaps": [f[1] for f in flags]}))
dist.destroy_process_group()
'''


def test_two_gloo_ranks_equal_unsharded_oracle(tmp_path):
    """world_size-2 gloo run of DistShard (the driver bench.py uses with nccl/RCCL on GPUs)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["ok"], res
    assert sum(res["boundary_swaps"]) > 0


def test_loopback_shards_extended_traces():
    """extended_traces through the shard drivers: every shard traces its local chains, the reduction concatenates them."""
    from pigeons_amd.sharded import LoopbackShards
    N, d, G = 8, 3, 4
    ref = O.OraclePT(n_chains=N, dim=d, explorer=O.EXPLORER_SLICE, record_traces=2)
    shards = LoopbackShards([O.OracleShard(rank=g, world_size=G, n_chains=N, dim=d, explorer=O.EXPLORER_SLICE, record_traces=2)
                             for g in range(G)])
    for n in (2, 4, 8):
        ref.begin_round(); ref.run_scans(n); ref.L.po_end_round(ref.h)
        shards.run_scans(1, n)
        red = shards.reduce()
        assert red.traces.shape == (n, N, d + 1)
        assert np.array_equal(red.traces, ref.traces())
        shards.set_schedule(ref.schedule())
