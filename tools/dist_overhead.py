"""Per-scan cost of the stream-ordered RCCL driver's host side (1-rank nccl group: same enqueue path, no peers)
against the fused single-engine loop, metric workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
import torch, torch.distributed as dist
import pigeons_amd as P
from pigeons_amd.sharded import DistShard
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
mk = lambda: P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, n_rounds=10, explorer=P.SliceSampler(), show_report=False,
                      record=[P.round_trip, P.log_sum_ratio])
a = P.PT(mk()); b = P.PT(mk())
sh = DistShard(b.replicas, 0, 1, device=torch.device("cuda", 0))
assert sh.stream_ordered
for runner, name in ((a.replicas, "pte_run_scans (one ccall)"), (sh, "DistShard stream-ordered (per-scan enqueue from Python)")):
    runner.run_scans(1, 8); torch.cuda.synchronize()
    t = time.perf_counter(); runner.run_scans(1, 64); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-58s %.3f ms/scan" % (name, dt / 64 * 1e3))
dist.destroy_process_group()
