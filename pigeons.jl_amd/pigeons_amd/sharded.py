"""Chain-sharded parallel tempering: N chains over G engines (one engine per GPU).

Rank g owns chains [g*K, (g+1)*K), K = N/G, and the replicas currently at those chains.  explore!
needs no communication.  A DEO scan couples only the G-1 boundary pairs: their two SwapStats
(16 B each way) are exchanged between the two phases of the swap, and iff the swap is accepted the
two replicas' payloads {state, sum x^2, rng, replica id, round-trip state} trade places.  Everything
a replica owns travels with it, so the output is identical for any G -- the reference's
"parallelism invariance" (docs/src/distributed.md:37-58; distributed swap! src/swap/swap.jl:79-102,
which instead shards by replica and moves chain labels with 4 any-to-any transmits per scan).

Drivers (all produce bit-identical results):
  * RcclShard      -- production, one engine per process / GPU.  The transport lives BEHIND the C ABI
                      (include/pte.h: pte_comm_init + pte_run_scans): libpte enqueues ncclSend / ncclRecv of
                      ONE message per active side and scan {SwapStat, speculative payload} on the engine's
                      own HIP stream between its pack and decide kernels -- no host synchronisation inside
                      the scan loop, no torch.distributed on the data path (this replaces the reference's
                      src/mpi_utils/Entangler.jl).  Python only hands the 128-byte communicator id around.
  * LoopbackShards -- G engines in ONE process (tests on a single GPU): transport="group" is the library's
                      pte_group_run_scans (stream-ordered device copies, same kernels as RcclShard);
                      device_messages=True drives the same kernels step by step from Python;
                      default: the two-phase calls with host copies.
  * DistShard      -- host-driven two-phase exchange over torch.distributed point-to-point (gloo); this is
                      what the CPU tests run with oracle-backed shards (world_size 2).
A shard "engine" is anything with the Engine methods used below (the HIP Engine in production; the
oracle-backed shard in the CPU tests).
"""
import numpy as np

from .pt import ReducedRecorders


def _combine_traces(parts):
    trs = [p.get("traces") for p in parts]
    if trs[-1] is None:
        return None
    if trs[-1].ndim == 3:                               # extended_traces: every shard traced its local chains
        return np.concatenate(trs, axis=1)
    return trs[-1]                                      # the last shard owns the target chain


def combine_reduced(parts, N, d):
    """Assemble the global reduced recorders from the per-shard slices (ordered by rank)."""
    cat = lambda xs: np.concatenate([np.asarray(x) for x in xs]) if xs else np.zeros(0)
    sw_m = cat([p["swap"][0] for p in parts]); sw_n = cat([p["swap"][1] for p in parts])
    up = cat([p["lsr"][0] for p in parts]); un = cat([p["lsr"][1] for p in parts])
    dn = cat([p["lsr"][2] for p in parts]); dnn = cat([p["lsr"][3] for p in parts])
    assert len(sw_m) == max(N - 1, 0), (len(sw_m), N)
    restarts = int(sum(p["round_trip"][0] for p in parts)); trips = int(sum(p["round_trip"][1] for p in parts))
    am = cat([p["explorer"][0] for p in parts]); an = cat([p["explorer"][1] for p in parts])
    ss = cat([p["explorer"][2] for p in parts]); sn = cat([p["explorer"][3] for p in parts])
    ip = None
    reps = [p["ip"][0] for p in parts]
    if reps and reps[0].size:
        T = reps[0].shape[0]
        ip = np.full((N, T), -1, dtype=np.int64)
        for p in parts:
            rep, ch = p["ip"]
            t_idx = np.repeat(np.arange(T)[:, None], rep.shape[1], axis=1)
            ip[rep, t_idx] = ch
        assert ip.min() >= 0
    fm = cat([p["automala"][0] for p in parts]); fn = cat([p["automala"][1] for p in parts])
    rm = cat([p["automala"][2] for p in parts]); rn = cat([p["automala"][3] for p in parts])
    online = parts[-1]["online"]            # the last shard owns the target chain
    eac = None
    if all("eac" in p for p in parts):
        eac = (cat([p["eac"][0] for p in parts]), cat([p["eac"][1] for p in parts]),
               np.concatenate([np.asarray(p["eac"][2]).reshape(-1, 5) for p in parts]))
    return ReducedRecorders(swap_acceptance_pr=(sw_m, sw_n), log_sum_ratio=(up, un, dn, dnn),
                            round_trip=(restarts, trips), index_process=ip,
                            explorer_acceptance_pr=(am, an), explorer_n_steps=(ss, sn), online=online,
                            am_factors=(fm, fn), reversibility_rate=(rm, rn),
                            online_log_density=parts[-1].get("online_lp"), energy_ac1=eac, traces=_combine_traces(parts),
                            timing_extrema={"round": None})


def local_reduced(eng):
    eng.reduce()
    extra = {}
    if hasattr(eng, "energy_ac1") and hasattr(eng, "online_log_density"):
        extra = {"eac": eng.energy_ac1(), "online_lp": eng.online_log_density(), "traces": eng.traces()}
    return {**extra, "swap": eng.swap_acceptance(), "lsr": eng.log_sum_ratio(), "round_trip": eng.round_trip(),
            "explorer": eng.explorer_stats(), "ip": eng.index_process_shard(), "online": eng.online(),
            "automala": eng.automala_stats() if hasattr(eng, "automala_stats") else tuple(np.zeros(eng.K) for _ in range(4))}


class LoopbackShards:
    """G shard engines in one process, run scan-synchronously; boundary bytes move by host copies."""

    def __init__(self, engines, device_messages=False, transport=None):
        self.engines = list(engines)
        self.transport = transport                      # "group": pte_group_run_scans (the library drives the G engines)
        self.G = len(self.engines)
        self.N = self.engines[0].N
        self.d = self.engines[0].d
        self.n_boundary_swaps = 0
        self.device_messages = device_messages
        if device_messages:
            import torch
            self.torch = torch
            w = self.engines[0].message_words()
            dev = torch.device("cuda", 0)
            self.msg = [[torch.zeros(w, dtype=torch.float64, device=dev) for _ in range(4)] for _ in self.engines]
            for e, m in zip(self.engines, self.msg):      # order: send_lo, recv_lo, send_hi, recv_hi
                e.shard_set_buffers(*[t.data_ptr() for t in m])

    def _run_scans_device(self, first_scan, n_scans):
        """Same kernels as the RCCL path; the transfers are device-to-device copies with a full
        synchronisation between the phases (this driver tests the kernels, not the stream ordering)."""
        E, G, torch = self.engines, self.G, self.torch
        for s in range(first_scan, first_scan + n_scans):
            active = [e.shard_scan_begin(s) for e in E]
            for e in E:
                e.shard_sync()
            for g in range(G - 1):
                assert active[g][1] == active[g + 1][0]
                if active[g][1]:
                    self.msg[g + 1][1].copy_(self.msg[g][2])      # upper neighbour's recv_lo <- my send_hi
                    self.msg[g][3].copy_(self.msg[g + 1][0])      # my recv_hi <- upper neighbour's send_lo
            torch.cuda.synchronize()
            for e in E:
                e.shard_scan_finish(s)
        self.n_boundary_swaps = int(sum(e.shard_sync()[1] for e in E))

    def run_scans(self, first_scan, n_scans):
        if self.transport == "group":
            from .engine import group_run_scans
            group_run_scans(self.engines, first_scan, n_scans)
            self.n_boundary_swaps = int(sum(e.comm_info()[2].sum() for e in self.engines)) // 2   # both sides count an applied swap
            return
        if self.device_messages:
            return self._run_scans_device(first_scan, n_scans)
        E, G = self.engines, self.G
        for s in range(first_scan, first_scan + n_scans):
            for e in E:
                e.explore(s)
            begun = [e.swap_begin(s) for e in E]
            accepted = []
            for g, e in enumerate(E):
                nbr = np.zeros(4)
                stats, active = begun[g]
                if active[0]:
                    assert g > 0 and begun[g - 1][1][1]
                    nbr[0:2] = begun[g - 1][0][2:4]          # upper stats of the lower neighbour
                if active[1]:
                    assert g + 1 < G and begun[g + 1][1][0]
                    nbr[2:4] = begun[g + 1][0][0:2]          # lower stats of the upper neighbour
                accepted.append(e.swap_finish(s, nbr))
            for g in range(G - 1):
                a, b = accepted[g][1], accepted[g + 1][0]
                assert a == b, "both sides of a boundary must take the same decision"
                if a:
                    w = E[g].payload_words()
                    lo = np.zeros(w); hi = np.zeros(w)
                    E[g].boundary_export(1, lo.ctypes.data, False)
                    E[g + 1].boundary_export(0, hi.ctypes.data, False)
                    E[g].boundary_import(1, hi.ctypes.data, False)
                    E[g + 1].boundary_import(0, lo.ctypes.data, False)
                    self.n_boundary_swaps += 1

    def reduce(self):
        return combine_reduced([local_reduced(e) for e in self.engines], self.N, self.d)

    def set_schedule(self, betas):
        for e in self.engines:
            e.set_schedule(betas)

    def states(self):
        """Global (x, chain, rng) in replica order, gathered from the shards."""
        N, d = self.N, self.d
        x = np.zeros((N, d)); chain = np.zeros(N, dtype=np.int64); rng = np.zeros((N, 2), dtype=np.uint64)
        for e in self.engines:
            xs, cs, rs = e.states()
            ids = e.replica_ids()
            x[ids] = xs; chain[ids] = cs; rng[ids] = rs
        return x, chain, rng


class RcclShard:
    """One shard per process; the boundary exchange is RCCL send/recv enqueued by libpte itself (pte_comm_*).

    `id_bytes`: the communicator id of rank 0's comm_unique_id(), already distributed by the caller; or
    `bcast(obj_or_None) -> obj`: any broadcast-from-rank-0 callable (default: torch.distributed's, when a process
    group exists).  Only these 128 bytes ever go through the host-side launcher."""

    def __init__(self, engine, rank, world, id_bytes=None, bcast=None):
        from .engine import comm_unique_id
        self.e, self.rank, self.world = engine, rank, world
        self.N, self.d = engine.N, engine.d
        self.n_boundary_swaps = 0
        if id_bytes is None:
            if bcast is None:
                import torch.distributed as dist
                if not dist.is_initialized():
                    raise RuntimeError("RcclShard needs id_bytes, a bcast callable, or an initialised torch.distributed group")

                def bcast(obj):
                    box = [obj]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            msg = None
            if rank == 0:
                try:
                    msg = ("ok", comm_unique_id())
                except Exception as exc:                  # every rank must fail, not only rank 0
                    msg = ("error", repr(exc))
            kind, val = bcast(msg)
            if kind != "ok":
                raise RuntimeError("rank 0 could not create the RCCL communicator id: %s" % val)
            id_bytes = val
        engine.comm_init(id_bytes)                       # collective: ncclCommInitRank
        self.n_ranks_seen = engine.comm_info()[1]

    def run_scans(self, first_scan, n_scans):
        self.e.run_scans(first_scan, n_scans)            # explore + swap + boundary exchange, all inside libpte
        self.n_boundary_swaps = int(self.e.comm_info()[2].sum())

    def barrier(self):
        self.e.comm_barrier()

    def allreduce_max(self, values):
        return self.e.comm_allreduce(values, "max")

    def _allgather(self, obj):
        import pickle
        return [pickle.loads(b) for b in self.e.comm_allgather_bytes(pickle.dumps(obj, protocol=4))]

    def reduce(self):
        """All ranks obtain the same global reduced recorders (all-gather of the local slices, rank order)."""
        return combine_reduced(self._allgather(local_reduced(self.e)), self.N, self.d)

    def set_schedule(self, betas):
        self.e.set_schedule(betas)

    def states(self):
        xs, cs, rs = self.e.states()
        parts = self._allgather((self.e.replica_ids(), xs, cs, rs))
        N, d = self.N, self.d
        x = np.zeros((N, d)); chain = np.zeros(N, dtype=np.int64); rng = np.zeros((N, 2), dtype=np.uint64)
        for ids, xx, cc, rr in parts:
            x[ids] = xx; chain[ids] = cc; rng[ids] = rr
        return x, chain, rng


class DistShard:
    """One shard per process; the HOST moves the boundary bytes between the two phases of the swap through
    torch.distributed point-to-point (gloo).  Three host synchronisations per scan -- the CPU tests' driver
    (oracle-backed shards) and the fallback of a host that has no RCCL; production uses RcclShard."""

    def __init__(self, engine, rank, world, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self.N, self.d = engine.N, engine.d
        self.device = device if device is not None else torch.device("cpu")
        self.on_device = self.device.type == "cuda"
        w = engine.payload_words()
        self.send_buf = [torch.zeros(w, dtype=torch.float64, device=self.device) for _ in range(2)]
        self.recv_buf = [torch.zeros(w, dtype=torch.float64, device=self.device) for _ in range(2)]
        self.stat_send = [torch.zeros(2, dtype=torch.float64, device=self.device) for _ in range(2)]
        self.stat_recv = [torch.zeros(2, dtype=torch.float64, device=self.device) for _ in range(2)]
        self.n_boundary_swaps = 0

    def barrier(self):
        self.dist.barrier(group=self.group)

    def _exchange(self, sides, send, recv):
        dist = self.dist
        ops = []
        for side in sides:
            peer = self.rank - 1 if side == 0 else self.rank + 1
            ops.append(dist.P2POp(dist.isend, send[side], peer, self.group))
            ops.append(dist.P2POp(dist.irecv, recv[side], peer, self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def run_scans(self, first_scan, n_scans):
        e, torch = self.e, self.torch
        for s in range(first_scan, first_scan + n_scans):
            e.explore(s)
            stats, active = e.swap_begin(s)
            sides = [sd for sd in (0, 1) if active[sd]]
            for sd in sides:
                self.stat_send[sd].copy_(torch.from_numpy(stats[2 * sd:2 * sd + 2].copy()))
            self._exchange(sides, self.stat_send, self.stat_recv)
            nbr = np.zeros(4)
            for sd in sides:
                nbr[2 * sd:2 * sd + 2] = self.stat_recv[sd].cpu().numpy()
            acc = e.swap_finish(s, nbr)
            sides = [sd for sd in (0, 1) if acc[sd]]
            for sd in sides:
                e.boundary_export(sd, self.send_buf[sd].data_ptr(), self.on_device)
            self._exchange(sides, self.send_buf, self.recv_buf)
            if self.on_device and sides:
                torch.cuda.synchronize(self.device)
            for sd in sides:
                e.boundary_import(sd, self.recv_buf[sd].data_ptr(), self.on_device)
                self.n_boundary_swaps += 1

    def reduce(self):
        """All ranks obtain the same global reduced recorders (all_gather of the local slices)."""
        part = local_reduced(self.e)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, part, group=self.group)
        return combine_reduced(parts, self.N, self.d)

    def set_schedule(self, betas):
        self.e.set_schedule(betas)

    def states(self):
        xs, cs, rs = self.e.states()
        part = (self.e.replica_ids(), xs, cs, rs)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, part, group=self.group)
        N, d = self.N, self.d
        x = np.zeros((N, d)); chain = np.zeros(N, dtype=np.int64); rng = np.zeros((N, 2), dtype=np.uint64)
        for ids, xx, cc, rr in parts:
            x[ids] = xx; chain[ids] = cc; rng[ids] = rr
        return x, chain, rng
