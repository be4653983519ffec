"""Checkpoint / resume of a device run (reference src/pt/checkpoint.jl:19-54,110-145,166-189).

Same folder layout as the reference --
    <exec_folder>/inputs.pkl
    <exec_folder>/round=<r>/checkpoint/{shared.pkl, reduced_recorders.pkl, replica=<i>.npz, .signal/finished_replica=<i>}
-- with Python pickles / npz in place of Julia's `.jls` serialisation (no Julia in the build image; the
Julia glue would `serialize` the same fields: INTEGRATION.md).  A replica file holds what `Replica`
holds (src/replicas/Replica.jl:5-30): state, chain, rng (seed, gamma), replica_index; everything is read
back through `pte_get_state` / written through `pte_set_state`, so a resumed run continues bit for bit.
The replica files are plain npz (no pickled objects); they are what another implementation resumes from --
tests/test_gpu_parity.py::test_checkpoint_resumed_on_the_oracle loads them into the CPU oracle.

Trust: `inputs.pkl`, `shared.pkl`, `reduced_recorders.pkl` are Python pickles, exactly as the reference's are Julia
`Serialization` streams (checkpoint.jl:26-33) -- loading either executes whatever the file says.  Only load checkpoint
folders you wrote yourself.

Sharded runs (one process per GPU): every rank takes part in gathering the replicas, rank 0 alone writes the files,
a barrier follows (the reference writes one replica file per process and waits on .signal files, checkpoint.jl:120-145).
"""
import os
import pickle
import time

import numpy as np


def next_exec_folder(root="results"):
    """next_exec_folder() (src/pt/exec_folder.jl via checkpoint.jl:110-113): results/all/<time stamp>-<suffix>, and
    results/latest pointing at it."""
    stamp = time.strftime("%Y-%m-%d-%H-%M-%S") + "-" + "".join("%02x" % b for b in os.urandom(4))
    folder = os.path.join(root, "all", stamp)
    os.makedirs(folder, exist_ok=True)
    latest = os.path.join(root, "latest")
    try:
        if os.path.islink(latest):
            os.unlink(latest)
        os.symlink(os.path.join("all", stamp), latest)
    except OSError:
        pass
    return folder


def checkpoint_folder(exec_folder, round_):
    return os.path.join(exec_folder, "round=%d" % round_, "checkpoint")


def write_checkpoint(pt, exec_folder=None):
    """write_checkpoint(pt) (checkpoint.jl:110-145): a no-op unless inputs.checkpoint (or a folder is given).  A folder
    passed explicitly is used for this call only; pt.exec_folder is not touched."""
    explicit = exec_folder is not None
    exec_folder = exec_folder or getattr(pt, "exec_folder", None)
    if exec_folder is None or pt.shards is not None and not hasattr(pt.shards, "states"):
        return None
    r = pt.shared.iterators.round
    folder = checkpoint_folder(exec_folder, r)
    eng = pt.shards if pt.shards is not None else pt.replicas
    x, chain, rng = eng.states()                       # replica order; sharded: a collective, every rank takes part
    rank = int(getattr(pt.shards, "rank", 0)) if pt.shards is not None else 0
    if rank == 0:
        os.makedirs(os.path.join(folder, ".signal"), exist_ok=True)
        for i in range(len(chain)):
            np.savez(os.path.join(folder, "replica=%d.npz" % (i + 1)), state=x[i], chain=np.int64(chain[i] + 1),
                     rng=rng[i], replica_index=np.int64(i + 1))
        with open(os.path.join(folder, "shared.pkl"), "wb") as f:
            pickle.dump(pt.shared, f)
        with open(os.path.join(folder, "reduced_recorders.pkl"), "wb") as f:
            pickle.dump(pt.reduced_recorders, f)
        if not os.path.exists(os.path.join(exec_folder, "inputs.pkl")):
            with open(os.path.join(exec_folder, "inputs.pkl"), "wb") as f:
                pickle.dump(pt.inputs, f)
        for i in range(len(chain)):
            open(os.path.join(folder, ".signal", "finished_replica=%d" % (i + 1)), "w").close()
    if pt.shards is not None and hasattr(pt.shards, "barrier"):
        pt.shards.barrier()                            # nobody returns before the files are complete
    if not explicit:
        pt.exec_folder = exec_folder
    return folder


def latest_checkpoint_folder(exec_folder):
    """checkpoint.jl:56-72: the last round whose checkpoint is complete (all replicas signalled), 0 if none."""
    try:
        with open(os.path.join(exec_folder, "inputs.pkl"), "rb") as f:
            inputs = pickle.load(f)
    except OSError:
        return 0
    best = 0
    for r in range(1, 64):
        folder = checkpoint_folder(exec_folder, r)
        if not os.path.isdir(folder):
            continue
        n_total = inputs.n_chains + int(getattr(inputs, "n_chains_variational", 0) or 0)          # Inputs.jl:128
        done = all(os.path.exists(os.path.join(folder, ".signal", "finished_replica=%d" % (i + 1))) for i in range(n_total))
        if done and os.path.exists(os.path.join(folder, "shared.pkl")):
            best = r
    return best


def load_checkpoint(source_exec_folder, round=None, n_rounds_increment=0, **pt_kwargs):
    """PT(source_exec_folder; round) (checkpoint.jl:19-54) [+ increment_n_rounds! :166-189]: a fresh engine whose
    replicas, schedule and explorer adaptation are those of the checkpoint."""
    from .pt import PT, AutoMALA, MALA, Compose
    round = latest_checkpoint_folder(source_exec_folder) if round is None else round
    if round == 0:
        raise RuntimeError("No checkpoint found for %s (was checkpoint=True set?)" % source_exec_folder)
    if round < 0:
        raise ValueError("round should be positive")
    folder = checkpoint_folder(source_exec_folder, round)
    with open(os.path.join(source_exec_folder, "inputs.pkl"), "rb") as f:
        inputs = pickle.load(f)
    with open(os.path.join(folder, "shared.pkl"), "rb") as f:
        shared = pickle.load(f)
    with open(os.path.join(folder, "reduced_recorders.pkl"), "rb") as f:
        reduced = pickle.load(f)
    inputs.n_rounds += n_rounds_increment
    inputs.explorer = shared.explorer                  # carries the adapted step size / std deviations
    pt = PT(inputs, **pt_kwargs)
    N = pt.replicas.N
    reps = [np.load(os.path.join(folder, "replica=%d.npz" % (i + 1))) for i in range(N)]
    x = np.stack([np.atleast_1d(r["state"]) for r in reps])
    chain = np.array([int(r["chain"]) - 1 for r in reps], dtype=np.int64)
    rng = np.stack([r["rng"] for r in reps]).astype(np.uint64)
    pt.shared = shared
    pt.reduced_recorders = reduced
    eng = pt.replicas
    if pt.shards is not None:
        raise NotImplementedError("load a checkpoint into a single engine (sharded runs checkpoint through shards.states())")
    eng.set_schedule(shared.tempering.schedule.grids)
    eng.set_states(x if eng.d > 0 else None, chain, rng)

    def grad_sampler(ex):
        if isinstance(ex, Compose):
            return grad_sampler(ex.first) or grad_sampler(ex.second)
        return ex if isinstance(ex, (AutoMALA, MALA)) else None
    gs = grad_sampler(shared.explorer)
    if gs is not None:
        eng.set_explorer_adaptation(gs.step_size, gs.estimated_target_std_deviations)
    from .pt import InterpolatingPath, StabilizedPT, GaussianReference
    temp = shared.tempering
    leg = temp.variational_leg if isinstance(temp, StabilizedPT) else temp
    if isinstance(leg.path, InterpolatingPath) and isinstance(leg.path.ref, GaussianReference):     # an activated variational reference
        ref = leg.path.ref
        uses = np.array([1 if (not isinstance(temp, StabilizedPT) or c < temp.n_var) else 0 for c in range(eng.N)], dtype=np.int32)
        eng.set_variational_reference(ref.mean, ref.standard_deviation, uses)
        pt.inputs.variational = ref
    # the reference gives the resumed run a fresh exec folder holding links to the old checkpoints (checkpoint.jl:36-52);
    # here the run simply keeps checkpointing into the folder it was resumed from
    pt.exec_folder = source_exec_folder if inputs.checkpoint else None
    return pt


def increment_n_rounds(pt, increment):
    """increment_n_rounds!(pt, k) (checkpoint.jl:166-172) for a live PT: device buffers are sized by n_rounds, so the
    engine is rebuilt from the current replicas (same arithmetic from here on)."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        write_checkpoint(pt, tmp)                      # explicit folder: pt.exec_folder stays what it was
        new = load_checkpoint(tmp, n_rounds_increment=increment)
    new.exec_folder = getattr(pt, "exec_folder", None)
    return new
