"""The host-side mirror of the reference's round loop (pigeons_amd/pt.py: run_one_round, reduce_recorders, adapt,
adapt_explorer, two-leg tempering, stepping stone, checkpoint) driven over an ORACLE-backed engine on the CPU and compared
with the oracle's own C implementation of the same loop.  The engine seam is exactly the one the HIP engine sits behind
(PT(engine_factory=...)); the product path never takes this factory."""
import numpy as np
import pytest

import oracle as O


class OracleEngine(O.OracleShard):
    """A world_size-1 oracle shard with the single-engine methods of pigeons_amd.engine.Engine."""

    def run_scans(self, first, n):
        z = np.zeros(4)
        for s in range(first, first + n):
            self.explore(s); self.swap_begin(s); self.swap_finish(s, z)

    def index_process(self):
        rep, ch = self.index_process_shard()
        if rep.size == 0:
            return None
        out = np.zeros((self.N, rep.shape[0]), dtype=np.int64)
        t = np.repeat(np.arange(rep.shape[0])[:, None], rep.shape[1], axis=1)
        out[rep, t] = ch
        return out

    def automala_stats(self):
        return self.am_stats()

    def set_explorer_adaptation(self, step_size, target_std=None):
        O.OraclePT.set_explorer_adaptation(self, step_size, target_std)

    def set_variational_reference(self, mean, std, uses):
        raise NotImplementedError("the oracle refits its GaussianReference itself (po_end_round)")


@pytest.fixture(scope="module")
def P():
    import pigeons_amd
    return pigeons_amd


CASES = {
    "slice": (lambda P: dict(target=P.toy_mvn_target(5), n_chains=6, explorer=P.SliceSampler()), dict(dim=5, n_chains=6, explorer=O.EXPLORER_SLICE)),
    "automala": (lambda P: dict(target=P.toy_mvn_target(6), n_chains=5, explorer=P.AutoMALA()),
                 dict(dim=6, n_chains=5, explorer=O.EXPLORER_AUTOMALA, am_preconditioner=2)),
    "compose": (lambda P: dict(target=P.toy_mvn_target(3), n_chains=4, explorer=P.Compose(P.SliceSampler(), P.AutoMALA())),
                dict(dim=3, n_chains=4, explorer=O.EXPLORER_SLICE, explorer2=O.EXPLORER_AUTOMALA, am_preconditioner=2)),
    "two_legs": (lambda P: dict(target=P.toy_mvn_target(4), n_chains=5, n_chains_variational=4, variational=None, explorer=P.SliceSampler()),
                 dict(dim=4, n_chains=5, n_chains_variational=4, explorer=O.EXPLORER_SLICE)),
    "funnel_mala": (lambda P: dict(target=P.Funnel(4), reference=P.ScaledPrecisionNormalLogPotential(1.0 / 9.0, 4), n_chains=5,
                                   explorer=P.MALA(step_size=0.3)),
                    dict(dim=4, n_chains=5, explorer=O.EXPLORER_MALA, am_step_size=0.3, am_preconditioner=2, target=O.TARGET_FUNNEL, p0=1.0 / 9.0)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_python_round_loop_equals_the_oracles_round_loop(P, name):
    mk, okw = CASES[name]
    rounds = 7
    rec = [P.round_trip, P.index_process, P.log_sum_ratio, P.online, P.energy_ac1]
    pt = P.PT(P.Inputs(n_rounds=rounds, record=rec, show_report=False, **mk(P)), engine_factory=OracleEngine)
    ref = O.OraclePT(record_online=1, record_energy_ac1=1, **okw)
    for _ in range(rounds):
        assert P.next_round(pt)
        red = P.run_one_round(pt); P.adapt(pt, red); P.report(pt)
        ref.run_round()
        assert np.array_equal(red.index_process, ref.index_process())
        assert red.round_trip == ref.round_trip()
        np.testing.assert_allclose(pt.shared.tempering.schedule.grids, ref.schedule(), rtol=1e-12)
        np.testing.assert_allclose(P.stepping_stone_pair(pt), ref.stepping_stone_pair(), rtol=1e-12)
        if name != "funnel_mala" or True:
            np.testing.assert_allclose(P.global_barrier(pt), ref.global_barrier(), rtol=1e-12)
        if name == "two_legs":
            np.testing.assert_allclose(P.global_barrier_variational(pt), ref.global_barrier_variational(), rtol=1e-12)
        if name in ("automala", "compose"):
            ex = pt.shared.explorer
            ex = ex if hasattr(ex, "step_size") else (ex.first if hasattr(ex.first, "step_size") else ex.second)
            np.testing.assert_allclose(ex.step_size, ref.step_size(), rtol=1e-13)
            np.testing.assert_allclose(ex.estimated_target_std_deviations, ref.target_std(), rtol=1e-10)
    x, chain, rng = pt.replicas.states(); xr, cr, rr = ref.states()
    assert np.array_equal(chain, cr) and np.array_equal(rng, rr)
    np.testing.assert_allclose(x, xr, rtol=1e-13, atol=1e-300)


def test_checkpoint_roundtrip_on_the_cpu(P, tmp_path):
    """write_checkpoint / load_checkpoint (reference src/pt/checkpoint.jl) over the oracle-backed engine: resume == uninterrupted."""
    mk = lambda n: P.Inputs(target=P.toy_mvn_target(4), n_chains=5, n_rounds=n, explorer=P.Compose(P.SliceSampler(), P.AutoMALA()),
                            record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False, checkpoint=True)
    import dataclasses
    straight = P.pigeons(P.PT(dataclasses.replace(mk(6), checkpoint=False), engine_factory=OracleEngine))
    folder = str(tmp_path / "exec")
    P.pigeons(P.PT(mk(3), engine_factory=OracleEngine), exec_folder=folder)
    assert P.latest_checkpoint_folder(folder) == 3
    resumed = P.pigeons(P.load_checkpoint(folder, n_rounds_increment=3, engine_factory=OracleEngine))
    assert np.array_equal(straight.reduced_recorders.index_process, resumed.reduced_recorders.index_process)
    assert np.array_equal(straight.shared.tempering.schedule.grids, resumed.shared.tempering.schedule.grids)
    for a, b in zip(straight.replicas.states(), resumed.replicas.states()):
        assert np.array_equal(a, b)
