"""Throughput of every BASELINE config shape on one GPU (not the driver's bench; see bench.py for the metric)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import pigeons_amd as P

def run(name, inputs, scans=16, warm=4):
    pt = P.PT(inputs)
    e = pt.replicas
    e.run_scans(1, warm)
    t = time.perf_counter(); e.run_scans(1, scans); dt = time.perf_counter() - t
    print("%-58s %9.3f ms/scan  %12.0f replica-steps/s" % (name, dt / scans * 1e3, inputs.n_chains * scans / dt), flush=True)

rec = [P.round_trip, P.log_sum_ratio]
run("C1 toy_mvn(2) N=10 SliceSampler", P.Inputs(target=P.toy_mvn_target(2), n_chains=10, explorer=P.SliceSampler(), record=rec, n_rounds=20, show_report=False), 256, 16)
run("C2 toy_mvn(1024) N=256 SliceSampler", P.Inputs(target=P.toy_mvn_target(1024), n_chains=256, explorer=P.SliceSampler(), record=rec, n_rounds=20, show_report=False))
run("metric toy_mvn(1024) N=1024 SliceSampler", P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, explorer=P.SliceSampler(), record=rec, n_rounds=20, show_report=False))
run("toy_mvn(1024) N=1024 ToyExplorer (HBM-bound explore)", P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, record=rec, n_rounds=20, show_report=False), 256, 16)
run("toy_mvn(4096) N=8192 ToyExplorer", P.Inputs(target=P.toy_mvn_target(4096), n_chains=8192, record=rec, n_rounds=20, show_report=False), 64, 8)
run("C3 funnel(128) N=1024 AutoMALA", P.Inputs(target=P.Funnel(128), reference=P.ScaledPrecisionNormalLogPotential(1/9., 128), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=20, show_report=False), 64, 8)
# (at d = 128 the coordinate-wise sampler started from zeros(d) walks y = z[1] towards -(d-1) * 4.5 and sigma = exp(y/2) underflows: the reference's
#  own procedure ends in "Maximum number of iterations reached" there -- the oracle does too; d = 32 is stable)
run("funnel(32) N=1024 SliceSampler (full log potential per proposal)", P.Inputs(target=P.Funnel(32), reference=P.ScaledPrecisionNormalLogPotential(1/9., 32), n_chains=1024, explorer=P.SliceSampler(), record=rec, n_rounds=20, show_report=False), 16, 4)
run("C4 shard: toy_mvn(4096) N=1024 SliceSampler", P.Inputs(target=P.toy_mvn_target(4096), n_chains=1024, explorer=P.SliceSampler(), record=rec, n_rounds=20, show_report=False), 8, 2)
run("C5 shard: Ising 256x256 N=512 IsingMetropolis", P.Inputs(target=P.IsingLogPotential(1.0, 256), n_chains=512, record=rec, n_rounds=20, show_report=False), 4, 1)
if os.environ.get("BC_LANGEVIN_LARGE", "1") != "0":
    # the largest register layouts of the Langevin kernel (E = 8 / 16 blocks per replica; the funnel instantiation at E = 16 keeps part of its state in scratch)
    run("toy_mvn(512) N=1024 AutoMALA", P.Inputs(target=P.toy_mvn_target(512), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=20, show_report=False), 16, 4)
    run("toy_mvn(1024) N=1024 AutoMALA", P.Inputs(target=P.toy_mvn_target(1024), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=20, show_report=False), 16, 4)
    run("funnel(512) N=1024 AutoMALA", P.Inputs(target=P.Funnel(512), reference=P.ScaledPrecisionNormalLogPotential(1/9., 512), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=20, show_report=False), 16, 4)
    run("funnel(1024) N=1024 AutoMALA", P.Inputs(target=P.Funnel(1024), reference=P.ScaledPrecisionNormalLogPotential(1/9., 1024), n_chains=1024, explorer=P.AutoMALA(), record=rec, n_rounds=20, show_report=False), 16, 4)
