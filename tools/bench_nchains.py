"""SliceSampler throughput of one MI355X as a function of the number of chains (d = 1024): waves per SIMD = N / 1024."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import pigeons_amd as P
for N in (256, 512, 1024, 2048, 4096, 8192):
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False,
                       record=[P.round_trip, P.log_sum_ratio]))
    e = pt.replicas
    e.run_scans(1, 4)
    t = time.perf_counter(); e.run_scans(1, 16); dt = time.perf_counter() - t
    print("N=%5d  %.3f ms/scan  %9.0f replica-steps/s" % (N, dt / 16 * 1e3, N * 16 / dt), flush=True)
