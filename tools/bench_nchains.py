"""SliceSampler throughput of one MI355X as a function of the number of chains: waves per SIMD = N / 1024.
d = 1024 (the metric's dimension) and d = 4096 (BASELINE configs[3]); the last line of the d = 4096 table, N = 8192 on ONE GPU,
is the 1-GPU anchor of the north star's strong-scaling clause (8192 chains, 1 -> 8 GPUs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
import _variant; _variant.apply()          # PTE_LIB=<path>: a tuning build (tools/_variant.py); the product itself never reads the variable
import pigeons_amd as P
for d, Ns, scans in ((1024, (256, 512, 1024, 2048, 4096, 8192), 16), (4096, (1024, 2048, 4096, 8192), 6)):
    for N in Ns:
        pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False,
                           record=[P.round_trip, P.log_sum_ratio]))
        e = pt.replicas
        e.run_scans(1, 3)
        t = time.perf_counter(); e.run_scans(1, scans); dt = time.perf_counter() - t
        print("d=%4d N=%5d  %8.3f ms/scan  %9.0f replica-steps/s" % (d, N, dt / scans * 1e3, N * scans / dt), flush=True)
        del pt, e
