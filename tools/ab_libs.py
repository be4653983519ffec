"""Two builds of libpte (build_variants/libpte_v_<a>.so, ..._<b>.so) on the same seeded runs: a hash of every state, stream position, index process and
recorder must agree (each build runs in its own process: one library per process).  usage: python tools/ab_libs.py a b"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, os, sys
sys.path[:0] = [%r, os.path.join(%r, "pigeons.jl_amd"), os.path.join(%r, "tools")]
import _variant; _variant.apply()          # PTE_LIB of the child's environment (tools/_variant.py)
import numpy as np, pigeons_amd as P
h = hashlib.sha256()
for (N, d), expl, seed in [((96, 1024), P.SliceSampler(), 1), ((16, 4096), P.SliceSampler(), 2), ((40, 1024), P.SliceSampler(w=1.0), 3), ((64, 4096), P.ToyExplorer(), 4), ((33, 1024), P.ToyExplorer(), 5)]:
    pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=4, seed=seed, explorer=expl, record=[P.round_trip, P.index_process, P.log_sum_ratio], show_report=False))
    for _ in range(4):
        P.next_round(pt); red = P.run_one_round(pt); P.adapt(pt, red)
        h.update(red.index_process.tobytes()); h.update(red.swap_acceptance_pr[0].tobytes())
    for a in pt.replicas.states(): h.update(np.ascontiguousarray(a).tobytes())
print(h.hexdigest())
''' % (ROOT, ROOT, ROOT)
out = []
for v in sys.argv[1:3]:
    env = dict(os.environ, PTE_LIB=os.path.join(ROOT, "build_variants", "libpte_v_%s.so" % v))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    out.append(r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "FAILED: " + r.stderr[-300:])
    print(v, out[-1])
print("identical" if out[0] == out[1] and not out[0].startswith("FAILED") else "DIFFERENT")
