"""Debug-only: build libpte with -DPTE_DEBUG_COUNTERS and print the slice-kernel path statistics."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
lib = os.path.join(ROOT, "gpurun_out", "libpte_dbg.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                "-Wno-unused-value", "-DPTE_DEBUG_COUNTERS", "-o", lib, os.path.join(ROOT, "pigeons.jl_amd/csrc/pte.hip")], check=True)
from pigeons_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
import pigeons_amd as P
N, d = 128, 1024      # d >= 5N so the counters fit in the on_m2 buffer
M = int(os.environ.get("PTE_SLICE_M", "4"))
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False,
                   record=[P.online, P.log_sum_ratio]))
scans = 8
pt.replicas.run_scans(1, scans)
import ctypes as C
pt.replicas.reduce()
m, v, n = pt.replicas.online()
# on_var = m2/(n-1) with n = scans (target chain records once per scan)
cnt = (v * (n - 1)).reshape(-1)[:5 * N].reshape(N, 5) / scans
tot = 3 * d
print("per replica-step, averaged over chains 1..N-1 (of %d coordinate updates):" % tot)
c = cnt[1:].mean(0)
print("fast accepted %.1f (%.1f%%)  continuation %.1f (%.1f%%)  doubling %.1f (%.1f%%)  fallback evals %.1f  doubling steps %.1f"
      % (c[0], 100*c[0]/tot, c[1], 100*c[1]/tot, c[2], 100*c[2]/tot, c[3], c[4]))
for ch in (1, N//4, N//2, N-1):
    print("chain", ch, np.round(cnt[ch], 1))
