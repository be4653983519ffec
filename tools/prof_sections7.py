"""Debug-only: -DPTE_PROFILE_SECTIONS build; where k_explore_slice7's cycles go."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pigeons.jl_amd")]
lib = os.path.join(ROOT, "gpurun_out", "libpte_prof.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
if os.environ.get("PROF_LIB"):       # a library built beforehand (tools/build_variant.sh prof -DPTE_PROFILE_SECTIONS): nothing to compile on the GPU box
    lib = os.environ["PROF_LIB"]
else:
  subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                  "-Wno-unused-value", "-DPTE_PROFILE_SECTIONS", "-DPTE_TEST_KERNELS", *(["-DPTE_DEBUG_S7"] if os.environ.get("S7DBG") else []), "-o", lib, os.path.join(ROOT, "pigeons.jl_amd/csrc/pte.hip"), "-ldl"], check=True)
from pigeons_amd import _lib
_lib.LIB_PATH = lib
DK = int(os.environ.get("S_IMPL", "7"))
import numpy as np
import pigeons_amd as P
N, d = (64 if os.environ.get('S_IMPL') == '8' else 128), 1024
pt = P.PT(P.Inputs(target=P.toy_mvn_target(d), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False,
                   record=[P.online, P.log_sum_ratio]))
assert DK in (7, 8)
if DK == 7:
    pt = P.PT(pt.inputs, debug_kernel=7)          # (the engine loads _lib.LIB_PATH: the profile build holds every generation)
scans = 8
pt.replicas.run_scans(1, scans)
pt.replicas.reduce()
m, v, n = pt.replicas.online()
W_ = 16 if os.environ.get('S_IMPL') == '8' else 8
cnt = (v * (n - 1)).reshape(-1)[:W_ * N].reshape(N, W_) / scans
names = ["head_dbl", "shrink", "accept", "rounds", "coords_spec", "chase", "fallbacks", "fallback_cyc"]
for ch in ((1, 16, 32, 48, 63) if N == 64 else (1, 32, 64, 127)):
    c = cnt[ch]
    tot = c[0] + c[1] + c[2] + c[5] + c[7]
    print("chain %3d: total %.2fM | " % (ch, tot / 1e6) + "  ".join("%s %.0f" % (nm, x) for nm, x in zip(names, c)))
    print("   per round: head+doubling %.0f  shrink %.0f  accept %.0f  chase %.0f cyc; coords/round %.2f; fallbacks %.0f at %.0f cyc each"
          % (c[0] / c[3], c[1] / c[3], c[2] / c[3], c[5] / c[3], c[4] / c[3], c[6], c[7] / max(c[6], 1)))
    if W_ == 16:
        print("   rounds ending: all 5 levels %.0f | slow-path E %.0f, doubling over budget / ends inside %.0f, no proposal inside within the budget %.0f, acceptance check %.0f, sliver %.0f, successor outside its window %.0f, end of block %.0f" % tuple(c[8:16]))
