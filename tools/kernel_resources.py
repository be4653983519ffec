"""Per-kernel register / scratch / LDS / occupancy table of libpte.so, from hipcc's own remark pass
(-Rpass-analysis=kernel-resource-usage).  Usage: python tools/kernel_resources.py [-DPTE_TEST_KERNELS ...] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pigeons.jl_amd", "csrc", "pte.hip")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-align-all-nofallthru-blocks=6", "-fPIC", "-shared",
           "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "-o", "/tmp/libpte_resources.so", SRC, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    keys = [("vgpr", r" VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"), ("spilled_vgpr", r"VGPRs Spill"), ("spilled_sgpr", r"SGPRs Spill"),
            ("scratch_B_per_lane", r"ScratchSize \[bytes/lane\]"), ("waves_per_simd", r"Occupancy \[waves/SIMD\]"),
            ("lds_B", r"LDS Size \[bytes/block\]")]
    print("# " + " ".join(cmd[1:]))
    print("%-78s " % "kernel" + " ".join("%s" % k for k, _ in keys))
    for b in blocks:
        name = b.split("\n")[0].split(" [")[0].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        name = re.sub(r"\(.*\)$", "", name).replace("void pte::", "")
        vals = []
        for k, pat in keys:
            m = re.search(pat + r": (\S+)", b)
            vals.append(m.group(1) if m else "?")
        print("%-78s " % name[:78] + " ".join("%*s" % (len(k), v) for (k, _), v in zip(keys, vals)))


if __name__ == "__main__":
    main()
