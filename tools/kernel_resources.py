"""Per-kernel register / scratch / LDS / occupancy table of libpte.so, from hipcc's own remark pass
(-Rpass-analysis=kernel-resource-usage).  Usage: python tools/kernel_resources.py [-DPTE_TEST_KERNELS ...] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pigeons.jl_amd", "csrc", "pte.hip")


def main():
    # the product's two translation units with the flags __graft_entry__.build_hip gives them
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    err = ""
    cmds = []
    for src, unit_flags in g.UNITS:
        cmd = [g.HIPCC, *g.FLAGS, *unit_flags, "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "-c", "-o", "/tmp/libpte_resources.o", os.path.join(g.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        err += r.stderr
        cmds.append(" ".join(cmd[1:]))
    blocks = re.split(r"remark: [^\n]*Function Name: ", err)[1:]
    keys = [("vgpr", r" VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"), ("spilled_vgpr", r"VGPRs Spill"), ("spilled_sgpr", r"SGPRs Spill"),
            ("scratch_B_per_lane", r"ScratchSize \[bytes/lane\]"), ("waves_per_simd", r"Occupancy \[waves/SIMD\]"),
            ("lds_B", r"LDS Size \[bytes/block\]")]
    for c in cmds:
        print("# " + c)
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
