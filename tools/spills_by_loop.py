"""Where a kernel's SGPR spills execute.  hipcc spills scalar registers into lanes of a VGPR: the spill is a v_writelane_b32 with a
constant lane, the reload a v_readlane_b32 with a constant lane from one of the registers such writes go to (other v_readlane -- a chase hop
with an SGPR lane select, a reduction read out of lane 0 / 16 / 32 / 48 / 63 -- are listed as "readlane"; v_readfirstlane is never a spill).  This script compiles pte.hip to gfx950 assembly with the shipped flags, takes the kernels
named on the command line (substrings of the mangled name; default: the AutoMALA instantiation of config 3 and the Ising kernel of config 5)
and, per loop of the kernel (LLVM's "Loop Header: Depth=N" / "in Loop: Header=..." block comments), prints the instruction counts of the
blocks that belong to that loop and to no deeper one, with the spill writes and reloads among them.  A spill that matters is one in an
innermost loop; one at depth 0 / 1 runs once per launch / per outer step.
Usage: python tools/spills_by_loop.py [kernel-substring ...] > profiles/rNN_spills_by_loop.txt      (ROUND_LOOP_REUSE=1: reuse /tmp/pte_round_loop.s)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["k_explore_automalaILi2ELi2ELb0ELb1E", "k_explore_ising_specILb0E", "k_explore_slice8ILi4ELi9E"]


def kernel_body(lines, sub):
    a = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3pte\w*:", l) and sub in l)
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i] and not any("s_endpgm" in lines[j] for j in range(i + 1, min(i + 400, len(lines))) if lines[j].startswith("\t.section") is False and False))
    # the kernel ends at its .Lfunc_end label
    e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[a].rstrip(":"), lines[a:e]


def analyse(name, body):
    # spill VGPRs: the targets of v_writelane_b32 with a constant lane; a reload is a constant-lane v_readlane FROM one of them (a
    # constant-lane v_readlane of any other register reads out a reduction: lanes 0 / 16 / 32 / 48 / 63 of a freshly computed value)
    spill_regs = set(m.group(1) for l in body for m in [re.match(r"^\s*v_writelane_b32 (v\d+), s\d+, \d+", l)] if m)
    # blocks: (label, loop header it belongs to (innermost) or None, depth)
    blocks, cur = [], None
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            nm = m.group(1) or ("bb." + m.group(2))
            ctx = " ".join(body[i:i + 6]) if True else ""
            # the comment block following a label: "; =>This Loop Header: Depth=1", ";   in Loop: Header=BB5_3 Depth=2", "; Parent Loop BB5_1 Depth=1"
            com = []
            for j in range(i, min(i + 12, len(body))):
                if j > i and not body[j].lstrip().startswith(";") and "Loop" not in body[j]: break
                com.append(body[j])
            com = " ".join(com)
            mh = re.search(r"Loop Header: Depth=(\d+)", com)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", com)
            if mh: head, depth = nm.lstrip(".L"), int(mh.group(1))
            elif mi: head, depth = mi.group(1), int(mi.group(2))
            else: head, depth = None, 0
            cur = {"name": nm, "head": head, "depth": depth, "v": 0, "s": 0, "l": 0, "m": 0, "w": 0, "r": 0, "nop": 0, "dyn": 0}
            blocks.append(cur)
            continue
        if cur is None:
            cur = {"name": "entry", "head": None, "depth": 0, "v": 0, "s": 0, "l": 0, "m": 0, "w": 0, "r": 0, "nop": 0, "dyn": 0}
            blocks.append(cur)
        t = l.strip()
        if re.match(r"^v_writelane_b32 v\d+, s\d+, \d+", t): cur["w"] += 1
        elif re.match(r"^v_readlane_b32 s\d+, (v\d+), \d+", t) and re.match(r"^v_readlane_b32 s\d+, (v\d+), \d+", t).group(1) in spill_regs: cur["r"] += 1
        elif re.match(r"^v_readlane_b32", t): cur["dyn"] += 1
        if re.match(r"^v_", t): cur["v"] += 1
        elif re.match(r"^s_nop", t): cur["nop"] += 1; cur["s"] += 1
        elif re.match(r"^s_", t): cur["s"] += 1
        elif re.match(r"^ds_", t): cur["l"] += 1
        elif re.match(r"^(global|scratch|buffer|flat)_", t): cur["m"] += 1
    loops = {}
    for b in blocks:
        k = (b["depth"], b["head"])
        L = loops.setdefault(k, {"n": 0, "v": 0, "s": 0, "l": 0, "m": 0, "w": 0, "r": 0, "nop": 0, "dyn": 0})
        L["n"] += 1
        for f in ("v", "s", "l", "m", "w", "r", "nop", "dyn"): L[f] += b[f]
    print("## %s" % name)
    print("%-5s %-12s %6s %6s %6s %5s %5s | %11s %12s %8s %6s" % ("depth", "loop", "blocks", "VALU", "SALU", "LDS", "VMEM", "spill-write", "spill-reload", "readlane", "s_nop"))
    tw = tr = 0
    for (depth, head), L in sorted(loops.items(), key=lambda kv: (kv[0][0], kv[0][1] or "")):
        print("%-5d %-12s %6d %6d %6d %5d %5d | %11d %12d %8d %6d" % (depth, head or "-", L["n"], L["v"], L["s"], L["l"], L["m"], L["w"], L["r"], L["dyn"], L["nop"]))
        tw += L["w"]; tr += L["r"]
    print("spill registers: %s; total spill writes %d, reloads %d (static)\n" % (" ".join(sorted(spill_regs, key=lambda r: int(r[1:]))), tw, tr))


def main():
    # the product's two translation units with the flags __graft_entry__.build_hip gives them
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    lines = []
    for src, unit_flags in g.UNITS:
        out = os.path.join(tempfile.gettempdir(), "pte_loops_%s.s" % os.path.splitext(src)[0])
        cmd = [g.HIPCC, *[f for f in g.FLAGS if f != "-fPIC"], *unit_flags, "--cuda-device-only", "-S", "-o", out, os.path.join(g.CSRC, src)]
        if not (os.environ.get("ROUND_LOOP_REUSE") and os.path.exists(out)):
            subprocess.run(cmd, check=True, capture_output=True)
        lines += open(out).read().split("\n")
        print("# " + " ".join(cmd[1:]))
    print("# per loop: the blocks whose INNERMOST loop it is (a block of a nested loop is counted with the nested loop only)\n")
    for sub in (sys.argv[1:] or DEFAULT):
        name, body = kernel_body(lines, sub)
        analyse(name, body)


if __name__ == "__main__":
    main()
