"""Which lane instructions (v_readlane / v_writelane / v_readfirstlane: SGPR spills and reloads look like the first two) the round loop of the
default SliceSampler kernel executes.  Compiles pte.hip to gfx950 assembly with the shipped flags, takes k_explore_slice8<NLU=4, BS=9>
(d = 1024), finds the round loop (the first loop of depth 3) and prints its basic blocks in layout order with their instruction counts
and every lane instruction in them.  The likely path is laid out contiguously from the loop header to the back edge: those blocks are
marked HOT; everything behind the back edge (window refill, lane 0's budget-free continuations, the exact sequential procedure) is COLD.
Usage: python tools/round_loop_lanes.py > profiles/rNN_slice8_round_loop.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN3pte16k_explore_slice8ILi4ELi9EEEvNS_9EngineDevENS_11SliceParamsE"


def main():
    out = os.path.join(tempfile.gettempdir(), "pte_round_loop.s")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g                     # the flags the product gives pte.hip
    src, unit_flags = g.UNITS[0]
    cmd = [g.HIPCC, *[f for f in g.FLAGS if f != "-fPIC"], *unit_flags, "--cuda-device-only", "-S", *sys.argv[1:], "-o", out, os.path.join(g.CSRC, src)]
    if not (os.environ.get("ROUND_LOOP_REUSE") and os.path.exists(out)):
        subprocess.run(cmd, check=True, capture_output=True)
    lines = open(out).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    body = lines[a:b + 1]
    h = next(i for i, l in enumerate(body) if "Loop Header: Depth=3" in l)
    # the loop's header label is the last label before the "Loop Header" comment
    hl = next(i for i in range(h, 0, -1) if re.match(r"^\.LBB\d+_\d+:", body[i]))
    header = body[hl].split(":")[0]
    # the loop ends where a block comment no longer says "in Loop: Header=<header>"
    hname = header[2:]                     # block comments say "Header=BB35_50" for the label .LBB35_50
    blocks, cur, name = [], None, None
    for i in range(hl, len(body)):
        l = body[i]
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            nm = m.group(1) or ("bb." + m.group(2))
            in_loop = (nm == header) or ("Header=" + hname in l) or any("Header=" + hname in body[j] or "Parent Loop " + hname in body[j] for j in range(i, min(i + 3, len(body))))
            if cur: blocks.append(cur)
            if not in_loop and nm != header:
                cur = None
                break
            cur = {"name": nm, "v": 0, "s": 0, "l": 0, "m": 0, "lane": [], "back": False}
            continue
        if cur is None: continue
        t = l.strip()
        if re.match(r"^v_(readlane|writelane|readfirstlane)", t): cur["lane"].append(t.split(";")[0].strip())
        if re.match(r"^v_", t): cur["v"] += 1
        elif re.match(r"^s_", t):
            cur["s"] += 1
            if re.match(r"^s_cbranch\w* " + re.escape(header) + r"\b", t) or re.match(r"^s_branch " + re.escape(header) + r"\b", t): cur["back"] = True
        elif re.match(r"^ds_", t): cur["l"] += 1
        elif re.match(r"^(global|scratch|buffer|flat)_", t): cur["m"] += 1
    if cur: blocks.append(cur)
    print("# " + " ".join(cmd[1:]))
    print("# round loop of k_explore_slice8<4, 9>: header %s; its blocks in layout order up to the first nested (cold) loop" % header)
    hot = True
    tot = [0, 0, 0, 0]
    nl_hot = nl_cold = 0
    for bl in blocks:
        tag = "HOT " if hot else "cold"
        print("%s %-12s VALU %4d  SALU %3d  LDS %2d  VMEM %2d%s" % (tag, bl["name"], bl["v"], bl["s"], bl["l"], bl["m"], "   <- back edge" if bl["back"] else ""))
        for x in bl["lane"]: print("         %s" % x)
        if hot:
            for k, key in enumerate("vslm"): tot[k] += bl[key]
            nl_hot += len(bl["lane"])
        else:
            nl_cold += len(bl["lane"])
        if bl["back"]: hot = False
    print("# HOT path per round: %d VALU + %d SALU + %d LDS + %d VMEM = %d instructions; lane instructions: %d on the hot path (the chase's v_readlane), %d in the cold blocks" %
          (tot[0], tot[1], tot[2], tot[3], sum(tot), nl_hot, nl_cold))
    print("# v_writelane on the hot path: %d" % sum(1 for bl in blocks[:next((i for i, b_ in enumerate(blocks) if b_["back"]), len(blocks)) + 1] for x in bl["lane"] if x.startswith("v_writelane")))


if __name__ == "__main__":
    main()
