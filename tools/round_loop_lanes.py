"""Which lane instructions (v_readlane / v_writelane / v_readfirstlane: SGPR spills and reloads look like the first two) the round loop of the
default SliceSampler kernel executes.  Takes k_explore_slice8<NLU=4, BS=9> (d = 1024; another kernel: a substring of its mangled name as the
argument) out of the product's assembly (tools/codegen.py: the shipped flags, cached), finds the round loop (the first loop of depth 3) and prints
its basic blocks in layout order with their instruction counts and every lane instruction in them.  The likely path is laid out contiguously from
the loop header to the back edge: those blocks are HOT; everything behind the back edge (window refill, lane 0's budget-free continuations, the
exact sequential procedure) is COLD.  tests/test_codegen_frozen.py asserts on the same numbers.
Usage: python tools/round_loop_lanes.py [kernel-substring] > profiles/rNN_slice8_round_loop.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codegen as C


def main():
    sub = sys.argv[1] if len(sys.argv) > 1 else "k_explore_slice8ILi4ELi9E"
    units = C.compile_units()
    name, body = C.kernel_body(C.asm_lines(units), sub)
    depth = 4 if "k_scans" in sub else 3              # (the fused kernels wrap the body in the scan loop: one level deeper)
    header = next(h for d, h in C.loop_headers(body) if d == depth)
    hot = C.hot_path(body, header)
    hot_names = set(b["name"] for b in hot)
    for _, cmd, _, _ in units:
        print("# " + " ".join(cmd[1:]))
    print("# %s\n# round loop: header %s; its blocks in layout order" % (C.demangle([name])[0], header))
    nl_cold = 0
    for b in C.blocks_of(body):
        in_loop = b["name"] == ".L" + header or b["head"] == header or header in b.get("parents", [])
        if not in_loop:
            continue
        is_hot = b["name"] in hot_names
        print("%s %-12s VALU %4d  SALU %3d  LDS %2d  VMEM %2d%s" % ("HOT " if is_hot else "cold", b["name"], b["v"], b["s"], b["l"], b["m"],
                                                                     "   <- back edge" if is_hot and b is hot[-1] else ""))
        for x in b["lane"]:
            print("         %s" % x)
        if not is_hot:
            nl_cold += len(b["lane"])
    t = C.totals(hot)
    print("# HOT path per round: %d VALU + %d SALU + %d LDS + %d VMEM = %d instructions in %d blocks; lane instructions: %d on the hot path (the chase's v_readlane), %d in the cold blocks"
          % (t["v"], t["s"], t["l"], t["m"], t["instructions"], t["blocks"], sum(len(b["lane"]) for b in hot), nl_cold))
    print("# on the hot path: v_writelane (spill writes) %d, spill reloads %d, scratch accesses %d" % (t["w"], t["r"], t["scratch"]))


if __name__ == "__main__":
    main()
