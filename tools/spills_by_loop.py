"""Where a kernel's SGPR spills execute.  hipcc spills scalar registers into lanes of a VGPR: the spill is a v_writelane_b32 with a
constant lane, the reload a v_readlane_b32 with a constant lane from one of the registers such writes go to (other v_readlane -- a chase hop
with an SGPR lane select, a reduction read out of lane 0 / 16 / 32 / 48 / 63 -- are listed as "readlane"; v_readfirstlane is never a spill).
Takes the kernels named on the command line (substrings of the mangled name; default: the AutoMALA instantiation of config 3, the Ising kernel of
config 5 and the metric's slice kernel) out of the product's assembly (tools/codegen.py: the shipped flags, cached) and, per loop of the kernel
(LLVM's "Loop Header: Depth=N" / "in Loop: Header=..." block comments), prints the instruction counts of the blocks that belong to that loop and
to no deeper one, with the spill writes and reloads among them.  A spill that matters is one in an innermost loop; one at depth 0 / 1 runs once
per launch / per outer step.  tests/test_codegen_frozen.py asserts on the same numbers.
Usage: python tools/spills_by_loop.py [kernel-substring ...] > profiles/rNN_spills_by_loop.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codegen as C
DEFAULT = ["k_explore_automalaILi2ELi2ELb0ELb1E", "k_explore_ising_specILb0E", "k_explore_slice8ILi4ELi9E"]


def main():
    units = C.compile_units()
    lines = C.asm_lines(units)
    for _, cmd, _, _ in units:
        print("# " + " ".join(cmd[1:]))
    print("# per loop: the blocks whose INNERMOST loop it is (a block of a nested loop is counted with the nested loop only)\n")
    for sub in (sys.argv[1:] or DEFAULT):
        name, body = C.kernel_body(lines, sub)
        print("## %s" % C.demangle([name])[0])
        print("%-5s %-12s %6s %6s %6s %5s %5s %7s | %11s %12s %8s %6s" % ("depth", "loop", "blocks", "VALU", "SALU", "LDS", "VMEM", "scratch", "spill-write", "spill-reload", "readlane", "s_nop"))
        tw = tr = 0
        for (depth, head), L in sorted(C.loops(body).items(), key=lambda kv: (kv[0][0], kv[0][1] or "")):
            print("%-5d %-12s %6d %6d %6d %5d %5d %7d | %11d %12d %8d %6d" % (depth, head or "-", L["n"], L["v"], L["s"], L["l"], L["m"], L["scratch"], L["w"], L["r"], L["dyn"], L["nop"]))
            tw += L["w"]; tr += L["r"]
        print("total spill writes %d, reloads %d (static)\n" % (tw, tr))


if __name__ == "__main__":
    main()
