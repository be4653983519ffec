"""Per-kernel register / scratch / LDS / occupancy table of libpte.so, from hipcc's own remark pass (-Rpass-analysis=kernel-resource-usage) over
the product's translation units with the shipped flags (tools/codegen.py, cached).  tests/test_codegen_frozen.py asserts on the same numbers.
Usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codegen as C


def main():
    units = C.compile_units(extra=tuple(sys.argv[1:]))
    res = C.resources(units)
    keys = [k for k, _ in C._RES_KEYS]
    for _, cmd, _, _ in units:
        print("# " + " ".join(cmd[1:]))
    print("%-78s " % "kernel" + " ".join(keys))
    for name in res["__order__"]:
        print("%-78s " % name[:78] + " ".join("%*s" % (len(k), "?" if res[name][k] is None else res[name][k]) for k in keys))


if __name__ == "__main__":
    main()
