#!/usr/bin/env python3
"""Install Julia's own ziggurat tables from the live-reference fixture.

    julia tools/gen_golden.jl > tests/golden/reference_pigeons.json && python tools/import_tables.py

Reads tests/golden/reference_pigeons.json["tables"] (Random.ki / wi / fi / ke / we / fe as bit patterns, dumped by
tools/gen_golden.jl) and rewrites the product's device header pigeons.jl_amd/csrc/zig_tables.h with exactly those 6 x 256
entries, reporting every entry that differs from the re-derived table it replaces; then rebuilds libpte.  The CPU oracle
has no table header: tests/oracle.py installs the same fixture tables into it when it loads (po_zig_install), and keeps
its own binary128 derivation next to them (tests/test_zig_tables.py compares all three).  Without the fixture this
script changes nothing and says so.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_ziggurat as G                                   # header writer (emit / fmt_*)


def main(argv):
    fixture = argv[1] if len(argv) > 1 else os.path.join(ROOT, "tests", "golden", "reference_pigeons.json")
    if not os.path.exists(fixture):
        print("no fixture at %s -- run tools/gen_golden.jl with Julia + Pigeons.jl first; nothing changed" % fixture)
        return 2
    ref = json.load(open(fixture))
    t = ref["tables"]
    tabs = {}
    for name in ("ki", "wi", "fi", "ke", "we", "fe"):
        a = np.array([int(v) for v in t[name]], dtype=np.uint64)
        if a.shape != (256,):
            raise SystemExit("fixture table %s has %d entries" % (name, a.size))
        tabs[name] = a
    old = G.parse_header(os.path.join(ROOT, "pigeons.jl_amd", "csrc", "zig_tables.h"))
    for name, a in tabs.items():
        diff = np.nonzero(a != old[name])[0]
        print("%s: %d of 256 entries differ from the re-derived table%s" % (name, diff.size, "" if not diff.size else
              " (first: index %d, julia %#x, derived %#x)" % (diff[0], int(a[diff[0]]), int(old[name][diff[0]]))))
    ints = lambda a: [int(v) for v in a]
    flts = lambda a: [float(v) for v in a.view(np.float64)]
    G.write_header(ints(tabs["ki"]), flts(tabs["wi"]), flts(tabs["fi"]), ints(tabs["ke"]), flts(tabs["we"]), flts(tabs["fe"]),
                   origin="Julia %s Random.ki/wi/fi/ke/we/fe (tests/golden/reference_pigeons.json, tools/import_tables.py)" % ref.get("julia_version", "?"))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_hip(force=True)
    print("pigeons.jl_amd/csrc/zig_tables.h rewritten from the fixture; libpte rebuilt")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
