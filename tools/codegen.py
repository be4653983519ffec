"""The generated code of libpte's hot loops as data: what tests/test_codegen_frozen.py asserts on and what tools/round_loop_lanes.py,
tools/spills_by_loop.py and tools/kernel_resources.py print.

The gains of rounds 3-4 live in places no correctness test sees -- physical registers pinned in asm constraints, an occupancy hint chosen for
where the scheduler then settles, -amdgpu-sched-strategy=max-ilp, -align-all-nofallthru-blocks=6, -O2 over -O3 -- so a compiler point release (or
an innocent edit of a header) that puts a spill or a scalar branch back into a round loop costs 5-10 % silently.  This module compiles the product's
two translation units to gfx950 assembly WITH THE SHIPPED FLAGS (__graft_entry__.FLAGS / UNITS; hipcc cross-compiles without a GPU), caches the
result under build/codegen/ keyed by a hash of sources + flags + compiler version, and answers:

  resources()            per kernel: VGPRs, SGPR / VGPR spills, scratch bytes per lane, LDS, waves per SIMD (hipcc's kernel-resource-usage remarks)
  kernel_body(sub)       the assembly of the kernel whose mangled name contains `sub`
  loops(body)            per loop (LLVM's "Loop Header: Depth=N" block comments): instruction counts, spill writes / reloads, scratch accesses
  hot_path(body, head)   the blocks from a loop's header to its back edge in layout order (hipcc lays the likely path out contiguously;
                         everything behind the back edge is out of line: refills, lane 0's continuations, the exact sequential procedure)
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "build", "codegen")


def _graft():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g


def _key(g, src, unit_flags, extra):
    h = hashlib.sha256()
    for f in sorted(os.listdir(g.CSRC)):
        h.update(f.encode()); h.update(open(os.path.join(g.CSRC, f), "rb").read())
    for f in ("pte.h", "pte_rng_policy.h"):
        h.update(open(os.path.join(ROOT, "include", f), "rb").read())
    h.update(repr((src, g.FLAGS, unit_flags, list(extra))).encode())
    try:
        h.update(subprocess.run([g.HIPCC, "--version"], capture_output=True, text=True).stdout.encode())
    except OSError:
        pass
    return h.hexdigest()[:16]


def compile_units(extra=(), force=False):
    """-> [(source, command, path of the .s, text of the resource remarks)] for the product's translation units, compiled in parallel, cached"""
    g = _graft()
    os.makedirs(CACHE, exist_ok=True)
    jobs = []
    for src, unit_flags in g.UNITS:
        key = _key(g, src, unit_flags, extra)
        stem = os.path.join(CACHE, "%s.%s" % (os.path.splitext(src)[0], key))
        cmd = [g.HIPCC, *[f for f in g.FLAGS if f != "-fPIC"], *unit_flags, *extra, "--cuda-device-only", "-S",
               "-Rpass-analysis=kernel-resource-usage", "-o", stem + ".s", os.path.join(g.CSRC, src)]
        jobs.append((src, cmd, stem))

    def run(job):
        src, cmd, stem = job
        if force or not (os.path.exists(stem + ".s") and os.path.exists(stem + ".remarks")):
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
            open(stem + ".remarks", "w").write(r.stderr)
        return src, cmd, stem + ".s", open(stem + ".remarks").read()

    with ThreadPoolExecutor(len(jobs)) as ex:
        out = list(ex.map(run, jobs))
    # prune: an assembly file is 15-40 MB (and build/ travels with every gpurun snapshot); keep what this call produced, nothing else
    keep = set(os.path.basename(stem) for _, _, stem in jobs)
    for f in os.listdir(CACHE):
        full = os.path.join(CACHE, f)
        if not extra and f.rsplit(".", 1)[0] not in keep:
            try:
                os.remove(full)
            except OSError:
                pass
    return out


_RES_KEYS = [("vgpr", r" VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"), ("spilled_vgpr", r"VGPRs Spill"), ("spilled_sgpr", r"SGPRs Spill"),
             ("scratch_B_per_lane", r"ScratchSize \[bytes/lane\]"), ("waves_per_simd", r"Occupancy \[waves/SIMD\]"), ("lds_B", r"LDS Size \[bytes/block\]")]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.split("\n")[:len(names)] if r.returncode == 0 else list(names)
    return [re.sub(r"\(.*\)$", "", o).replace("void pte::", "").replace("pte::", "") for o in out]


def resources(units=None):
    """{demangled kernel name without arguments, e.g. 'k_explore_slice8<4, 9>': {vgpr, sgpr, spilled_vgpr, spilled_sgpr, scratch_B_per_lane, ...}}"""
    units = units or compile_units()
    res, order = {}, []
    for _, _, _, remarks in units:
        blocks = re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]
        names = demangle([b.split("\n")[0].split(" [")[0].strip() for b in blocks])
        for name, b in zip(names, blocks):
            vals = {}
            for k, pat in _RES_KEYS:
                m = re.search(pat + r": (\S+)", b)
                vals[k] = int(m.group(1)) if m and m.group(1).isdigit() else None
            res[name] = vals
            order.append(name)
    res["__order__"] = order
    return res


def asm_lines(units=None):
    units = units or compile_units()
    lines = []
    for _, _, path, _ in units:
        lines += open(path).read().split("\n")
    return lines


def kernel_body(lines, sub):
    """(mangled name, lines of the kernel) for the first kernel whose mangled name contains `sub`"""
    a = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3pte\w*:", l) and sub in l)
    e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[a].split(":")[0], lines[a:e]


_LABEL = re.compile(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):")


def blocks_of(body):
    """basic blocks in layout order: name, innermost loop header (BBn_m) or None, depth, is_header, instruction counts, spill / scratch traffic,
    the lane instructions, and the labels its branches go to"""
    spill_regs = set(m.group(1) for l in body for m in [re.match(r"^\s*v_writelane_b32 (v\d+), s\d+, \d+", l)] if m)
    blocks, cur = [], None

    def new(name, head, depth, is_header):
        return {"name": name, "head": head, "depth": depth, "is_header": is_header, "v": 0, "s": 0, "l": 0, "m": 0, "scratch": 0, "w": 0, "r": 0,
                "dyn": 0, "nop": 0, "branches": [], "lane": [], "text": []}
    for i, l in enumerate(body):
        m = _LABEL.match(l)
        if m:
            nm = m.group(1) or ("bb." + m.group(2))
            com = []
            for j in range(i, min(i + 12, len(body))):
                if j > i and not body[j].lstrip().startswith(";") and "Loop" not in body[j]:
                    break
                com.append(body[j])
            com = " ".join(com)
            mh = re.search(r"Loop Header: Depth=(\d+)", com)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", com)
            if mh:
                cur = new(nm, nm[2:] if nm.startswith(".L") else nm, int(mh.group(1)), True)
            elif mi:
                cur = new(nm, mi.group(1), int(mi.group(2)), False)
            else:
                cur = new(nm, None, 0, False)
            cur["parents"] = re.findall(r"Parent Loop (BB\d+_\d+)", com)
            blocks.append(cur)
            continue
        if cur is None:
            cur = new("entry", None, 0, False); cur["parents"] = []
            blocks.append(cur)
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        cur["text"].append(t)
        mr = re.match(r"^v_readlane_b32 s\d+, (v\d+), (\S+)", t)
        if re.match(r"^v_writelane_b32 v\d+, s\d+, \d+", t):
            cur["w"] += 1
        elif mr and mr.group(2).isdigit() and mr.group(1) in spill_regs:
            cur["r"] += 1
        elif mr:
            cur["dyn"] += 1
        if re.match(r"^v_(readlane|writelane|readfirstlane)", t):
            cur["lane"].append(t)
        if t.startswith("v_"):
            cur["v"] += 1
        elif t.startswith("s_"):
            cur["s"] += 1
            if t.startswith("s_nop"):
                cur["nop"] += 1
            mb = re.match(r"^s_c?branch\w* (\.LBB\d+_\d+)", t)
            if mb:
                cur["branches"].append(mb.group(1))
        elif t.startswith("ds_"):
            cur["l"] += 1
        elif re.match(r"^(global|scratch|buffer|flat)_", t):
            cur["m"] += 1
            if t.startswith("scratch_"):
                cur["scratch"] += 1
    return blocks


_SUM = ("v", "s", "l", "m", "scratch", "w", "r", "dyn", "nop")


def loops(body):
    """{(depth, header): counts over the blocks whose INNERMOST loop it is}; (0, None) is the code outside every loop"""
    out = {}
    for b in blocks_of(body):
        L = out.setdefault((b["depth"], b["head"]), dict({k: 0 for k in _SUM}, n=0))
        L["n"] += 1
        for k in _SUM:
            L[k] += b[k]
    return out


def hot_path(body, header):
    """the blocks of the loop `header` ('BB29_50') from its header label to the block that branches back to it, in layout order -- the
    fall-through path hipcc laid out for the likely case; nested loops met on the way are included (their blocks are marked by depth)"""
    bl = blocks_of(body)
    label = ".L" + header
    a = next(i for i, b in enumerate(bl) if b["name"] == label)
    path = []
    for b in bl[a:]:
        if b["name"] != label and b["head"] != header and header not in b.get("parents", []):
            break                                   # left the loop without meeting a back edge
        path.append(b)
        if label in b["branches"]:
            return path
    return path


def totals(path):
    t = {k: sum(b[k] for b in path) for k in _SUM}
    t["instructions"] = t["v"] + t["s"] + t["l"] + t["m"]
    t["blocks"] = len(path)
    return t


def loop_headers(body, depth=None):
    return [(b["depth"], b["head"]) for b in blocks_of(body) if b["is_header"] and (depth is None or b["depth"] == depth)]
