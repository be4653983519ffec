#!/bin/bash
# What does a wave of the slice kernel lose when the other three SIMDs of its compute unit are busy too?  256 chains (one wave per CU) against
# 1024 (four): per-wave counters of the fused scan loop (64 scans per launch), instruction fetch and LDS side.  Counters in their own runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_cu_sharing; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|LDS|INST_CYCLES|WAIT_INST|BUSY_CY|INSTS_SMEM|DCACHE)[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ' > $O/counters.txt; echo >> $O/counters.txt
for N in 256 1024; do
  A="--no-cpu-baseline --no-extra --round-trip-rounds 0 --steps 64 --warmup 64 --prepare 0 --chains $N"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o lds$N -- python3 $R/bench.py $A > $O/lds$N.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O -o if$N -- python3 $R/bench.py $A > $O/if$N.log 2>&1
done
python3 - "$O" <<'PY'
import sqlite3, sys, os
O = sys.argv[1]
print(open(os.path.join(O, "counters.txt")).read()[:1500])
for N in (256, 1024):
    for tag in ("lds", "if"):
        p = os.path.join(O, "%s%d_results.db" % (tag, N))
        if not os.path.exists(p): print("missing", p, open(os.path.join(O, "%s%d.log" % (tag, N))).read()[-600:]); continue
        con = sqlite3.connect(p)
        rows = con.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_scans_slice8%' group by counter_name").fetchall()
        d = {r[0]: r[1] for r in rows}
        w = d.get("SQ_WAVES", 0) or 1
        print("N=%d %s: per wave (64 scans): %s" % (N, tag, "  ".join("%s %.0f" % (k, v / w) for k, v in sorted(d.items()) if k != "SQ_WAVES")), " [waves %.0f, launches %d]" % (w, rows[0][2] if rows else 0))
PY
