# usage (GPU box): bash tools/r06_mw_pmc.sh [variant]  -- SQ counters of k_explore_langevin_mw at toy_mvn(1024) / funnel(1024), N = 1024 (tools/bench_mw.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mw_pmc; mkdir -p $O
V=$1
if [ -n "$V" ]; then export PTE_LIB=$R/build_variants/libpte_mw_$V.so; fi
export BM_ONLY=${BM_ONLY:-mw}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- python3 $R/tools/bench_mw.py > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 -d $O -o sq2 -- python3 $R/tools/bench_mw.py > $O/sq2.log 2>&1
python3 - "$O" <<'PY'
import sqlite3, sys, os
O = sys.argv[1]
for db in ("sq_results.db", "sq2_results.db"):
    con = sqlite3.connect(os.path.join(O, db))
    rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%langevin_mw%' group by kernel_name, counter_name").fetchall()
    by = {}
    for k, c, v, n in rows: by.setdefault(k, {})[c] = v
    for k, c in by.items():
        print(k[:60])
        if "SQ_WAVES" in c:
            W = c["SQ_WAVES"]; ins = (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"]) / W; cyc = c["SQ_WAVE_CYCLES"] * 4 / W
            print("   per wave and scan: VALU %.0f SALU %.0f branch %.0f = %.1f k instructions; wave cycles %.3f M; %.2f cycles / instruction; WAIT_ANY %.3f WAIT_INST_ANY %.3f ACTIVE %.3f" % (c["SQ_INSTS_VALU"] / W, c["SQ_INSTS_SALU"] / W, c["SQ_INSTS_BRANCH"] / W, ins / 1e3, cyc / 1e6, cyc / ins, c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
        else:
            print("   per dispatch: " + "  ".join("%s %.3g" % (a.replace("SQ_INSTS_", "").replace("SQ_", ""), b) for a, b in sorted(c.items())))
PY
find $O -name "*.db" -delete
