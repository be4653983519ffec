cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mw_trace; mkdir -p $O
BM_ONLY=mw rocprofv3 --kernel-trace -d $O -o t -- python3 $R/tools/bench_mw.py > $O/log.txt 2>&1
python3 - "$O" <<'PY'
import sqlite3, sys, os
con = sqlite3.connect(os.path.join(sys.argv[1], "t_results.db"))
for r in con.execute("select name, count(*), avg(end-start)/1e3, max(grid_x), max(workgroup_x), max(lds_size), max(static_lds_size), max(scratch_size), max(static_scratch_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) from kernels where name like '%langevin_mw%' group by name").fetchall(): print(r)
PY
find $O -name "*.db" -delete
