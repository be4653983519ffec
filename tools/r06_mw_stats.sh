# usage (GPU box): bash tools/r06_mw_stats.sh  -- rocprofv3 --kernel-trace --stats of tools/bench_mw.py (the final library): per-kernel launch durations of the
# four-wave Langevin kernels, per-scan launches (mw2) and the one-launch scan loop (mw: 2, 4, 8, 16, 2, BM_SCANS scans per call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mw_stats; mkdir -p $O
export BM_ONLY=mw,mw2 BM_SCANS=64
rocprofv3 --kernel-trace --stats -d $O -o t -- python3 $R/tools/bench_mw.py > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
python3 - "$O" <<'PY'
import sqlite3, sys, os, glob
db = glob.glob(os.path.join(sys.argv[1], "**", "t_results.db"), recursive=True)[0]
con = sqlite3.connect(db)
print("kernel | launches | avg us | min us | max us | grid | workgroup | LDS B | scratch B/lane | VGPRs")
for r in con.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(grid_x), max(workgroup_x), max(lds_size), max(scratch_size), max(vgpr_count) from kernels where name like '%langevin_mw%' or name like '%k_swap%' group by name order by name").fetchall():
    print(" | ".join(str(round(x, 1)) if isinstance(x, float) else str(x) for x in r))
# the last 64-scan call of every configuration: the scan loop's launch / 64 against the mean of the last 64 per-scan launches
for name in [r[0] for r in con.execute("select distinct name from kernels where name like '%k_scans_langevin_mw%'")]:
    rows = con.execute("select (end-start)/1e3 from kernels where name = ? order by start", (name,)).fetchall()
    print(name[:70], "launches (us):", [round(x[0]) for x in rows], "-> last call per scan: %.1f us" % (rows[-1][0] / 64.0))
for name in [r[0] for r in con.execute("select distinct name from kernels where name like '%k_explore_langevin_mw%'")]:
    rows = [x[0] for x in con.execute("select (end-start)/1e3 from kernels where name = ? order by start", (name,)).fetchall()]
    print(name[:70], "%d launches; mean of the last 64: %.1f us, min %.1f, max %.1f" % (len(rows), sum(rows[-64:]) / 64.0, min(rows[-64:]), max(rows[-64:])))
PY
find $O -name "*.db" -delete
