cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; O=$R/gpurun_out/prof_${TAG}_steps20; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --round-trip-rounds 0 > $O/stats.log 2>&1
python3 - "$O" <<'PY'
import sqlite3, sys, json, os
O = sys.argv[1]
con = sqlite3.connect(os.path.join(O, "stats_results.db"))
rows = con.execute("select name, (end-start)/1e6 from kernels where name like '%k_scans%' order by start").fetchall()
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --round-trip-rounds 0   (the driver's step counts)")
print("# dispatches of the fused scan loop, in order: preparation (64 scans, then 64 after the adaptation), warm-up (5 scans), TIMED region (20 scans, HIP events riding on the launch), the same 20 scans without events, long_run (256 scans)")
for n, ms in rows: print("%-90s %10.3f ms" % (n[:90], ms))
j = json.loads([l for l in open(os.path.join(O, "stats.log")) if l.startswith("{")][-1]); r = j["roofline"]
print("# bench line of the same process: roofline.kernel %s, avg_launch_ms %.3f over %d launch(es) of %d scans = %.4f ms per scan; ms_per_step %.4f; long_run %.4f ms per step" % (r["kernel"], r["avg_launch_ms"], r["launches"], r["scans_per_launch"], r["avg_launch_ms_per_scan"], j["ms_per_step"], j["long_run"]["ms_per_step"]))
for r_ in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:5]: print("%-90s calls %4d  total %10.0f us  avg %10.0f us  %6.2f %%" % (r_[0][:90], r_[1], r_[2], r_[3], r_[4]))
PY
