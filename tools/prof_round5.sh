# usage (on the GPU box): bash tools/prof_round5.sh <tag>
# Round 5: pte_run_scans is ONE launch (k_scans_slice8) holding all the scans of a call, so the profile runs give every launch the same
# number of scans (--steps S --warmup S: warm-up, timed and event-free passes are three launches of S scans each) and the summary is per scan.
# Counters are collected in their own runs (no --sys-trace etc.), as the pool requires.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; S=${2:-256}
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
A="--no-cpu-baseline --no-extra --round-trip-rounds 0 --steps $S --warmup $S"
rocprofv3 --kernel-trace --stats -d $O -o stats -- python3 $R/bench.py $A > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- python3 $R/bench.py $A > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch -- python3 $R/bench.py $A > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write -- python3 $R/bench.py $A > $O/write.log 2>&1
python3 $R/tools/rocpd_summary.py $O/stats_results.db $O/sq_results.db $O/fetch_results.db $O/write_results.db > $O/summary.txt
echo "scans per launch: $S" >> $O/summary.txt
grep -E "scans|slice|swap|^==|^kernel" $O/summary.txt | cut -c1-170
grep -h "^{" $O/stats.log | python3 -c "
import sys, json
for ln in sys.stdin:
    j = json.loads(ln); r = j['roofline']
    print('bench line of the --kernel-trace run: value %.0f, ms_per_step %.4f, roofline kernel %s avg_launch_ms %.3f = %.4f ms per scan over %d scans' % (j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['avg_launch_ms_per_scan'], r['scans_per_launch']))
" | tee -a $O/summary.txt
