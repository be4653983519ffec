# usage (GPU box): bash tools/prof_configs.sh <tag>   -- kernel-trace stats of every BASELINE config shape (tools/bench_configs.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}
O=$R/gpurun_out/prof_configs_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o stats -- python3 $R/tools/bench_configs.py > $O/stats.log 2>&1
python3 $R/tools/rocpd_summary.py $O/stats_results.db > $O/summary.txt
cat $O/stats.log | grep "ms/scan"
head -20 $O/summary.txt | cut -c1-150
