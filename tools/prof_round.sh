# usage (on the GPU box): bash tools/prof_round.sh <round-tag>
# kernel-trace stats + PMC passes (SQ instruction mix, HBM fetch / write) of one short bench.py run each.
# Counters are collected in their own runs (no --sys-trace etc.), as the pool requires.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o stats -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra --round-trip-rounds 0 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --round-trip-rounds 0 > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --round-trip-rounds 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --round-trip-rounds 0 > $O/write.log 2>&1
python3 $R/tools/rocpd_summary.py $O/stats_results.db $O/sq_results.db $O/fetch_results.db $O/write_results.db > $O/summary.txt
grep -E "slice|swap|^==|^kernel" $O/summary.txt | cut -c1-160
