# usage (on the GPU box): bash tools/prof_configs6.sh <tag>
# Round 6: rocprofv3 --kernel-trace --stats + SQ / FP64 / FETCH / WRITE passes of the FINAL library for every config of
# bench.py's extra_configs (C1, C2, C3 = k_scans_automala_wg, C4 shard, C4 on one GPU, C5 shard), prepared exactly as bench.py prepares them
# (tools/prof_configs6.py).  Counters are collected in their own runs, the program directly after `--`, as the pool requires.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}
O=$R/gpurun_out/prof_configs6_$TAG; mkdir -p $O
P="python3 $R/tools/prof_configs6.py"
rocprofv3 --kernel-trace --stats -d $O -o stats -- $P > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- $P > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $O -o fp -- $P > $O/fp.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O -o grbm -- $P > $O/grbm.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch -- $P > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write -- $P > $O/write.log 2>&1
python3 $R/tools/r06_configs_summary.py $O > $O/summary.txt 2> $O/summary.err
cp $O/configs.json $R/gpurun_out/r06_configs_$TAG.json 2>/dev/null
cp $O/summary.txt $R/gpurun_out/r06_configs_pmc_summary_$TAG.txt
ls -la $O | head -30; du -sh $O
find $O -name "*.db" -delete          # (gpurun copies back at most 64 MiB: the summaries travel, the databases do not)
tail -5 $O/summary.err; cut -c1-200 $O/summary.txt | head -150
