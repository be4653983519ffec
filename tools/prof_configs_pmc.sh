# usage (GPU box): bash tools/prof_configs_pmc.sh <tag>  -- kernel-trace stats AND SQ instruction counters of every BASELINE config shape
# (tools/bench_configs.py): k_explore_automala (C3), k_explore_ising_spec (C5), k_explore_toy, k_explore_slice8 at d = 4096 (C4 shard)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
O=$R/gpurun_out/prof_configs_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o stats -- python3 $R/tools/bench_configs.py > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- python3 $R/tools/bench_configs.py > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O -o sq2 -- python3 $R/tools/bench_configs.py > $O/sq2.log 2>&1
python3 $R/tools/rocpd_summary.py $O/stats_results.db $O/sq_results.db $O/sq2_results.db > $O/summary.txt
grep "ms/scan" $O/stats.log
grep -E "^==|^kernel|k_explore|k_swap" $O/summary.txt | grep -v "k_explore_slice8<4, 9>.*SQ_\|fillBuffer" | cut -c1-150 | head -90
