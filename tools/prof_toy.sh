# usage: bash tools/prof_toy.sh <tag>   -- rocprofv3 kernel stats + SQ / TCC counters of k_explore_toy / k_init / k_swap at N = 8192, d = 4096
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-toy}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
export PTE_BENCH_TOY_ONLY_FIRST=1
rocprofv3 --kernel-trace --stats -d $O -o kt -- python3 $R/tools/bench_toy.py > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O -o sq -- python3 $R/tools/bench_toy.py > $O/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O -o sq2 -- python3 $R/tools/bench_toy.py > $O/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch -- python3 $R/tools/bench_toy.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write -- python3 $R/tools/bench_toy.py > $O/write.log 2>&1
python3 $R/tools/rocpd_summary.py $O/kt_results.db $O/sq_results.db $O/sq2_results.db $O/fetch_results.db $O/write_results.db > $O/summary.txt 2>&1
grep -E "k_explore_toy|k_init|k_swap|^kernel|^==" $O/summary.txt | cut -c1-160
