# usage: bash tools/prof_pmc.sh <tag> [bench args...]   -- collects SQ counters for one short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/pmc_$TAG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_$TAG -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/pmc_$TAG -o q -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra "$@" >> $R/gpurun_out/pmc_$TAG.log 2>&1
python3 $R/tools/rocpd_summary.py $R/gpurun_out/pmc_$TAG/p_results.db $R/gpurun_out/pmc_$TAG/q_results.db | grep -E "slice|^kernel|^==" 
