cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_stats $R/gpurun_out/prof_pmc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/prof_pmc -o p1 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_pmc.log 2>&1
ls -R $R/gpurun_out/prof_stats $R/gpurun_out/prof_pmc | head -40
