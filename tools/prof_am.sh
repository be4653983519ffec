cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_am; mkdir -p $O
python3 $R/tools/am_stats.py 2>&1 | tail -7
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS SQ_WAIT_ANY SQ_INSTS_LDS -d $O -o sq -- python3 $R/tools/am_stats.py > $O/sq.log 2>&1
python3 $R/tools/rocpd_summary.py $O/sq_results.db | grep -i "automala" | cut -c1-150
