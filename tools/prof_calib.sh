# usage (GPU box): bash tools/prof_calib.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_calib; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch -- python3 $R/tools/calib_traffic.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write -- python3 $R/tools/calib_traffic.py > $O/write.log 2>&1
python3 $R/tools/rocpd_summary.py $O/fetch_results.db $O/write_results.db > $O/summary.txt
grep -E "sqr_norm|explore_toy|^kernel" $O/summary.txt | cut -c1-150
tail -1 $O/fetch.log
