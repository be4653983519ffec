R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run14; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -6 $O/pytest_gpu.log
bash tools/prof_toy.sh r03_toy_final > $O/prof_toy.txt 2>&1; tail -30 $O/prof_toy.txt | cut -c1-170
