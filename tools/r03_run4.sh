R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run4; mkdir -p $O
cd $R
S_IMPL=8 python tools/prof_sections7.py > $O/sections8.txt 2>&1; cat $O/sections8.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q --durations=5 -k "config3 or toy" > $O/pytest_shapes.log 2>&1; echo "pytest shapes rc=$?" ; tail -12 $O/pytest_shapes.log
