# round 3 wrap-up measurements on one GPU: full GPU test suite, every config, chains-per-GPU table, toy N sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/pytest_gpu.log
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.txt
python tools/bench_nchains.py 2>&1 | grep -v amdgpu.ids | tee $O/nchains.txt
python tools/bench_toy_n.py 2>&1 | grep "N=" | tee $O/toy_n.txt
STRESS_NSEEDS=6 python tools/stress_slice.py 2>&1 | tail -2 | tee $O/stress_slice.txt
