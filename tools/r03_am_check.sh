# after a change to the Langevin kernels: C3 speed, parity against the oracle, the stress sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_am_check; mkdir -p $O
cd $R
for rep in 1 2; do python tools/bench_configs.py 2>&1 | grep -E "C3|funnel\(32\)"; done | tee $O/c3.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmarked_shapes.py -x -q -k "automala or mala or compose or funnel or variational or gaussian or config3 or two_leg" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/stress_langevin.py 2>&1 | tail -2
