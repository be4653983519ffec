# after a change to the slice kernel: speed, parity against the oracle, bit-identity against the sequential kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_slice_check; mkdir -p $O
cd $R
for rep in 1 2; do python tools/bench_variant.py 2>&1 | tail -1; done | tee $O/bv.txt
BV_N=8192 python tools/bench_variant.py 2>&1 | tail -1 | tee -a $O/bv.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmarked_shapes.py tests/test_gpu_configs.py -x -q -k "slice or metric or config1 or config2 or config4 or many_replica" > $O/pytest_slice.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_slice.log
STRESS_NSEEDS=${STRESS_NSEEDS:-3} python tools/stress_slice.py 2>&1 | tail -3 | tee $O/stress.txt
