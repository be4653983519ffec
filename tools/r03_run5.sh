R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_normals.py -x -q --durations=5 > $O/pytest_normals.log 2>&1; echo "pytest normals rc=$?" ; tail -15 $O/pytest_normals.log
python tools/bench_toy.py 2>&1 | tail -4 | tee $O/bench_toy.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmarked_shapes.py -x -q -k "toy or create_replicas or quickstart or rng" > $O/pytest_toy.log 2>&1; echo "pytest toy rc=$?" ; tail -8 $O/pytest_toy.log
