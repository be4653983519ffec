# after a change to the normal generator: speed by chain count, bit-identity to the sequential procedure, oracle parity of the toy paths
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_toy_check; mkdir -p $O
cd $R
python tools/bench_toy_n.py 2>&1 | grep "N=" | tee $O/toy_n.txt
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_benchmarked_shapes.py tests/test_gpu_parity.py -x -q -k "normals or toy or create_replicas or quickstart or rng or init or divisor" 2>&1 | tail -3
./tools/ubench/normals_prof.bin 2>/dev/null | tee $O/normals_prof.txt
