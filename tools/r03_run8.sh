R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run8; mkdir -p $O
cd $R
for v in t1 t3 t4; do PTE_BENCH_TOY_ONLY_FIRST=1 PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy.py 2>&1 | tail -1; done | tee $O/toy_ab.txt
python tools/bench_toy.py 2>&1 | tail -3 | tee -a $O/toy_ab.txt
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_benchmarked_shapes.py tests/test_gpu_parity.py -x -q -k "normals or toy or create_replicas or quickstart or rng or init or divisor" 2>&1 | tail -3
