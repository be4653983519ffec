R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run9; mkdir -p $O
cd $R
for v in w2o5 w2o5u3 w4o5 w4o6u3 w4o6; do PTE_BENCH_TOY_ONLY_FIRST=1 PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy.py 2>&1 | tail -1; done | tee $O/toy_ab.txt
PTE_LIB=$R/build_variants/libpte_w4o5.so timeout 600 python -m pytest tests/test_gpu_normals.py -x -q 2>&1 | tail -2
