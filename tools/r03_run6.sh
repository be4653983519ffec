R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run6; mkdir -p $O
cd $R
for v in toy_occ4 toy_occ5 toy_occ6; do PTE_BENCH_TOY_ONLY_FIRST=1 PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy.py 2>&1 | tail -1; done | tee $O/toy_ab.txt
PTE_LIB=$R/build_variants/libpte_toy_occ4.so timeout 600 python -m pytest tests/test_gpu_normals.py -x -q 2>&1 | tail -3
PTE_LIB=$R/build_variants/libpte_toy_occ5.so timeout 600 python -m pytest tests/test_gpu_normals.py -x -q 2>&1 | tail -3
