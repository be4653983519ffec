R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run7; mkdir -p $O
cd $R
./tools/ubench/store_pattern.bin | tee $O/store_pattern.txt
for v in ta tb tc td te tf; do PTE_BENCH_TOY_ONLY_FIRST=1 PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy.py 2>&1 | tail -1; done | tee $O/toy_ab.txt
