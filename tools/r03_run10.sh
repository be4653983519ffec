R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run10; mkdir -p $O
cd $R
python tools/bench_toy_n.py 2>&1 | grep "N=" | tee $O/toy_n.txt
PTE_LIB=$R/build_variants/libpte_w4o6.so python tools/bench_toy_n.py 2>&1 | grep "N=" | tee $O/toy_n_w4o6.txt
