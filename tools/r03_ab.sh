R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for v in x_head x_old x_chain; do PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_variant.py 2>&1 | tail -1; done; done
