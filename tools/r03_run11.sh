R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run11; mkdir -p $O
cd $R
for v in f1 f4o5 f8o6 g4o6; do echo $v; PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy_n.py 2>&1 | grep "N=" | grep -E "N=  1024|N=  8192|N= 32768"; done | tee $O/toy_ab.txt
