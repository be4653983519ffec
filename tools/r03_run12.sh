R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run12; mkdir -p $O
cd $R
./tools/ubench/normals_prof.bin | tee $O/normals_prof.txt
timeout 900 python -m pytest tests/test_gpu_normals.py -x -q 2>&1 | tail -4
python tools/bench_toy_n.py 2>&1 | grep "N=" | tee $O/toy_n.txt
for v in n4o5 n2o5; do echo $v; PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_toy_n.py 2>&1 | grep "N=" | grep -E "N=  1024|N=  8192|N= 32768"; done | tee $O/toy_ab.txt
PTE_LIB=$R/build_variants/libpte_n4o5.so timeout 900 python -m pytest tests/test_gpu_normals.py -x -q 2>&1 | tail -2
