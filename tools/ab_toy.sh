# A/B of normal-generator build variants ON ONE BOX (boxes differ by +-3 %): every variant measured three times, interleaved
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for v in default "$@"; do
    if [ "$v" = default ]; then L=$R/pigeons.jl_amd/lib/libpte.so; else L=$R/build_variants/libpte_v_$v.so; fi
    printf "%-10s rep %d: " $v $rep; PTE_LIB=$L python tools/bench_toy_n.py 2>&1 | grep -E "N=  8192|N= 32768" | awk '{printf "%s %s GB/s   ", $1 $2, $8}'; echo
  done
done
