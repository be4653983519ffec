# A/B of PTE_LIB builds on ONE box over tools/bench_configs.py lines matching $1 (a grep pattern): default and each variant, three times, interleaved
R=$GRAFT_REPO_ROOT; cd $R; PAT=$1; shift
for rep in 1 2 3; do
  for v in default "$@"; do
    if [ "$v" = default ]; then L=$R/pigeons.jl_amd/lib/libpte.so; else L=$R/build_variants/libpte_v_$v.so; fi
    printf "%-10s rep %d: " $v $rep; PTE_LIB=$L BC_LANGEVIN_LARGE=0 python tools/bench_configs.py 2>&1 | grep -E "$PAT" | awk '{printf "%s ms   ", $(NF-3)}'; echo
  done
done
