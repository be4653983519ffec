# usage (GPU box): bash tools/r06_mw_ab.sh name1 name2 ...   -- ms / scan of k_explore_langevin_mw in the development builds build_variants/libpte_mw_<name>.so, interleaved twice
for rep in 1 2; do for v in "$@"; do echo "== $v"; PTE_LIB=build_variants/libpte_mw_$v.so BM_ONLY=mw python tools/bench_mw.py 2>&1 | grep "ms/scan"; done; done
