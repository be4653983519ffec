# A/B of scan-loop builds on one box: ms / scan at the metric shape, d = 4096, C2 and C1
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for v in "$@"; do
    PTE_LIB=$R/build_variants/libpte_v_$v.so python tools/bench_variant.py 2>&1 | grep ms/scan
  done
  BV_TWO=1 PTE_LIB=$R/build_variants/libpte_v_$1.so python tools/bench_variant.py 2>&1 | grep ms/scan
done
for v in "$@"; do
  BV_D=4096 PTE_LIB=$R/build_variants/libpte_v_$v.so python tools/bench_variant.py 2>&1 | grep ms/scan
  BV_N=256 PTE_LIB=$R/build_variants/libpte_v_$v.so python tools/bench_variant.py 2>&1 | grep ms/scan
done
BV_D=4096 BV_TWO=1 PTE_LIB=$R/build_variants/libpte_v_$1.so python tools/bench_variant.py 2>&1 | grep ms/scan
BV_N=256 BV_TWO=1 PTE_LIB=$R/build_variants/libpte_v_$1.so python tools/bench_variant.py 2>&1 | grep ms/scan
