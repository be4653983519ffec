# fused scan loop against the launch-per-scan loop over shapes (the product library), two repetitions each
R=$GRAFT_REPO_ROOT; cd $R
for shape in "1024 2048" "2048 1024" "512 4096" "2500 1024" "1024 4096" "2048 256" "1024 512" "256 4096"; do
  set -- $shape
  for rep in 1 2; do
    BV_N=$1 BV_D=$2 python tools/bench_variant.py 2>&1 | grep ms/scan
    BV_N=$1 BV_D=$2 BV_TWO=1 python tools/bench_variant.py 2>&1 | grep ms/scan
  done
done
