#!/bin/bash
# the one-kernel scan loop with four consecutive chains per workgroup (LDS hand-shakes for three of four pairs; default where it exists) against
# the form with one chain per workgroup (BV_ONE=1) and the launch-per-scan loop (BV_TWO=1): correctness first, then ms / scan by shape
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_scan_loop.py tests/test_gpu_reference_reduction.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -q -x -k "slice" 2>&1 | tail -2
for rep in 1 2 3; do
  for shape in "1024 1024" "256 1024" "10 2" "1024 512" "1024 2048"; do
    set -- $shape
    for form in "" BV_ONE=1 BV_TWO=1; do
      env BV_N=$1 BV_D=$2 $form python tools/bench_variant.py 2>&1 | grep ms/scan
    done
  done
done | tee gpurun_out/r05_wg_ab.txt
