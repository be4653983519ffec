#!/bin/bash
# AutoMALA: the scan loop with four chains per workgroup against one chain per workgroup and the launch-per-scan loop, by chain count
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_scan_loop.py -q -x 2>&1 | tail -2
for n in 1024 512 256 64 10; do echo "# funnel d = 128, $n chains"; BV_N=$n python tools/bench_am_forms.py 2>&1 | grep ms/scan | sort | awk '{a[$2]=a[$2]" "$(NF-1)} END {for (k in a) print k, a[k]}'; done | tee gpurun_out/r05_am_wg_ab.txt
for n in 1024 256; do echo "# MVN d = 512, $n chains"; BV_MVN=512 BV_N=$n python tools/bench_am_forms.py 2>&1 | grep ms/scan | sort | awk '{a[$2]=a[$2]" "$(NF-1)} END {for (k in a) print k, a[k]}'; done | tee -a gpurun_out/r05_am_wg_ab.txt
