#!/bin/bash
# the verified same-XCD release (PTE_HS_NEAR=1, shipped) against the full agent-scope release for every pair (=0): correctness first, then ms / scan
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_scan_loop.py tests/test_gpu_reference_reduction.py -q -x 2>&1 | tail -3
bash tools/r05_hs_ab.sh near1 near0 2>&1 | tee gpurun_out/r05_near_ab.txt
