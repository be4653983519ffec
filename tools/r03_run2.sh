# round 3, GPU call 2: placement ubench, the remaining benchmarked-shape tests, the production transport with a peer
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run2; mkdir -p $O
cd $R
./tools/ubench/placement.bin > $O/placement.txt 2>&1; cat $O/placement.txt
timeout 1200 python -m pytest tests/test_gpu_rccl_peer.py -x -q --durations=10 > $O/pytest_peer.log 2>&1; echo "pytest peer rc=$?" ; tail -25 $O/pytest_peer.log
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q --durations=10 > $O/pytest_shapes.log 2>&1; echo "pytest shapes rc=$?" ; tail -12 $O/pytest_shapes.log
