# round 3, GPU call 3: per-wave profile of slice8, instruction rates, remaining tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run3; mkdir -p $O
cd $R
python tools/prof_waves.py > $O/prof_waves.txt 2>&1; cat $O/prof_waves.txt
./tools/ubench/rate.bin > $O/rate.txt 2>&1; cat $O/rate.txt
timeout 1200 python -m pytest tests/test_gpu_rccl_peer.py -x -q --durations=5 > $O/pytest_peer.log 2>&1; echo "pytest peer rc=$?" ; tail -12 $O/pytest_peer.log
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q --durations=5 -k "config3 or toy" > $O/pytest_shapes.log 2>&1; echo "pytest shapes rc=$?" ; tail -12 $O/pytest_shapes.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/grbm -o g -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --round-trip-rounds 0 > $O/grbm.log 2>&1
python3 $R/tools/rocpd_summary.py $O/grbm/*_results.db 2>&1 | grep -E "slice8|^kernel" | cut -c1-170
