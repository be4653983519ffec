# after a change to the Ising kernel: C5 speed, parity against the oracle and the byte-lattice kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_ising_check; mkdir -p $O
cd $R
for rep in 1 2; do python tools/bench_configs.py 2>&1 | grep -E "C5"; done | tee $O/c5.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rccl_peer.py -x -q -k "ising" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
