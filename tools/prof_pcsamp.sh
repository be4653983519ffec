cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcsamp; mkdir -p $O
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d $O -o pcs -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $O/log.txt 2>&1
echo rc=$?; tail -5 $O/log.txt | cut -c1-200; ls -la $O | head -20
