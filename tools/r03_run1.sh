# round 3, GPU call 1: oracle checks of the benchmarked shapes, twin-kernel A/B, issue floor + LDS latency ubench, PC sampling probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q --durations=10 > $O/pytest_shapes.log 2>&1; echo "pytest rc=$?" ; tail -15 $O/pytest_shapes.log
./tools/ubench/issue_floor.bin > $O/issue_floor.txt 2>&1; cat $O/issue_floor.txt
./tools/ubench/lds_lat.bin > $O/lds_lat.txt 2>&1; cat $O/lds_lat.txt
for N in 3072 4096 8192; do for v in base twin3 notwin; do BV_N=$N PTE_LIB=$R/build_variants/libpte_$v.so python tools/bench_variant.py 2>&1 | tail -1; done; done | tee $O/twin_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/floor_sq -o sq -- $R/tools/ubench/issue_floor.bin > $O/floor_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/floor_grbm -o g -- $R/tools/ubench/issue_floor.bin > $O/floor_grbm.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/floor_kt -o kt -- $R/tools/ubench/issue_floor.bin > $O/floor_kt.log 2>&1
python3 $R/tools/rocpd_summary.py $O/floor_sq/*_results.db $O/floor_grbm/*_results.db $O/floor_kt/*_results.db > $O/floor_pmc_summary.txt 2>&1; cat $O/floor_pmc_summary.txt | cut -c1-170
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d $O/pcs -o pcs -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/pcs.log 2>&1; echo "pcs rc=$?"; tail -3 $O/pcs.log | cut -c1-200; ls $O/pcs | head
