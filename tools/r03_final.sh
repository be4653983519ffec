# round 3 wrap-up on one GPU: full GPU test suite, the bench line (default flags), rocprofv3 passes of the bench command and of the
# HBM-bound kernels, every config, chains-per-GPU table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('$O/bench_line.json')); print({k:j[k] for k in ('value','ms_per_step','ms_per_step_without_hip_events','round_trip_rate')}); print(j['hbm_kernels']['k_explore_toy'], j['hbm_kernels']['k_init']); print([(c['config'][:12], round(c['ms_per_scan'],3)) for c in j['extra_configs']])"
python bench.py --scaling strong --gpus 1 --no-extra --no-cpu-baseline --round-trip-rounds 0 --steps 8 --warmup 2 > $O/bench_strong_1gpu.json 2>> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench_strong_1gpu.json')); print('strong anchor', j['value'], j['ms_per_step'], j['roofline']['kernel'])"
bash tools/prof_round.sh r03b > $O/prof_round.txt 2>&1; grep -E "slice8.*(calls|SQ_WAVE_CYCLES|SQ_INSTS)|avg_us" $O/prof_round.txt | head; grep -E "k_explore_slice8<4, 9>" $O/prof_round.txt | head -3 | cut -c1-130
bash tools/prof_toy.sh r03_toy_final2 > $O/prof_toy.txt 2>&1; grep -E "k_explore_toy<6>|k_init<6>" $O/prof_toy.txt | grep -v SQ_ | head -4 | cut -c1-130
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.txt
python tools/bench_nchains.py 2>&1 | grep -v amdgpu.ids | tee $O/nchains.txt
