# round 4 wrap-up on one GPU: the bench line (default flags), rocprofv3 passes of the bench command and of the HBM-bound kernels,
# every config, chains-per-GPU table, the lone-wave microbenchmarks of round 4.  (The GPU test suite is run separately.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('$O/bench_line.json')); print({k:j[k] for k in ('value','ms_per_step','ms_per_step_without_hip_events','round_trip_rate','vs_baseline')}); print(j['long_run']); print(j['hbm_kernels']['k_explore_toy'], j['hbm_kernels']['k_init']); print([(c['config'][:12], round(c['ms_per_scan'],3)) for c in j['extra_configs']])"
python bench.py --scaling strong --gpus 1 --no-extra --no-cpu-baseline --round-trip-rounds 0 --steps 8 --warmup 2 > $O/bench_strong_1gpu.json 2>> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench_strong_1gpu.json')); print('strong anchor', j['value'], j['ms_per_step'], j['roofline']['kernel'])"
bash tools/prof_round.sh r04 > $O/prof_round.txt 2>&1; grep -E "k_explore_slice8<4, 9>" $O/prof_round.txt | head -12 | cut -c1-150
bash tools/prof_toy.sh r04_toy > $O/prof_toy.txt 2>&1; grep -E "k_explore_toy<6>|k_init<6>" $O/prof_toy.txt | head -30 | cut -c1-150
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.txt
python tools/bench_nchains.py 2>&1 | grep -v amdgpu.ids | tee $O/nchains.txt
python tools/bench_toy_n.py 2>&1 | grep -v amdgpu.ids | tee $O/toy_nchains.txt
./tools/ubench/round_cost.bin > $O/round_cost.txt 2>&1; ./tools/ubench/hop_check.bin > $O/hop_check.txt 2>&1
for p in 1 2 3 5; do timeout 60 ./tools/ubench/rate2.bin $p; done > $O/rate2.txt 2>&1
for b in normals_dev_r03 normals_dev_plain normals_dev; do for i in 1 2 3; do printf "%-18s " $b; ./tools/ubench/$b.bin; done; printf "%-18s " $b; ./tools/ubench/$b.bin 32768; done > $O/normals_dev.txt 2>&1
./tools/ubench/wave_sum_pairs_r03.bin > $O/wave_sum_pairs.txt 2>&1; ./tools/ubench/wave_sum_pairs.bin >> $O/wave_sum_pairs.txt 2>&1
