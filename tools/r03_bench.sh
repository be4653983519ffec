# the driver's own sequence on one GPU: bench line (default flags), then the rocprofv3 passes of the same command (tools/prof_round.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_bench; mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python - <<'P'
import json,os
j=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03_bench/bench_line.json"))
print(json.dumps({k:j[k] for k in ("value","ms_per_step","ms_per_step_without_hip_events","round_trip_rate")}))
print(json.dumps(j["roofline"])[:900])
print(json.dumps(j["hbm_kernels"]))
print(json.dumps(j["extra_configs"]))
print(json.dumps(j["cpu_baseline"])[:400])
P
bash tools/prof_round.sh r03 > $O/prof_round.txt 2>&1; tail -25 $O/prof_round.txt | cut -c1-170
