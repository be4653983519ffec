#!/bin/bash
# bench.py as the driver runs it (--steps 20 --warmup 5) with the declared preparation of the input (default) and without (--prepare 0), then the
# contract tests.  Output: gpurun_out/r05_bench_prep.{json,txt}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_prep.json 2> gpurun_out/r05_bench_prep.err
python - > gpurun_out/r05_bench_prep.txt <<'PY'
import json, subprocess, sys
def line(txt): return json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
j = line(open("gpurun_out/r05_bench_prep.json").read())
print("prepared :", "%.0f" % j["value"], "replica-steps/s  ms_per_step %.4f  long_run %.4f  launch/scan %.4f" % (j["ms_per_step"], j["long_run"]["ms_per_step"], j["roofline"]["avg_launch_ms_per_scan"]))
print("          ", j["config"]["preparation"])
for rep in range(2):
    for prep in ("64", "0"):
        p = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--prepare", prep, "--no-extra", "--no-cpu-baseline", "--round-trip-rounds", "0"], capture_output=True, text=True)
        k = line(p.stdout)
        print("--prepare %-2s: %.0f replica-steps/s  ms_per_step %.4f  without events %.4f  long_run %.4f" % (prep, k["value"], k["ms_per_step"], k["ms_per_step_without_hip_events"], k["long_run"]["ms_per_step"]))
PY
cat gpurun_out/r05_bench_prep.txt
python -m pytest tests/test_bench_contract.py -q -x -m gpu 2>&1 | tail -3
