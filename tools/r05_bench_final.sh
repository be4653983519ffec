#!/bin/bash
# the driver's bench command on the final library, its line summarised, then the contract tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05_bench_final.json") if l.startswith("{")][-1])
print("value %.0f  ms_per_step %.4f  long_run %.4f  roofline.frac %.5f  launch/scan %.4f" % (j["value"], j["ms_per_step"], j["long_run"]["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms_per_scan"]))
for e in j["extra_configs"]: print("  %-60s %-24s %.4f ms/scan" % (e["config"][:60], e["scan_loop"][:24], e["ms_per_scan"]))
print({k: (round(v.get("frac_of_6.29TBps", 0), 4), round(v.get("avg_launch_us", 0), 1)) for k, v in j["hbm_kernels"].items() if isinstance(v, dict)})
PY
python -m pytest tests/test_bench_contract.py -q -x -m gpu 2>&1 | tail -2
