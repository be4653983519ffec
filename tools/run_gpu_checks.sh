# usage (on the GPU box, via gpurun): bash tools/run_gpu_checks.sh [pytest -k expression]
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "${1:-slice or quickstart or sharded or full_size}" 2>&1 | tail -8 > gpurun_out/tests.log
cat gpurun_out/tests.log
run() { echo "== $1"; env $1 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
run "PTE_SLICE_IMPL=1"
run "PTE_SLICE_IMPL=2"
run "PTE_SLICE_IMPL=5"
run "PTE_SLICE_IMPL=7"
