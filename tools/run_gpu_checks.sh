# usage (on the GPU box, via gpurun): bash tools/run_gpu_checks.sh
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/tests.log
cat gpurun_out/tests.log
for impl in 3 2 1; do
  echo "== slice impl $impl"; PTE_SLICE_IMPL=$impl python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
