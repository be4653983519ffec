#!/bin/bash
# A/B the slice kernel versions on the metric workload (run on the GPU box).
for impl in "$@"; do
  echo "== PTE_SLICE_IMPL=$impl"
  PTE_SLICE_IMPL=$impl python bench.py --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done
