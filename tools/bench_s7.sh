#!/bin/bash
# sweep slice7 speculation budgets on the metric workload (run on the GPU box)
for b in "$@"; do
  echo "== PTE_S7_BUDGETS=$b"
  PTE_SLICE_IMPL=7 PTE_S7_BUDGETS=$b python bench.py --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done
