#!/bin/bash
# First contact with a multi-GPU node, one command:  tools/scale8.sh [max_gpus] [out_dir]
# Runs bench.py at 1 / 2 / 4 / 8 GPUs, weak (BASELINE's metric shape on every GPU) and strong (toy_mvn_target(4096), 8192 chains cut over
# the ranks: the north star's scaling clause), one JSON line each into <out_dir>/scale_{weak,strong}_<N>.json, then a table.
# bench.py --gpus N launches its own ranks from a process that never touches the GPU (no exec of a GPU process); every line of a
# multi-rank run carries: transport_library (the file ncclSend / ncclRecv come from + its version), parallelism_invariant (a short seeded
# run of the cut ladder against one engine, bit for bit, on this node's transport), boundary_exchange (HIP events around the grouped
# send / recv), n_ranks_seen, boundary_swaps_per_rank, env_overrides.  A run that fails leaves its stderr in <out_dir>/*.err and the
# table says so; nothing here needs root or changes machine settings.
cd "$(dirname "$0")/.." || exit 1
MAXG=${1:-8}; OUT=${2:-gpurun_out/scale8}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
unset PTE_RCCL_LIB PTE_LIB PTE_BENCH_BACKEND            # a measurement run takes no overrides (bench.py would name them in the line)
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "scale8: $NG GPU(s) visible, running up to $MAXG" | tee "$OUT/README.txt"
# before anything is timed: the real librccl with real peers, G = 2 / 4 / 8, C4's and C5's payloads, bit-identical to one engine
# (tests/test_zz_gpu_rccl_multigpu.py; skipped on a 1-GPU box).  A failure here makes every number below meaningless: say so and stop.
if [ "$NG" -ge 2 ]; then
  timeout 3000 python -m pytest tests/test_zz_gpu_rccl_multigpu.py -m gpu -x -q -s > "$OUT/rccl_multigpu_tests.log" 2>&1 \
    || { echo "FAILED: real-RCCL parity (see $OUT/rccl_multigpu_tests.log); not timing anything"; tail -30 "$OUT/rccl_multigpu_tests.log"; exit 1; }
  grep "real RCCL" "$OUT/rccl_multigpu_tests.log"
fi
for mode in weak strong; do
  for g in 1 2 4 8; do
    [ "$g" -gt "$MAXG" ] && continue
    [ "$g" -gt "$NG" ] && { echo "skip $mode $g: only $NG GPU(s)"; continue; }
    extra="--no-cpu-baseline --no-extra --round-trip-rounds 0 --steps 64 --warmup 8"
    [ "$mode" = strong ] && extra="$extra --scaling strong"
    [ "$mode" = weak ] && [ "$g" = 1 ] && extra="--steps 64 --warmup 8"          # the full single-GPU line once (CPU baseline, extra configs)
    timeout 1500 python bench.py --gpus "$g" $extra > "$OUT/scale_${mode}_${g}.json" 2> "$OUT/scale_${mode}_${g}.err" \
      || echo "FAILED: $mode $g (see $OUT/scale_${mode}_${g}.err)"
  done
done
python - "$OUT" <<'PY'
import json, sys, os, glob
out = sys.argv[1]
print("%-7s %2s  %14s  %9s  %6s  %-9s  %-28s  %s" % ("mode", "G", "replica-steps/s", "ms/step", "x vs 1", "invariant", "boundary exchange us (med)", "transport library"))
for mode in ("weak", "strong"):
    base = None
    for g in (1, 2, 4, 8):
        f = os.path.join(out, "scale_%s_%d.json" % (mode, g))
        try:
            j = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
        except Exception:
            if os.path.exists(f): print("%-7s %2d  (no line: see the .err file)" % (mode, g))
            continue
        base = base or j["value"]
        c = j["config"]; be = c.get("boundary_exchange") or {}; tl = c.get("transport_library") or {}
        print("%-7s %2d  %14.0f  %9.4f  %6.2f  %-9s  %-28s  %s" % (mode, g, j["value"], j["ms_per_step"], j["value"] / base, c.get("parallelism_invariant"),
              ("%.1f" % be["us_min_median_max"][1]) if be else "-", ("%s (v%s)" % (tl.get("path"), tl.get("nccl_version"))) if tl else "-"))
PY
