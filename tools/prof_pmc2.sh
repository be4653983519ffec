# usage: bash tools/prof_pmc2.sh <tag> "<counters>" [env assignments]   -- one PMC pass of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; CTRS=$2; shift; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/pmc_$TAG
rocprofv3 --pmc $CTRS -d $R/gpurun_out/pmc_$TAG -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 $R/tools/rocpd_summary.py $R/gpurun_out/pmc_$TAG/p_results.db | grep -E "slice" | cut -c1-110
