# instruction-cache counters of every config's explore kernel (is a kernel's code too large for the 64 KB instruction cache?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_icache; mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES -d $O -o ic -- python3 $R/tools/bench_configs.py > $O/ic.log 2>&1
python3 $R/tools/rocpd_summary.py $O/ic_results.db > $O/summary.txt
grep -E "k_explore_automala<2|k_explore_slice8<4|k_explore_ising_spec|k_explore_toy<6" $O/summary.txt | grep -E "ICACHE|IFETCH|WAVE_CYCLES" | cut -c1-140
