#!/bin/bash
# rocprofv3 counters of the normal-generator harness (tools/ubench/normals_dev*.bin): usage tools/r04_normals_pmc.sh <bin> [<bin> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in "$@"; do
  O=$R/gpurun_out/r04/pmc_$b; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o kt -- $R/tools/ubench/$b.bin 8192 4096 > $O/kt.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O -o sq -- $R/tools/ubench/$b.bin 8192 4096 > $O/sq.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O -o sq2 -- $R/tools/ubench/$b.bin 8192 4096 > $O/sq2.log 2>&1
  python3 $R/tools/rocpd_summary.py $O/kt_results.db $O/sq_results.db $O/sq2_results.db > $O/summary.txt 2>&1
  echo "=== $b"; cut -c1-150 $O/summary.txt | grep -v "^== "
  rm -f $O/*.db
done
