#!/bin/bash
# round-4 development runs on the GPU box: instruction rates (parts with their own timeouts) and the normal-generator harness
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for p in 2 3 4 5; do timeout 40 ./tools/ubench/rate2.bin $p > gpurun_out/r04/rate2_part$p.txt 2>&1; echo "part $p rc=$?"; done
for b in normals_dev_r03 normals_dev; do
  for rep in 1 2; do timeout 60 ./tools/ubench/$b.bin 8192 4096; done
  timeout 60 ./tools/ubench/$b.bin 32768 4096
  timeout 60 ./tools/ubench/$b.bin 1000 1000 1.0
  timeout 60 ./tools/ubench/$b.bin 777 70 3.3
done 2>&1 | tee gpurun_out/r04/normals_dev.txt
