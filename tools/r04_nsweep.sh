#!/bin/bash
# normal generator: time against rows and row length (what is the fixed ~15 us of a launch made of?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for n in 64 256 1024 2048; do printf "small-N  "; timeout 60 ./tools/ubench/normals_dev.bin $n 4096; done 2>&1 | tee gpurun_out/r04/nsweep2.txt
for d in 256 512 1024 2048 4096; do for n in 5120 8192; do printf "d-sweep  "; timeout 60 ./tools/ubench/normals_dev.bin $n $d; done; done 2>&1 | tee -a gpurun_out/r04/nsweep2.txt
