cd tools/ubench
for rep in 1 2 3; do for v in base c768w4o4 c768w2o4 c1024w4o3; do printf "%-10s " $v; ./nd_$v.bin 8192 4096; done; done
for v in base c768w4o4 c1024w4o3; do for n in 1024 4096 16384; do printf "%-10s " $v; ./nd_$v.bin $n 4096; done; done
