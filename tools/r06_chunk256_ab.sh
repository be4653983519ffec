cd tools/ubench
for rep in 1 2 3; do for v in base c256w8o8 c256w8o7 c256w8o6 c256w4o6; do printf "%-10s " $v; ./nd_$v.bin 8192 4096; done; done
for v in base c256w8o8 c256w8o7; do for n in 1024 2048 4096 16384; do printf "%-10s " $v; ./nd_$v.bin $n 4096; done; done
for v in base c256w8o8; do for d in 1023 700 3000; do printf "%-10s " $v; ./nd_$v.bin 300 $d 1.3; done; done
