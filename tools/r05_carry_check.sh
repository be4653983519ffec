cd tools/ubench
ok=1
for d in 4096 1024 1536 5000 700 512 513 1023 1025 2048 100 3000 8192; do
  for pair in "carry nocarry" "carry_cut nocarry_cut"; do
    set -- $pair
    a=$(./nd_$1.bin 300 $d 1.3 | sed 's/.*checksum //'); b=$(./nd_$2.bin 300 $d 1.3 | sed 's/.*checksum //')
    if [ "$a" != "$b" ]; then echo "MISMATCH d=$d $1 $a vs $2 $b"; ok=0; fi
  done
done
echo "checksum battery ok=$ok"
for rep in 1 2 3; do for v in nocarry carry; do printf "%-8s " $v; ./nd_$v.bin 8192 4096; done; done
for v in nocarry carry; do printf "%-8s " $v; ./nd_$v.bin 1024 4096; printf "%-8s " $v; ./nd_$v.bin 32768 4096; printf "%-8s " $v; ./nd_$v.bin 8192 1024; done
