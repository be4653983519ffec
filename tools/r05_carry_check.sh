# two builds of tools/ubench/normals_dev.hip (nd_$1.bin / nd_$2.bin and their -DNRM_MAX_EV_=2 twins nd_$1_cut / nd_$2_cut) must print the same
# checksum over a battery of row lengths; then their speed, interleaved
cd tools/ubench
A=${1:-pc}; B=${2:-base}
ok=1
for d in 4096 1024 1536 700 512 513 1023 1025 2048 100 3000 4095; do
  for pair in "$A $B" "${A}_cut ${B}_cut"; do
    set -- $pair
    a=$(./nd_$1.bin 300 $d 1.3 | sed 's/.*checksum //'); b=$(./nd_$2.bin 300 $d 1.3 | sed 's/.*checksum //')
    if [ "$a" != "$b" ]; then echo "MISMATCH d=$d $1 $a vs $2 $b"; ok=0; fi
  done
done
echo "checksum battery ok=$ok"
for rep in 1 2 3; do for v in $B $A; do printf "%-8s " $v; ./nd_$v.bin 8192 4096; done; done
for v in $B $A; do printf "%-8s " $v; ./nd_$v.bin 1024 4096; printf "%-8s " $v; ./nd_$v.bin 32768 4096; printf "%-8s " $v; ./nd_$v.bin 4096 4096; done
