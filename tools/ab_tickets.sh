#!/bin/bash
# tickets A/B: product (chunks by workgroup id: the built-in deal) against the ticket build at 16 / 8 / 4 / 2 elements per lane per chunk
out=$1
: > $out
run() { echo "=== $1 | $3" >> $out; D377_LIB=$2 timeout -k 10 300 python3 -u tools/size_sweep.py --sizes 1048576,4194304 --ops scalar_mul_var,sqrt_ratio_zeta,encode_to_curve --tune "$4" 2>&1 | grep -v "amdgpu.ids\|^one MI355X\|^$" >> $out; }
P=$PWD/decaf377_amd/lib/libdecaf377_amd.so; T=$PWD/build/variants/tickets.so
for rep in 1 2; do
  run product $P "built-in deal" ""
  run tickets $T "built-in deal, by ticket" ""
  run tickets $T "8 per lane, by ticket" "chunk_per_lane=8"
  run tickets $T "4 per lane, by ticket" "chunk_per_lane=4"
  run tickets $T "2 per lane, by ticket" "chunk_per_lane=2"
  run product $P "8 per lane, by id" "chunk_per_lane=8"
done
cat $out
