#!/bin/bash
# tools/sort_shape_ab.py over builds of the sort's shapes (build/variants/st*.so), A B ... then again in reverse order
out=$1
: > "$out"
for pass in 1 2; do
  libs="st1024x8:0,16,32,64 st512x8:0,32,48,64,96 st512x16:0,16,32,64 st256x16:0,32,64,96,128"
  [ $pass = 2 ] && libs=$(echo $libs | tr ' ' '\n' | tac | tr '\n' ' ')
  for ls in $libs; do
    lib=${ls%%:*}; sl=${ls##*:}
    D377_LIB=build/variants/$lib.so timeout -k 10 300 python3 -u tools/attic/sort_shape_ab.py $sl 2>&1 | grep -v amdgpu.ids >> "$out" || echo "FAILED $lib" >> "$out"
  done
done
