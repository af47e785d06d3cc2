#!/bin/bash
# Builds build/variants/<name>.so from the current sources with extra -D flags (A/B runs: tools/ab_bench.sh).
# usage: tools/build_variant.sh name [-DD377_WAVES_PER_SIMD=3 ...]
set -e
name=$1; shift
mkdir -p build/variants build/obj_$name
for u in d377 msm; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c decaf377_amd/csrc/$u.hip -o build/obj_$name/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/variants/$name.so build/obj_$name/d377.o build/obj_$name/msm.o
echo built build/variants/$name.so
