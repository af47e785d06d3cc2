#!/bin/bash
# Builds build/variants/<name>.so from the current sources with extra -D flags (A/B runs: tools/ab_bench.sh).
# usage: tools/build_variant.sh name [-DD377_WAVES_PER_SIMD=3 ...]
# Goes through the Makefile: a failed compile fails the script, and the objects are rebuilt whenever the flags change
# (no stale build/obj_<name>/*.o from an earlier variant of the same name).
set -e
name=$1; shift
make -j2 lib VARIANT="$name" EXTRA="$*"
echo built build/variants/$name.so
