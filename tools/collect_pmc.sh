#!/bin/bash
# Collects rocprofv3 evidence for the exact bench.py command, on the GPU box (run through gpurun):
#   1. kernel trace + stats (per-kernel average durations)
#   2. PMC counters, one pass per group (the groups do not fit one pass; --pmc is never combined with
#      tracing domains other than --kernel-trace)
# usage: tools/collect_pmc.sh <outdir> [bench args]      every rocprofv3 run is wrapped in `timeout`
set -u
out=$1; shift
mkdir -p "$out"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/stats" -- $BENCH > "$ROOT/$out/bench_under_rocprof.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 420 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$out/pmc$i" -- $BENCH --no-extra > "$ROOT/$out/pmc$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
cd "$ROOT"
python3 tools/pmc_summarize.py "$out"
