#!/bin/bash
# Per-kernel breakdown of MSM calls (Elements) under tuning overrides: one rocprofv3 --kernel-trace run per override set.
# usage: tools/msm_trace_tuned.sh <outfile> "<log2 sizes>" "key=v key=v" ["key=v ..." ...]     ("" = the built-in rules)
out=$1; sizes=$2; shift 2
ROOT=$(pwd)
mkdir -p "$(dirname "$out")"; : > "$out"
for tune in "$@"; do
  echo "== sizes $sizes tuning: ${tune:-default}" >> "$out"
  rm -rf /tmp/msm_tt
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/msm_tt -- python3 "$ROOT/tools/msm_profile.py" $sizes $tune --elements-only > /tmp/msm_tt.log 2>&1) || { tail -5 /tmp/msm_tt.log >> "$out"; }
  python3 tools/msm_breakdown.py "$(find /tmp/msm_tt -name '*kernel_trace.csv' | head -1)" >> "$out" 2>&1
done
cat "$out"
