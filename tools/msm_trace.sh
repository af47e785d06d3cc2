#!/bin/bash
# Per-kernel breakdown of MSM calls at the given log2 sizes: rocprofv3 --kernel-trace of tools/msm_profile.py, read by
# tools/msm_breakdown.py.  usage: tools/msm_trace.sh <outdir> <log2 sizes...>
out=$1; shift
ROOT=$(pwd)
mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$out/trace" -- python3 "$ROOT/tools/msm_profile.py" "$@" > "$ROOT/$out/trace.log" 2>&1)
python3 tools/msm_breakdown.py "$(find "$out/trace" -name '*kernel_trace.csv' | head -1)" > "$out/msm_kernel_breakdown.txt" 2>&1
rm -rf "$out/trace"
cat "$out/msm_kernel_breakdown.txt"
