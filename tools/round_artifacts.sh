#!/bin/bash
# Everything profiles/rNN_* is made of, in one run on the GPU box (through gpurun):
#   tools/round_artifacts.sh gpurun_out/r06/final [a|b|ab]
# `a` = kernel stats + PMC passes of the exact bench.py command (pmc_traffic.json carries the hash of csrc/ they belong to), the
# plain bench line, the size sweeps, the MSM per-kernel breakdown, PMC per operation at 2^20; `b` = multi-GPU self-test, MSM
# skew, host path, HBM-priced ops, soak, route stress, the small-sums bench, the lane-spread arithmetic's checks, clock against
# table traffic.  Every rocprofv3 run is wrapped in `timeout`; --pmc is only ever combined with --kernel-trace.
# Every leg's output is checked: a leg that fails, or leaves an EMPTY file, is listed at the end and the script exits 1 -- an
# empty artefact is a run that printed nothing, not evidence (round 5 committed four of them).  python runs unbuffered (-u), so
# a leg that is killed by its timeout still leaves what it had printed.
set -u
out=${1:-gpurun_out/final}
part=${2:-ab}
ROOT=$(pwd)
mkdir -p "$out"
FAILED=()
# leg <seconds> <outfile> <command ...>: stdout + stderr -> outfile
leg() {
  local secs=$1 file=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$file" 2>&1
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s "$file" ]; then FAILED+=("$(basename "$file") (rc=$rc, $(wc -c < "$file") bytes)"); fi
  echo "$(basename "$file"): rc=$rc, $(wc -l < "$file") lines"
}
export PYTHONUNBUFFERED=1
if [[ $part == *a* ]]; then
  leg 900 "$out/collect_pmc.log" bash tools/collect_pmc.sh "$out/pmc"
  timeout -k 10 600 python3 -u bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
  [ -s "$out/bench_line.json" ] && grep -q '"roofline"' "$out/bench_line.json" || FAILED+=("bench_line.json")
  echo "bench: $(head -c 160 "$out/bench_line.json")"
  leg 700 "$out/size_sweep.txt" python3 -u tools/size_sweep.py
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$out/msmtrace" -- python3 "$ROOT/tools/msm_profile.py" 8 12 16 18 20 22 24 > "$ROOT/$out/msmtrace.log" 2>&1)
  leg 120 "$out/msm_kernel_breakdown.txt" python3 -u tools/msm_breakdown.py "$(find "$out/msmtrace" -name '*kernel_trace.csv' | head -1)"
  leg 900 "$out/pmc_ops_2^20.txt" bash tools/pmc_ops.sh "$out/pmc_ops20" "sqrt_ratio_zeta,decompress,compress,roundtrip,encode_to_curve,hash_to_curve,scalar_mul_base,msm (Elements),msm_small (3 terms)" 1048576
fi
if [[ $part == *b* ]]; then
  leg 900 "$out/multigpu_selftest.txt" python3 -u tools/multigpu_selftest.py --log2n 14
  leg 600 "$out/msm_skew.txt" python3 -u tools/msm_skew_bench.py
  leg 600 "$out/host_path.txt" python3 -u tools/host_path_bench.py
  leg 600 "$out/hbm_priced_ops.txt" python3 -u tools/hbm_ops_bench.py 22
  leg 900 "$out/soak.txt" python3 -u tools/soak.py 22 3
  grep -q "bit-exact" "$out/soak.txt" && ! grep -q MISMATCH "$out/soak.txt" || FAILED+=("soak.txt: $(grep -c bit-exact "$out/soak.txt") bit-exact, $(grep -c MISMATCH "$out/soak.txt") mismatches")
  leg 600 "$out/route_stress.txt" python3 -u tools/route_stress.py 200 5
  leg 400 "$out/msm_small_bench.txt" python3 -u tools/msm_small_bench.py
  # quarter-octave sweep of the chunked kernels (ragged chunks), half-octave sweep of the MSM
  leg 600 "$out/size_sweep_quarter.txt" python3 -u tools/size_sweep.py --sizes 65536,77936,92682,110218,131072,155872,185364,220436,262144,311744,370728,440872,524288,623487,741455,881744,1048576,1310720,1572864,1835008,2097152 \
    --ops sqrt_ratio_zeta,encode_to_curve,hash_to_curve,scalar_mul_base,decompress,scalar_mul_var
  leg 600 "$out/size_sweep_msm.txt" python3 -u tools/size_sweep.py --sizes 16,256,1024,1025,4096,8192,11585,16384,23170,32768,46341,65536,92682,131072,185364,262144,370728,524288,1048576,2097152,3145727,3145728,4194304,8388608,16777216 \
    --ops "msm (Elements),msm (Encodings)"
  # the lane-spread arithmetic's models and GPU checks
  mkdir -p build
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared tools/row_proto.hip -o build/row_proto.so > "$out/row_ops.txt" 2>&1
  for t in row_model.py row_proto.py row_point_check.py inv_wave_model.py row_invert_check.py; do timeout -k 10 200 python3 -u tools/$t >> "$out/row_ops.txt" 2>&1 || FAILED+=("row_ops: $t"); done
  echo "row ops: $(grep -c "OK\|DONE" "$out/row_ops.txt") of 5 legs"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Idecaf377_amd/csrc tools/clock_vs_traffic.hip -o tools/clock_vs_traffic > /dev/null 2>&1
  bash tools/clock_vs_traffic.sh "$out/clock_vs_traffic.txt" > /dev/null 2>&1
  [ -s "$out/clock_vs_traffic.txt" ] || FAILED+=("clock_vs_traffic.txt")
  echo "clock vs traffic: $(grep -c MHz "$out/clock_vs_traffic.txt") lines"
fi
rm -rf "$out/msmtrace" "$out"/pmc/pmc[0-9] "$out"/pmc/stats "$out"/pmc_ops20/pmc[0-9]
if [ ${#FAILED[@]} -ne 0 ]; then
  echo "FAILED or EMPTY artefacts:"; printf '  %s\n' "${FAILED[@]}"
  exit 1
fi
echo done
