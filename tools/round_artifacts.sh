#!/bin/bash
# Everything profiles/rNN_* is made of, in one run on the GPU box (through gpurun):
#   tools/round_artifacts.sh gpurun_out/r04/final
# kernel stats + PMC passes of the exact bench.py command (pmc_traffic.json carries the hash of csrc/ they belong to), the
# plain bench line, the size sweep, the MSM per-kernel breakdown, PMC per operation at 2^20, the multi-GPU self-test.
# Every rocprofv3 run is wrapped in `timeout`; --pmc is only ever combined with --kernel-trace.
# A second argument selects a part (the whole takes longer than one gpurun call allows): `a` = counters, bench line, sweeps, MSM
# breakdown, PMC per operation; `b` = self-test, skew, host path, HBM-priced ops, soak, route stress, round-4 extras.
set -u
out=${1:-gpurun_out/final}
part=${2:-ab}
ROOT=$(pwd)
mkdir -p "$out"
if [[ $part == *a* ]]; then
bash tools/collect_pmc.sh "$out/pmc" > "$out/collect_pmc.log" 2>&1
echo "collect_pmc done"
timeout 600 python3 bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
echo "bench done: $(tail -c 300 "$out/bench_line.json" | head -c 120)"
timeout 600 python3 tools/size_sweep.py > "$out/size_sweep.txt" 2>&1
echo "sweep done"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$out/msmtrace" -- python3 "$ROOT/tools/msm_profile.py" 8 12 16 18 20 22 24 > "$ROOT/$out/msmtrace.log" 2>&1)
python3 tools/msm_breakdown.py "$(find "$out/msmtrace" -name '*kernel_trace.csv' | head -1)" > "$out/msm_kernel_breakdown.txt" 2>&1
echo "msm breakdown done"
bash tools/pmc_ops.sh "$out/pmc_ops20" "sqrt_ratio_zeta,decompress,compress,roundtrip,encode_to_curve,hash_to_curve,scalar_mul_base,msm (Elements)" 1048576 > "$out/pmc_ops_2^20.txt" 2>&1
echo "pmc ops done"
fi
if [[ $part == *b* ]]; then
timeout 900 python3 tools/multigpu_selftest.py --log2n 14 > "$out/multigpu_selftest.txt" 2>&1
echo "selftest: $(tail -1 "$out/multigpu_selftest.txt")"
timeout 600 python3 tools/msm_skew_bench.py > "$out/msm_skew.txt" 2>&1
timeout 600 python3 tools/host_path_bench.py > "$out/host_path.txt" 2>&1
timeout 600 python3 tools/hbm_ops_bench.py 22 > "$out/hbm_priced_ops.txt" 2>&1
timeout 900 python3 tools/soak.py 22 3 > "$out/soak.txt" 2>&1
echo "soak: $(grep -c bit-exact "$out/soak.txt") bit-exact, $(grep -c MISMATCH "$out/soak.txt") mismatches"
timeout 600 python3 tools/route_stress.py 200 5 > "$out/route_stress.txt" 2>&1
echo "route stress: $(tail -1 "$out/route_stress.txt")"
# round 4: quarter-octave sweep of the chunked kernels (ragged chunks), half-octave sweep of the MSM, the lane-spread
# arithmetic's microbenchmarks, clock against table traffic, the release check of the PMC record
timeout 600 python3 tools/size_sweep.py --sizes 65536,77936,92682,110218,131072,155872,185364,220436,262144,311744,370728,440872,524288,623487,741455,881744,1048576,1310720,1572864,1835008,2097152 \
  --ops sqrt_ratio_zeta,encode_to_curve,hash_to_curve,scalar_mul_base,decompress,scalar_mul_var > "$out/size_sweep_quarter.txt" 2>&1
timeout 600 python3 tools/size_sweep.py --sizes 16,256,1024,1025,4096,8192,11585,16384,23170,32768,46341,65536,92682,131072,185364,262144,370728,524288,1048576,2097152,3145727,3145728,4194304,8388608,16777216 \
  --ops "msm (Elements),msm (Encodings)" > "$out/size_sweep_msm.txt" 2>&1
mkdir -p build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared tools/row_proto.hip -o build/row_proto.so > "$out/row_ops.txt" 2>&1
{ timeout 200 python3 tools/row_model.py; timeout 200 python3 tools/row_proto.py; timeout 200 python3 tools/row_point_check.py;
  timeout 200 python3 tools/inv_wave_model.py; timeout 200 python3 tools/row_invert_check.py; } >> "$out/row_ops.txt" 2>&1
echo "row ops: $(grep -c "OK\|DONE" "$out/row_ops.txt") of 5 legs"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Idecaf377_amd/csrc tools/clock_vs_traffic.hip -o tools/clock_vs_traffic > /dev/null 2>&1
bash tools/clock_vs_traffic.sh "$out/clock_vs_traffic.txt" > /dev/null 2>&1
echo "clock vs traffic: $(grep -c MHz "$out/clock_vs_traffic.txt") lines"
fi
rm -rf "$out/msmtrace" "$out"/pmc/pmc[0-9] "$out"/pmc/stats "$out"/pmc_ops20/pmc[0-9]
echo done
