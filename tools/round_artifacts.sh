#!/bin/bash
# Everything profiles/rNN_* is made of, in one run on the GPU box (through gpurun):
#   tools/round_artifacts.sh gpurun_out/r03/final
# kernel stats + PMC passes of the exact bench.py command (pmc_traffic.json carries the hash of csrc/ they belong to), the
# plain bench line, the size sweep, the MSM per-kernel breakdown, PMC per operation at 2^20, the multi-GPU self-test.
# Every rocprofv3 run is wrapped in `timeout`; --pmc is only ever combined with --kernel-trace.
set -u
out=${1:-gpurun_out/final}
ROOT=$(pwd)
mkdir -p "$out"
bash tools/collect_pmc.sh "$out/pmc" > "$out/collect_pmc.log" 2>&1
echo "collect_pmc done"
timeout 600 python3 bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
echo "bench done: $(tail -c 300 "$out/bench_line.json" | head -c 120)"
timeout 600 python3 tools/size_sweep.py > "$out/size_sweep.txt" 2>&1
echo "sweep done"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$out/msmtrace" -- python3 "$ROOT/tools/msm_profile.py" > "$ROOT/$out/msmtrace.log" 2>&1)
python3 tools/msm_breakdown.py "$(find "$out/msmtrace" -name '*kernel_trace.csv' | head -1)" > "$out/msm_kernel_breakdown.txt" 2>&1
echo "msm breakdown done"
bash tools/pmc_ops.sh "$out/pmc_ops20" "sqrt_ratio_zeta,decompress,compress,roundtrip,encode_to_curve,hash_to_curve,scalar_mul_base,msm (Elements)" 1048576 > "$out/pmc_ops_2^20.txt" 2>&1
echo "pmc ops done"
timeout 900 python3 tools/multigpu_selftest.py --log2n 14 > "$out/multigpu_selftest.txt" 2>&1
echo "selftest: $(tail -1 "$out/multigpu_selftest.txt")"
timeout 600 python3 tools/msm_skew_bench.py > "$out/msm_skew.txt" 2>&1
timeout 600 python3 tools/host_path_bench.py > "$out/host_path.txt" 2>&1
timeout 600 python3 tools/hbm_ops_bench.py 22 > "$out/hbm_priced_ops.txt" 2>&1
timeout 900 python3 tools/soak.py 22 3 > "$out/soak.txt" 2>&1
echo "soak: $(grep -c bit-exact "$out/soak.txt") bit-exact, $(grep -c MISMATCH "$out/soak.txt") mismatches"
timeout 600 python3 tools/route_stress.py 200 5 > "$out/route_stress.txt" 2>&1
echo "route stress: $(tail -1 "$out/route_stress.txt")"
rm -rf "$out/msmtrace" "$out"/pmc/pmc[0-9] "$out"/pmc/stats "$out"/pmc_ops20/pmc[0-9]
echo done
