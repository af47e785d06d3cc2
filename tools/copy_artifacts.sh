#!/bin/bash
# Copies what tools/round_artifacts.sh produced (merged back by gpurun) into profiles/ under this round's names.
# usage: tools/copy_artifacts.sh gpurun_out/r04/final r04
F=$1; R=$2
cp $F/pmc/pmc_traffic.json profiles/pmc_traffic.json
sed -i "s#\"source\": \"$F/pmc/pmc_summary.csv\"#\"source\": \"profiles/${R}_pmc_summary.csv\"#" profiles/pmc_traffic.json
cp $F/pmc/pmc_summary.csv profiles/${R}_pmc_summary.csv
cp $F/pmc/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
grep "^{" $F/pmc/bench_under_rocprof.log | tail -1 > profiles/${R}_bench_line_under_rocprof.json
tail -1 $F/bench_line.json > profiles/${R}_bench_line.json
grep -v "amdgpu.ids" $F/size_sweep.txt > profiles/${R}_size_sweep.txt
cp $F/msm_kernel_breakdown.txt profiles/${R}_msm_kernel_breakdown.txt
grep -v "^pass\|amdgpu.ids" "$F/pmc_ops_2^20.txt" > "profiles/${R}_pmc_ops_2^20.txt"
cp $F/multigpu_selftest.txt profiles/${R}_multigpu_selftest_1gpu.txt
grep -v "amdgpu.ids" $F/msm_skew.txt > profiles/${R}_msm_skew.txt
grep -v "amdgpu.ids" $F/soak.txt > profiles/${R}_soak.txt
grep -v "amdgpu.ids" $F/host_path.txt > profiles/${R}_host_path.txt
grep -v "amdgpu.ids" $F/hbm_priced_ops.txt > profiles/${R}_hbm_priced_ops.txt
[ -f $F/route_stress.txt ] && grep -v "amdgpu.ids" $F/route_stress.txt > profiles/${R}_route_stress.txt
[ -f $F/msm_small_bench.txt ] && grep -v "amdgpu.ids" $F/msm_small_bench.txt > profiles/${R}_msm_small_bench.txt
for f in size_sweep_quarter size_sweep_msm row_ops; do [ -f $F/$f.txt ] && grep -v "amdgpu.ids\|warning\|hipFree\|\^~\|^ *[0-9]* |" $F/$f.txt > profiles/${R}_$f.txt; done
[ -f $F/clock_vs_traffic.txt ] && cp $F/clock_vs_traffic.txt profiles/${R}_clock_vs_traffic_rerun.txt
tools/resource_usage.sh > profiles/${R}_resource_usage.txt 2>/dev/null
# no artefact may be empty (round 5 committed four empty files)
for f in profiles/${R}_*; do [ -s "$f" ] || { echo "EMPTY artefact: $f" >&2; exit 1; }; done
# the release check of the PMC record against the kernel sources in the tree (tests/test_abi.py, skipped in the ordinary suites)
D377_CHECK_ARTEFACTS=1 python3 -m pytest tests/test_abi.py -q -k pmc_record | tail -1
