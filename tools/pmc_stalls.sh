#!/bin/bash
# Where do the waves of a kernel wait?  Extra PMC groups (instruction fetch, scalar memory, LDS, wait reasons) per kernel for
# chosen operations (tools/size_sweep.py as the workload); separate rocprofv3 --pmc passes, only --kernel-trace beside them.
# usage: tools/pmc_stalls.sh <outdir> "<op,op,...>" "<size,...>"
set -u
out=$(realpath -m "$1"); ops=$2; sizes=$3
ROOT=$(realpath "$(dirname "$0")/..")
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$out/sq_counters.txt"
i=0
failed=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INST_LEVEL_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc$i" -- python3 "$ROOT/tools/size_sweep.py" --ops "$ops" --sizes "$sizes" > "$out/pmc$i.log" 2>&1
  rc=$?
  nf=$(find "$out/pmc$i" -name '*counter_collection.csv' 2>/dev/null | wc -l)
  echo "pass $i ($grp): rc=$rc counter files=$nf"
  [ "$rc" -eq 0 ] && [ "$nf" -gt 0 ] || failed=$((failed+1))
done
cd "$ROOT"
[ "$failed" -eq 0 ] || { echo "$failed pass(es) failed or produced no counter file: no summary" >&2; exit 1; }
python3 - "$out" <<'P'
import csv, glob, os, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if not m or m.group(1).startswith("k_init"): continue
        key = (m.group(1), int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for key in sorted(agg):
    c = {k: v / max(1, len(disp[key][k])) for k, v in agg[key].items()}
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0))
    print("%-22s grid %8d" % key)
    for k in sorted(c):
        print("    %-24s %14.5g   / wave_cycles %.4f" % (k, c[k], c[k] / wc))
P
