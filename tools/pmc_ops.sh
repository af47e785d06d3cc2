#!/bin/bash
# PMC counters per kernel for chosen operations of the path at chosen sizes (tools/size_sweep.py as the workload):
# separate rocprofv3 --pmc passes, only --kernel-trace beside them.  Prints, per kernel and grid size: VALU instructions
# per element, cycles, issue-slot utilisation (VALU instructions x 4 cycles / SIMD against GRBM_GUI_ACTIVE / 8 XCDs), wait
# share, LDS / VMEM instructions, L2 hit rate, fetched / written bytes (FETCH_SIZE doubled: gfx950 correction of the guide).
# usage: tools/pmc_ops.sh <outdir> "<op,op,...>" "<size,size,...>"      (D377_LIB selects a variant build)
set -u
out=$(realpath -m "$1"); ops=$2; sizes=$3
ROOT=$(realpath "$(dirname "$0")/..")
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
failed=0
for grp in "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc$i" -- python3 "$ROOT/tools/size_sweep.py" --ops "$ops" --sizes "$sizes" > "$out/pmc$i.log" 2>&1
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
print("kernel, grid (threads): per launch")
for key in sorted(agg):
    c = {k: v / max(1, len(disp[key][k])) for k, v in agg[key].items()}
    valu, cyc = c.get("SQ_INSTS_VALU", 0), c.get("GRBM_GUI_ACTIVE", 0) / 8
    util = valu / 1024 * 4 / cyc if cyc else 0
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("%-24s grid %9d  launches %2d  VALU wave-instr %.4g  cycles/XCD %.4g  issue-slot utilisation %.3f  wait_any/wave_cycles %.3f  "
          "active_valu/busy %.3f  LDS instr %.3g  VMEM instr %.3g  SALU %.3g  L2 hit %.3f (hit %.3g miss %.3g)  fetch %.1f MB  write %.1f MB" % (
          key[0], key[1], len(disp[key].get("SQ_INSTS_VALU", [1])), valu, cyc, util,
          c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0)), c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_BUSY_CYCLES", 0)),
          c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM", 0), c.get("SQ_INSTS_SALU", 0), hit / max(1.0, hit + miss), hit, miss,
          2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6))
P
