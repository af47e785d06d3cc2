#!/bin/bash
# Which unit do a kernel's vector-memory waits end in?  Address translation (UTCL1 / UTCL2), the texture path (TA / TD / TCP
# stalls), L2 (TCC tag stalls, requests in flight) and the fabric behind it (EA read requests, DRAM credit stalls), per kernel
# and grid size, for chosen operations (tools/size_sweep.py as the workload).  One rocprofv3 --pmc pass per group, only
# --kernel-trace beside it; a pass that fails (a counter the device refuses) is reported and skipped, and the script exits
# non-zero when NO pass produced a counter file.  (The TA_* and TD_* groups are left out: on this pool rocprofv3 never returns
# from a pass that holds them -- two passes, 300 s each, killed by their timeouts in round 6.)
# usage: tools/pmc_translation.sh <outdir> "<op,op,...>" "<size,...>"        (D377_LIB selects a variant build;
#        PMC_GROUPS=<file> replaces the counter groups below: one group per line)
set -u
out=$(realpath -m "$1"); ops=$2; sizes=$3
ROOT=$(realpath "$(dirname "$0")/..")
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
ok=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc$i" -- python3 "$ROOT/tools/size_sweep.py" --ops "$ops" --sizes "$sizes" > "$out/pmc$i.log" 2>&1
  rc=$?
  nf=$(find "$out/pmc$i" -name '*counter_collection.csv' 2>/dev/null | wc -l)
  echo "pass $i ($grp): rc=$rc counter files=$nf"
  [ "$nf" -gt 0 ] && ok=$((ok+1))
done < <(if [ -n "${PMC_GROUPS:-}" ]; then cat "$PMC_GROUPS"; else cat <<'G'
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU
SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum
TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
GRBM_UTCL2_BUSY GRBM_TA_BUSY GRBM_GUI_ACTIVE
G
fi)
cd "$ROOT"
[ "$ok" -gt 0 ] || { echo "no pass produced counters"; exit 1; }
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
    gui = max(1.0, c.get("GRBM_GUI_ACTIVE", 0))
    print("%-24s grid %9d   (per launch; GRBM_GUI_ACTIVE %.4g)" % (key[0], key[1], gui))
    for k in sorted(c):
        print("    %-48s %14.5g   / GUI_ACTIVE %.4f" % (k, c[k], c[k] / gui))
    d = lambda a, b: c.get(a, 0) / max(1.0, c.get(b, 0))
    print("    -- UTCL1 miss rate %.4f; translation misses per VMEM read wave-instruction %.3f; TCP->TCC read latency %.0f cycles/request; "
          "EA read requests in flight (LEVEL / TCC_BUSY) %.1f; L2 hit %.3f" % (
          d("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"), d("TCP_UTCL1_TRANSLATION_MISS_sum", "SQ_INSTS_VMEM_RD"),
          d("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"), d("TCC_EA0_RDREQ_LEVEL_sum", "TCC_BUSY_sum"),
          c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))))
P
rm -rf "$out"/pmc[0-9]*/
