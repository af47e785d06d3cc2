#!/bin/bash
# Review item 6 (round 4): clock against table traffic for the variable-base window loop.  tools/clock_vs_traffic twice:
# plain (HIP-event times, in-kernel clock ratio) and under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE (cycles per
# dispatch / dispatch duration = the frequency the chip sustained).  usage: tools/clock_vs_traffic.sh <outfile>
out=${1:-gpurun_out/clock_vs_traffic.txt}
ROOT=$(pwd)
mkdir -p "$(dirname "$out")"
{
  echo "# tools/clock_vs_traffic.hip on one MI355X (tools/clock_vs_traffic.sh)"
  timeout -k 10 300 ./tools/clock_vs_traffic
  echo
  echo "# the same program under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE: cycles (sum over the 8 XCDs / 8) per dispatch / its duration"
} > "$out" 2>&1
tmp=$ROOT/gpurun_out/cvt_prof
rm -rf "$tmp"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$tmp" -- "$ROOT/tools/clock_vs_traffic" > "$tmp.log" 2>&1)
python3 - "$tmp" >> "$out" <<'P'
import csv, glob, os, sys
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not cc or not kt:
    print("no counter / trace output (see gpurun_out/cvt_prof.log)"); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cyc = {}
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for disp in sorted(cyc, key=int):
    if disp in dur and "k_loop" in dur[disp][1]:
        ns, name = dur[disp]
        kind = "gathers  " if "Lb1" in name or "<true>" in name else "registers"
        print("%s dispatch %3s  %8.2f ms  GRBM_GUI_ACTIVE/8 %.4e  -> %7.1f MHz" % (kind, disp, ns / 1e6, cyc[disp] / 8, cyc[disp] / 8 / ns * 1e3))
P
rm -rf "$tmp"
cat "$out"
