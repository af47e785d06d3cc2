#!/usr/bin/env python3
"""Summarises a tools/collect_pmc.sh output directory: per-kernel averages of every counter, the kernel
stats table, and the JSON record bench.py reads for roofline.traffic / valu_insts_per_element.
usage: tools/pmc_summarize.py <dir> [label [elements]]   -> <dir>/pmc_summary.csv, <dir>/pmc_traffic.json, <dir>/kernel_stats.csv
`elements`: records per k_scalar_mul_var launch of the profiled command (default 2^22, bench.py's workload); with it the
record carries valu_insts_per_element = SQ_INSTS_VALU (wave-instructions) / (elements / 64)."""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    """Identity of the kernels the counters were collected on: sha256 over the names and contents of decaf377_amd/csrc/*.
    tests/test_abi.py::test_pmc_record_matches_sources recomputes it, so a profile that predates a kernel change fails."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "decaf377_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else d
    elements = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
    agg = {}          # (kernel, grid, counter) -> [sum, launches]
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0, r["Counter_Name"])
            a = agg.setdefault(k, {})
            disp = r["Dispatch_Id"]
            a[disp] = a.get(disp, 0.0) + float(r["Counter_Value"])          # counters come per XCD / instance: sum
    rows = []
    per_kernel = {}
    for (kern, grid, ctr), disp in sorted(agg.items()):
        vals = list(disp.values())
        avg = sum(vals) / len(vals)
        rows.append((kern, grid, ctr, len(vals), avg))
        per_kernel.setdefault((kern, grid), {})[ctr] = avg
    with open(os.path.join(d, "pmc_summary.csv"), "w") as f:
        f.write("# %s: rocprofv3 --pmc, one pass per counter group; average per launch, summed over XCDs\n" % label)
        f.write("kernel,grid_size,counter,launches,avg_per_launch\n")
        for r in rows:
            f.write("%s,%d,%s,%d,%.1f\n" % r)
    stats = sorted(glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        with open(os.path.join(d, "kernel_stats.csv"), "w") as f:
            for r in csv.DictReader(open(stats[0])):
                if "k_" in r["Name"] and "rocclr" not in r["Name"]:
                    f.write("%s,calls=%s,avg_ns=%s,min_ns=%s,max_ns=%s\n" % (short(r["Name"]), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
    out = {}
    for (kern, grid), c in per_kernel.items():
        if "FETCH_SIZE" not in c and "SQ_INSTS_VALU" not in c:
            continue
        # the largest grid of a kernel is the timed launch (warm-up and extra ops use the same or smaller ones)
        if kern in out and out[kern]["grid_size"] > grid:
            continue
        rec = {"grid_size": grid, "source": label + "/pmc_summary.csv"}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["fetch_size_kib"], rec["write_size_kib"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
            rec["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            rec["correction"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports "
                                 "half of 16 B/lane reads, so fetch is doubled; WRITE_SIZE is exact; Infinity-Cache hits are "
                                 "counted too, so this is an upper bound on HBM bytes")
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                  "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                  "TCC_HIT_sum", "TCC_MISS_sum"):
            if k in c:
                rec[k] = c[k]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            rec["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if kern == "k_scalar_mul_var" and "SQ_INSTS_VALU" in c:
            rec["elements"] = elements
            rec["valu_insts_per_element"] = c["SQ_INSTS_VALU"] / (elements / 64.0)
        out[kern] = rec
    out["_sources"] = {"csrc_sha256": csrc_sha256(), "note": "sha256 over decaf377_amd/csrc/*.{hip,hpp,inc} of the build that was profiled"}
    json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if k.startswith("_"):
            continue
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("correction", "source")})


if __name__ == "__main__":
    main()
