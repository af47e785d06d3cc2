#!/usr/bin/env python3
"""Per-kernel microseconds of every MSM call in a rocprofv3 --kernel-trace CSV (a call ends with k_msm_final, or with k_msm_small_sum on the small-batch route).  Dev tool.
usage: tools/msm_breakdown.py <kernel_trace.csv> [labels...]   one output line per call; with labels, one label per call"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
labels = sys.argv[2:]
calls, cur, t0 = [], collections.OrderedDict(), None
for r in rows:
    m = re.search(r"k_msm_\w+", r["Kernel_Name"])
    if not m:
        continue
    short = m.group(0)
    if t0 is None:
        t0 = int(r["Start_Timestamp"])
    cur[short] = cur.get(short, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if short in ("k_msm_final", "k_msm_small_sum"):
        cur["_span"] = (int(r["End_Timestamp"]) - t0) / 1e3
        calls.append(cur)
        cur, t0 = collections.OrderedDict(), None
for i, c in enumerate(calls):
    span = c.pop("_span")
    names = "  ".join("%s %.0f" % (k.replace("k_msm_", ""), v) for k, v in c.items())
    print("%-22s kernels %6.0f us (first start to last end %6.0f)   %s" % (labels[i] if i < len(labels) else "call %d" % i, sum(c.values()), span, names))
