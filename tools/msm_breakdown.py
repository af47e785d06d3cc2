#!/usr/bin/env python3
"""Per-kernel microseconds of one MSM call from a rocprofv3 --kernel-trace CSV of tools/msm_bench.py. Dev tool.
usage: tools/msm_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
calls, cur = [], None
for r in rows:
    m = re.search(r"k_msm_\w+", r["Kernel_Name"])
    if not m:
        continue
    short = m.group(0)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if short.startswith("k_msm_prepare"):
        cur = collections.OrderedDict()
        calls.append(cur)
    cur[short] = cur.get(short, 0) + d
for i, c in enumerate(calls[3::4]):          # msm_bench.py: one warm-up + three timed calls per (size, input form)
    lg = (12, 16, 20, 22)[i // 3]
    form = ("Z=1 ", "", "Z!=1")[i % 3]
    names = "  ".join("%s %.0f" % (k.replace("k_msm_", "").replace("prepare_el", "prepare<el>").replace("prepare_enc", "prepare<enc>"), v)
                      for k, v in c.items())
    print("n=2^%d %-4s total %6.0f us   %s" % (lg, form, sum(c.values()), names))
