#!/usr/bin/env python3
"""Where the scratch (spill / stack) instructions of each kernel sit: per kernel, scratch_load / scratch_store counts by the
loop depth of their basic block, read from hipcc's assembly (-S).  Depth 0 is straight-line code around the loops (kernel
prologue, epilogue), the per-element loops of the chunked kernels are depth 2 and deeper.
usage: tools/spill_sites.py [extra -D flags]   (compiles both translation units to /tmp; a few minutes)"""
import collections
import re
import subprocess
import sys

FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-DD377_DCB_K=8", "-DD377_WAVES_PER_SIMD=2"] + sys.argv[1:]
for unit in ("d377", "msm", "codec_chunked"):
    out = "/tmp/spill_sites_%s.s" % unit
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", "decaf377_amd/csrc/%s.hip" % unit, "-o", out],
                          stderr=subprocess.DEVNULL)
    kernel, depth, calls = None, 0, 0
    counts = collections.OrderedDict()
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = re.search(r"k_[a-z0-9_]+", m.group(1))
            kernel = name.group(0) if name else None          # device functions (the out-of-line fallback) are not kernels
            depth = 0
            if kernel and kernel not in counts:
                counts[kernel] = collections.Counter()
            continue
        if kernel is None:
            continue
        if line.startswith(".LBB") or line.lstrip().startswith("; %bb."):
            d = re.search(r"Depth[= ](\d+)", line)
            depth = int(d.group(1)) if d else 0
            continue
        if re.match(r"\s+; =>", line):                           # continuation lines of a loop-header comment
            d = re.search(r"Depth[= ](\d+)", line)
            if d:
                depth = max(depth, int(d.group(1)))
            continue
        ins = line.split()
        if not ins:
            continue
        if ins[0].startswith("scratch_"):
            counts[kernel][(ins[0].split("_")[1], depth)] += 1
        elif ins[0] == "s_swappc_b64":
            counts[kernel][("call", depth)] += 1
        elif ins[0] == "s_endpgm":
            kernel = None
    for k, c in counts.items():
        if not c:
            continue
        by = collections.defaultdict(dict)
        for (kind, d), v in sorted(c.items()):
            by[d][kind] = v
        print("%-26s %s" % (k, "; ".join("depth %d: %s" % (d, ", ".join("%d %s" % (v, kk) for kk, v in sorted(kinds.items()))) for d, kinds in sorted(by.items()))))
