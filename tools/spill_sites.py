#!/usr/bin/env python3
"""Where the spill instructions of each kernel sit: per kernel, scratch_load / scratch_store (VGPR spills, stack) and
v_writelane_b32 / v_readlane_b32 (SGPR spills parked in VGPR lanes, and the few lane moves a kernel writes itself) counted by
the loop depth of their basic block, read from hipcc's assembly (-S).  Depth 0 is straight-line code around the loops
(kernel prologue, epilogue); in the chunked kernels depth 1 is the walk over chunks, depth 2 the per-element loops, depth 3
and deeper the loops inside an element (the window loop of a scalar multiplication, its four doublings).
usage: tools/spill_sites.py [--units a,b] [extra -D flags]   (compiles the translation units to /tmp; a few minutes)"""
import collections
import re
import subprocess
import sys

ARGS = sys.argv[1:]
UNITS = ("d377", "msm", "codec_chunked", "batch_msm")
if ARGS and ARGS[0] == "--units":
    UNITS = tuple(ARGS[1].split(","))
    ARGS = ARGS[2:]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-DD377_DCB_K=8", "-DD377_WAVES_PER_SIMD=2"] + ARGS
for unit in UNITS:
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
        elif ins[0] in ("v_writelane_b32", "v_readlane_b32"):
            counts[kernel][("writelane" if ins[0] == "v_writelane_b32" else "readlane", depth)] += 1
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
