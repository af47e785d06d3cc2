#!/usr/bin/env python3
"""One warm-up and ONE timed MSM call per (size, input form): the workload tools/msm_breakdown.py reads back from a
rocprofv3 --kernel-trace.  usage: python tools/msm_profile.py [log2 sizes ...] [key=value ...] [--elements-only]
key=value: tuning overrides (engine.TUNE_KEYS, e.g. msm_seg=64 msm_window=16) in force for every call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d

ctx = d.Context([0])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
args = [a for a in sys.argv[1:] if a != "--elements-only"]
elements_only = "--elements-only" in sys.argv
for a in args:
    if "=" in a:
        ctx.set_tuning(a.split("=")[0], int(a.split("=")[1]))
sizes = [int(a) for a in args if "=" not in a] or [12, 16, 20, 22]
for lg in sizes:
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    encs = ctx.encode_to_curve(r0)
    P, _ = ctx.decompress(encs)
    for pts in ((P,) if elements_only else (P, encs)):
        for _ in range(2):
            ctx.msm(pts, k)
            torch.cuda.synchronize()
