#!/usr/bin/env python3
"""One warm-up and ONE timed MSM call per (size, input form): the workload tools/msm_breakdown.py reads back from a
rocprofv3 --kernel-trace.  usage: python tools/msm_profile.py [log2 sizes ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d

ctx = d.Context([0])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 20, 22]
for lg in sizes:
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    encs = ctx.encode_to_curve(r0)
    P, _ = ctx.decompress(encs)
    for pts in (P, encs):
        for _ in range(2):
            ctx.msm(pts, k)
            torch.cuda.synchronize()
