#!/usr/bin/env python3
"""Sweep of the counting sort's slices per window (tuning key msm_slices) at 2^20 and 2^22 Elements. Dev tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
for lg in (20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    for s in (1, 2, 4, 8, 16, 29, 64):
        with ctx.tuning(msm_slices=s):
            ctx.msm(P, k); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): ctx.msm(P, k)
            torch.cuda.synchronize()
            print("slices", s, "n=2^%d %.3f ms" % (lg, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
