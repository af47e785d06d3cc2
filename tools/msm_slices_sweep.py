#!/usr/bin/env python3
"""Sweep of the counting sort's slices per window (D377_MSM_SLICES) at 2^20 and 2^22 Elements. Dev tool."""
import os, sys, subprocess
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
for lg in (20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    ctx.msm(P, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ctx.msm(P, k)
    torch.cuda.synchronize()
    print("slices", os.environ.get("D377_MSM_SLICES"), "n=2^%d %.3f ms" % (lg, (time.perf_counter() - t0) / 3 * 1e3))
'''
for s in (1, 2, 4, 8, 16, 29, 64):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, D377_MSM_SLICES=str(s)))
