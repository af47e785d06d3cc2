#!/usr/bin/env python3
"""Elements per lane per chunk (tuning key chunk_per_lane) for the chunked kernels at chosen sizes: us per call.  Dev tool.
usage: python tools/chunk_sweep.py <log2n,...> [per_lane,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d

sizes = [int(x) for x in sys.argv[1].split(",")]
pls = [None] + [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,3,4,5,6,8").split(",")]
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(3)
nmax = 1 << max(sizes)
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
o32 = torch.empty((nmax, 32), dtype=torch.uint8, device=dev); o1 = torch.empty((nmax,), dtype=torch.uint8, device=dev)
ops = {
    "sqrt_ratio_zeta": lambda n: ctx.sqrt_ratio_zeta(r0[:n], k[:n], outs=[o32[:n], o1[:n]]),
    "encode_to_curve": lambda n: ctx.encode_to_curve(r0[:n], outs=[o32[:n]]),
    "hash_to_curve": lambda n: ctx.hash_to_curve(r0[:n], k[:n], outs=[o32[:n]]),
    "scalar_mul_base": lambda n: ctx.scalar_mul_base(k[:n], outs=[o32[:n]]),
    "scalar_mul_var": lambda n: ctx.scalar_mul_var(enc[:n], k[:n], outs=[o32[:n], o1[:n]]),
}
for name, fn in ops.items():
    for lg in sizes:
        n = 1 << lg
        row = []
        for pl in pls:
            with ctx.tuning(chunk_per_lane=pl):
                fn(n); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5 if name != "scalar_mul_var" else 2
                e0.record()
                for _ in range(reps): fn(n)
                e1.record(); torch.cuda.synchronize()
            row.append("%s: %.1f" % ("rule" if pl is None else pl, e0.elapsed_time(e1) / reps * 1e3))
        print("%-16s n=2^%d  us per call by elements per lane per chunk   %s" % (name, lg, "   ".join(row)), flush=True)
