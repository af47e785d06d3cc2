#!/usr/bin/env python3
"""Times vartime_multiscalar_mul (Pippenger MSM) at a few sizes on cuda:0. Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d

ctx = d.Context([0])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for lg in (12, 16, 20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    encs = ctx.encode_to_curve(r0)
    P, _ = ctx.decompress(encs)
    P2 = ctx.double(P)                                     # projective (Z != 1): the normalisation's inversion is not skipped
    for name, pts in (("elements", P), ("encodings", encs), ("el. Z!=1", P2)):
        ctx.msm(pts, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ctx.msm(pts, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("n=2^%d %-9s %8.3f ms  %.3e points/s" % (lg, name, dt * 1e3, n / dt))
