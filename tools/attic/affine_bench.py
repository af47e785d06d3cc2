#!/usr/bin/env python3
"""to_affine at 1 / 2 / 4 / 8 workgroups per CU sharing the batch (tuning key affine_blocks_per_cu). Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for lg in (16, 20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    out = torch.empty((n, 8), dtype=torch.int64, device=dev)
    for bpc in (1, 2, 4, 8):
        with ctx.tuning(affine_blocks_per_cu=bpc):
            ctx.to_affine(P, outs=[out]); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): ctx.to_affine(P, outs=[out])
            b.record(); torch.cuda.synchronize()
            print("blocks/CU", bpc, "n=2^%d" % lg, "%.3f ms" % (a.elapsed_time(b) / 5), flush=True)
