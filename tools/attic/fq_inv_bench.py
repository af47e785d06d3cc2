"""Batch Fq::inverse and to_affine: us per call at 2^16 / 2^20 / 2^22.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
for lg in (16, 20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    P = ctx.double(P)                                  # Z != 1
    a = P[:, :4].contiguous()                          # X coordinates as Fq records
    for name, fn in (("fq_inverse", lambda: ctx.fq_op("inverse", a)), ("to_affine", lambda: ctx.to_affine(P))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print("%-10s n=2^%d  %9.1f us" % (name, lg, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
