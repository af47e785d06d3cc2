#!/usr/bin/env python3
"""The HBM-priced entry points (a few field products per record at most): kernel time and algorithmic GB/s at 2^22
records resident in HBM, against the 8 TB/s peak.  Dev tool; numbers go to profiles/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d

ctx = d.Context([0])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
r1 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
P = ctx.encode_to_curve_element(r0)
Q = ctx.double(P)
a, _ = ctx.fq_from_bytes_checked(ctx.fq_from_wide_bytes(torch.cat([r0, r1[:, :16]], dim=1).contiguous()))
b, _ = ctx.fq_from_bytes_checked(ctx.fq_from_wide_bytes(torch.cat([r1, r0[:, :16]], dim=1).contiguous()))
oE, oE2 = torch.empty_like(P), torch.empty_like(P)
o32, o32b = torch.empty_like(r0), torch.empty_like(r0)
oF = torch.empty((n,), dtype=torch.uint8, device=dev)
oA = torch.empty((n, 8), dtype=torch.int64, device=dev)
w64 = torch.cat([r0, r1], dim=1).contiguous()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [
    ("neg", 256, lambda: ctx.neg(P, outs=[oE])),
    ("double", 256, lambda: ctx.double(P, outs=[oE])),
    ("add", 384, lambda: ctx.add(P, Q, outs=[oE])),
    ("eq", 257, lambda: ctx.eq(P, Q, outs=[oF])),
    ("is_identity", 129, lambda: ctx.is_identity(P, outs=[oF])),
    ("to_affine", 192, lambda: ctx.to_affine(Q, outs=[oA])),
    ("fq mul", 96, lambda: ctx.fq_op("mul", a, b)),
    ("fr mul", 97, lambda: ctx.fr_op("mul", r0, r1, outs=[o32, oF])),
    ("fr add", 97, lambda: ctx.fr_op("add", r0, r1, outs=[o32, oF])),
    ("fr from 64 bytes", 96, lambda: ctx.fr_from_wide_bytes(w64)),
    ("fq from 64 bytes", 96, lambda: ctx.fq_from_wide_bytes(w64)),
    ("fr_from_le_bytes_mod_order(32) [host-only entry point skipped]", 0, None),
]
print("n = 2^%d records; HBM peak 8000 GB/s" % (n.bit_length() - 1))
for name, bytes_per, fn in cases:
    if fn is None:
        continue
    ms = timed(fn)
    gbs = bytes_per * n / (ms * 1e-3) / 1e9
    print("%-18s %8.3f ms  %9.3e /s  %7.0f GB/s algorithmic  (%.2f of peak)" % (name, ms, n / (ms * 1e-3), gbs, gbs / 8000))
