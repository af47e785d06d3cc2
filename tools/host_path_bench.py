#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (numpy buffers in pageable host memory).
Dev tool; the numbers are quoted in DESIGN.md section 5 (they are never bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import decaf377_amd as d

ctx = d.Context([0])
rng = np.random.default_rng(5)
for name, lg in (("scalar_mul_var", 22), ("roundtrip", 20), ("encode_to_curve", 20), ("scalar_mul_base", 20)):
    n = 1 << lg
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    enc = ctx.encode_to_curve(r0)
    fn = {"scalar_mul_var": lambda: ctx.scalar_mul_var(enc, k), "roundtrip": lambda: ctx.roundtrip(enc),
          "encode_to_curve": lambda: ctx.encode_to_curve(r0), "scalar_mul_base": lambda: ctx.scalar_mul_base(k)}[name]
    fn()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t0) / 3
    # the same with output arrays the caller allocated (and touched) once: a fresh np.empty of 32 MB is mmap'ed
    # anew on every call and page-faults inside the D2H copy
    o32, st = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
    fn2 = {"scalar_mul_var": lambda: ctx.scalar_mul_var(enc, k, outs=[o32, st]), "roundtrip": lambda: ctx.roundtrip(enc, outs=[o32, st]),
           "encode_to_curve": lambda: ctx.encode_to_curve(r0, outs=[o32]), "scalar_mul_base": lambda: ctx.scalar_mul_base(k, outs=[o32])}[name]
    fn2()
    t0 = time.perf_counter()
    for _ in range(3):
        fn2()
    dt2 = (time.perf_counter() - t0) / 3
    print("%-16s n=2^%d  host-pointer path %8.2f ms  %.3e /s   | reused output buffers %8.2f ms  %.3e /s" % (name, lg, dt * 1e3, n / dt, dt2 * 1e3, n / dt2))
