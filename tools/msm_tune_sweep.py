#!/usr/bin/env python3
"""MSM (Elements) whole-call time under tuning overrides: python tools/msm_tune_sweep.py <log2n,...> key=v1,v2,... [key=...]
Every combination of the listed values; us per call (HIP events, 10 calls).  Dev tool."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d

def main():
    sizes = [int(x) for x in sys.argv[1].split(",")]
    keys = [a.split("=")[0] for a in sys.argv[2:]]
    vals = [[None if v == "d" else int(v) for v in a.split("=")[1].split(",")] for a in sys.argv[2:]]
    ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
    nmax = 1 << max(sizes)
    r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    for lg in sizes:
        n = 1 << lg
        ref = None
        for combo in itertools.product(*vals):
            with ctx.tuning(**dict(zip(keys, combo))):
                enc = ctx.msm(P[:n], k[:n])[0]
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ctx.msm(P[:n], k[:n])
                e1.record(); torch.cuda.synchronize()
            ref = ref or bytes(enc)
            assert bytes(enc) == ref
            print("n=2^%d %s  %8.1f us" % (lg, " ".join("%s=%s" % kv for kv in zip(keys, combo)), e0.elapsed_time(e1) * 100), flush=True)

main()
