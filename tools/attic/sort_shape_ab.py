#!/usr/bin/env python3
"""One build of the library (D377_LIB) against slice counts of the MSM's sort: ms per call at 2^20 / 2^22 / 2^24 Elements and the
result's encoding (equal across builds: the inputs are seeded).  Dev tool for A/Bs of -DD377_SORT_THREADS / -DD377_TILE_PER_THREAD.
(the two macros: tools/attic/r06_sort_shape_macros.patch)  usage: D377_LIB=build/variants/x.so python tools/attic/sort_shape_ab.py [slices,...]  (0 = the library's own rule)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d


def main():
    slices = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
    ctx = d.Context([0], comb_lazy=True)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(99)
    n = 1 << 24
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    del r0
    print("lib %s" % os.environ.get("D377_LIB", "product"), flush=True)
    for sl in slices:
        for lg in (20, 22, 24):
            m = 1 << lg
            with ctx.tuning(**({"msm_slices": sl} if sl else {})):
                enc = ctx.msm(P[:m], k[:m])[0]
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        ctx.msm(P[:m], k[:m])
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 3)
            print("  slices %4d  n=2^%d  %8.3f ms   %s" % (sl, lg, best, bytes(enc.cpu().numpy()).hex()[:16]), flush=True)


if __name__ == "__main__":
    main()
