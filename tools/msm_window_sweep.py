#!/usr/bin/env python3
"""Whole-call time of d377_msm_dev on Elements for every window width the override allows, per batch size: where does
pick_window's rule stand?     python tools/msm_window_sweep.py [log2 sizes ...] [--widths 14,15,...]
-> table on stdout (profiles/rNN_msm_window_sweep.txt); every width's result is checked against the first one's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d


def main():
    args = sys.argv[1:]
    widths = [12, 13, 14, 15, 16, 17, 18]
    if "--widths" in args:
        i = args.index("--widths")
        widths = [int(x) for x in args[i + 1].split(",")]
        args = args[:i] + args[i + 2:]
    logs = [int(a) for a in args] or [18, 19, 20, 21, 22, 23, 24]
    ctx = d.Context([0], comb_lazy=True)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    nmax = 1 << max(logs)
    r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    print("one MI355X, Elements resident in HBM, us per call (best of 3 batches of reps); widths: " + " ".join("%7d" % w for w in widths) + "   | the built-in rule")
    for lg in logs:
        n = 1 << lg
        reps = 5 if lg <= 20 else (3 if lg <= 22 else 2)
        row, ref = [], None
        for w in widths + [None]:
            with ctx.tuning(msm_window=w, msm_small_max=0):
                enc = ctx.msm(P[:n], k[:n])[0]
                torch.cuda.synchronize()
                if ref is None:
                    ref = enc.clone()
                assert torch.equal(enc, ref), (lg, w)
                best = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        ctx.msm(P[:n], k[:n])
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / reps * 1e3
                    best = us if best is None or us < best else best
                row.append(best)
        print("  n=2^%-2d  " % lg + " ".join("%7.0f" % v for v in row[:-1]) + "   | %7.0f" % row[-1], flush=True)


if __name__ == "__main__":
    main()
