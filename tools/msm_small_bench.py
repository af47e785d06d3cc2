#!/usr/bin/env python3
"""d377_batch_msm_small against the composition it replaces: n independent m-term sums as ONE call, and as m
scalar_mul_var_element batches, m - 1 add batches and a compress (all on device tensors, HIP events).
    python tools/msm_small_bench.py [log2 sizes ...]   ->  table on stdout (profiles/rNN_msm_small_bench.txt)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        best = us if best is None or us < best else best
    return best


def main():
    logs = [int(a) for a in sys.argv[1:]] or [8, 12, 16, 18, 20]
    ctx = d.Context([0], comb_lazy=True)                       # no fixed-base leg here: no comb
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    print("one MI355X, device tensors; us per call: d377_batch_msm_small on Elements | on Encodings | the composition on Elements "
          "(m x scalar_mul_var_element, m - 1 x add, compress) | composition / msm_small")
    for m in (2, 3, 8):
        for lg in logs:
            n = 1 << lg
            if n * m > (1 << 23):
                continue
            r0 = torch.randint(0, 256, (n * m, 32), dtype=torch.uint8, device=dev, generator=g)
            k = torch.randint(0, 256, (n * m, 32), dtype=torch.uint8, device=dev, generator=g)
            enc = ctx.encode_to_curve(r0)
            P, _ = ctx.decompress(enc)
            cols = [P[j::m].contiguous() for j in range(m)]
            ks = [k[j::m].contiguous() for j in range(m)]
            out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
            st = torch.empty((n * m,), dtype=torch.uint8, device=dev)

            def composed():
                acc = ctx.scalar_mul_var_element(cols[0], ks[0])
                for j in range(1, m):
                    acc = ctx.add(acc, ctx.scalar_mul_var_element(cols[j], ks[j]))
                return ctx.compress(acc)

            ref = composed()
            got = ctx.msm_small(P, k, m, outs=[out])
            got_e, _ = ctx.msm_small(enc, k, m, outs=[out.clone(), st])
            torch.cuda.synchronize()
            assert torch.equal(got, ref) and torch.equal(got_e, ref), (m, lg)
            reps = 20 if lg <= 12 else (5 if lg <= 16 else 2)
            t_el = timed(lambda: ctx.msm_small(P, k, m, outs=[out]), reps)
            t_en = timed(lambda: ctx.msm_small(enc, k, m, outs=[out, st]), reps)
            t_co = timed(composed, reps)
            print("  m=%d n=2^%-2d  %10.1f | %10.1f | %10.1f | x%.2f   (%.3e sums/s, %.3e terms/s)" % (
                m, lg, t_el, t_en, t_co, t_co / t_el, n / t_el * 1e6, n * m / t_el * 1e6), flush=True)


if __name__ == "__main__":
    main()
