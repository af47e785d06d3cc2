#!/usr/bin/env python3
"""Rate vs batch size for every operation of the path, 2^8 ... 2^22 records resident in HBM (device-pointer entry
points on torch's current stream), plus the hipGraph replay of the same call for n <= 2^12 (launch-bound sizes).
Real callers of the reference hold 10^2 - 10^5 elements (src/ark_curve/element/projective.rs:99-117, encoding.rs:32-128).

    python tools/size_sweep.py [--max 22] [--ops a,b,c]  ->  table on stdout (committed as profiles/rNN_size_sweep.txt)

Columns: us per call (HIP events around `reps` back-to-back calls), elements/s, and that rate relative to the op's
rate at 2^20; `graph` = the same for a captured graph of one call replayed `reps` times."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decaf377_amd as d


def timed(fn, reps):
    # the better of two batches: a host-side stall inside a batch of 100-us calls (seen: +40 us per call on the first small
    # size after a 2^22 batch, while the graph replay of the same call showed the kernel's time) is not the operation's time
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=8)
    ap.add_argument("--max", type=int, default=22)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--ops", type=str, default="")
    ap.add_argument("--sizes", type=str, default="", help="explicit sizes (records), comma separated, instead of powers of two")
    ap.add_argument("--tune", type=str, default="", help="developer overrides in force for every call: key=value,key=value (engine.TUNE_KEYS)")
    args = ap.parse_args()
    ctx = d.Context([0])
    for kv in filter(None, args.tune.split(",")):
        ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    nmax = 1 << args.max
    sizes = [int(x) for x in args.sizes.split(",")] if args.sizes else [1 << l for l in range(args.min, args.max + 1, args.step)]
    if not args.sizes and (1 << 20) not in sizes and args.max >= 20:
        sizes = sorted(set(sizes + [1 << 20]))
    nmax = max(sizes + [nmax if not args.sizes else 0])
    r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
    enc = ctx.encode_to_curve(r0)
    P, _ = ctx.decompress(enc)
    o32 = torch.empty((nmax, 32), dtype=torch.uint8, device=dev)
    o1 = torch.empty((nmax,), dtype=torch.uint8, device=dev)
    oE = torch.empty((nmax, 16), dtype=torch.int64, device=dev)
    oA = torch.empty((nmax, 8), dtype=torch.int64, device=dev)
    ops = {
        "sqrt_ratio_zeta": lambda n: ctx.sqrt_ratio_zeta(r0[:n], k[:n], outs=[o32[:n], o1[:n]]),
        "decompress": lambda n: ctx.decompress(enc[:n], outs=[oE[:n], o1[:n]]),
        "compress": lambda n: ctx.compress(P[:n], outs=[o32[:n]]),
        "roundtrip": lambda n: ctx.roundtrip(enc[:n], outs=[o32[:n], o1[:n]]),
        "encode_to_curve": lambda n: ctx.encode_to_curve(r0[:n], outs=[o32[:n]]),
        "hash_to_curve": lambda n: ctx.hash_to_curve(r0[:n], k[:n], outs=[o32[:n]]),
        "scalar_mul_base": lambda n: ctx.scalar_mul_base(k[:n], outs=[o32[:n]]),
        "scalar_mul_var": lambda n: ctx.scalar_mul_var(enc[:n], k[:n], outs=[o32[:n], o1[:n]]),
        "scalar_mul_var_element": lambda n: ctx.scalar_mul_var_element(P[:n], k[:n], outs=[oE[:n]]),
        "to_affine": lambda n: ctx.to_affine(P[:n], outs=[oA[:n]]),
        "add": lambda n: ctx.add(P[:n], P[:n], outs=[oE[:n]]),
        # n TERMS in n // m independent sums of m (d377_batch_msm_small): the rate is terms per second
        "msm_small (3 terms)": lambda n: ctx.msm_small(P[:3 * (n // 3)], k[:3 * (n // 3)], 3, outs=[o32[:n // 3]]),
        "msm_small (8 terms)": lambda n: ctx.msm_small(P[:8 * (n // 8)], k[:8 * (n // 8)], 8, outs=[o32[:n // 8]]),
        "msm (Elements)": lambda n: ctx.msm(P[:n], k[:n]),
        "msm (Encodings)": lambda n: ctx.msm(enc[:n], k[:n]),
    }
    if args.ops:
        ops = {o: ops[o] for o in args.ops.split(",")}
    print("one MI355X, records resident in HBM, device-pointer entry points; us per call | elements/s | rate relative to 2^20")
    for name, fn in ops.items():
        rows = []
        for n in sizes:
            reps = 20 if n <= (1 << 16) else (5 if n <= (1 << 20) else 3)
            us = timed(lambda: fn(n), reps)
            gus = None
            if n <= (1 << 12):
                fn(n)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    fn(n)
                gus = timed(gr.replay, reps)
                del gr
            rows.append((n, us, gus))
        ref = [n / us for n, us, _ in rows if n == (1 << 20)]
        ref = ref[0] if ref else max(n / us for n, us, _ in rows)
        print("\n%s" % name)
        for n, us, gus in rows:
            lg = n.bit_length() - 1
            tag = "2^%-2d" % lg if n == 1 << lg else "%d" % n
            line = "  n=%-8s %10.1f us  %10.3e /s  %5.2f" % (tag, us, n / us * 1e6, (n / us) / ref)
            if gus is not None:
                line += "   graph: %8.1f us  %10.3e /s" % (gus, n / gus * 1e6)
            print(line, flush=True)


if __name__ == "__main__":
    main()
