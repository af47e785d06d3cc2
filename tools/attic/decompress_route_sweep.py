"""decompress / compress / round trip: wide grid against the chunked form (tuning key decompress_chunked_min) by batch size,
warm clocks; the outputs of the two routes are compared too.  Dev tool: profiles/r05_decompress_route_sweep.txt.
usage: decompress_route_sweep.py [decompress,compress,roundtrip]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
nmax = 1 << 22
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
enc[13::4001] = torch.randint(0, 256, enc[13::4001].shape, dtype=torch.uint8, device=dev, generator=g)   # raw strings: mostly invalid
enc[17::5003] = 0                                                                                        # the identity
with ctx.tuning(decompress_chunked_min=1 << 24):
    pm, _ = ctx.decompress(enc)
ops = {"decompress": lambda n: ctx.decompress(enc[:n]), "compress": lambda n: (ctx.compress(pm[:n]),), "roundtrip": lambda n: ctx.roundtrip(enc[:n])}
only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(ops)
def t(fn, n, chunked, reps):
    with ctx.tuning(decompress_chunked_min=(1 if chunked else (1 << 24))):
        for _ in range(reps): fn(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn(n)
        e1.record(); torch.cuda.synchronize()
        out = fn(n)
    return e0.elapsed_time(e1) / reps * 1e3, out
for _ in range(20): ctx.decompress(enc)
for name in only:
    fn = ops[name]
    for n in [65536, 131072, 196608, 262144, 393216, 524288, 786432, 1048576, 1310720, 1572864, 2097152, 3145728, 4194304, 4194304 - 77]:
        reps = max(5, min(100, (1 << 23) // n))
        (a0, w0), (a1, c1), (a2, _), (a3, _) = t(fn, n, False, reps), t(fn, n, True, reps), t(fn, n, False, reps), t(fn, n, True, reps)
        same = all(torch.equal(x, y) for x, y in zip(w0, c1))
        print("%-10s n=%8d  wide grid %8.1f %8.1f   chunked %8.1f %8.1f   chunked/wide %.3f   outputs %s" % (name, n, a0, a2, a1, a3, (a1 + a3) / (a0 + a2), "equal" if same else "DIFFER"), flush=True)
