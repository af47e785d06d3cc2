"""Chunked kernels between full generations (n not a multiple of 8 x the resident lanes): the rule against uniform chunks of
p elements per lane (tuning key chunk_per_lane), warm clocks.  Dev tool: profiles/r05_chunk_between_generations.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(3)
nmax = 3 << 20
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
o32 = torch.empty((nmax, 32), dtype=torch.uint8, device=dev); o1 = torch.empty((nmax,), dtype=torch.uint8, device=dev)
ops = {
    "sqrt_ratio_zeta": lambda n: ctx.sqrt_ratio_zeta(r0[:n], k[:n], outs=[o32[:n], o1[:n]]),
    "encode_to_curve": lambda n: ctx.encode_to_curve(r0[:n], outs=[o32[:n]]),
    "hash_to_curve": lambda n: ctx.hash_to_curve(r0[:n], k[:n], outs=[o32[:n]]),
    "scalar_mul_var": lambda n: ctx.scalar_mul_var(enc[:n], k[:n], outs=[o32[:n], o1[:n]]),
}
only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(ops)
def t(fn, n, pl, reps):
    with ctx.tuning(chunk_per_lane=pl):
        for _ in range(reps): fn(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn(n)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(3): ops["scalar_mul_var"](1 << 20)
for name in only:
    fn = ops[name]
    for n in [1 << 20, 5 << 18, 6 << 18, 7 << 18, 1 << 21, 5 << 19, 3 << 20]:
        reps = 3 if name == "scalar_mul_var" else 10
        row = ["rule %.0f" % t(fn, n, None, reps)]
        for pl in (4, 5, 6, 7, 8):
            row.append("%d: %.0f" % (pl, t(fn, n, pl, reps)))
        print("%-16s n=%8d (%.2f x 2^20) us: %s" % (name, n, n / (1 << 20), "   ".join(row)), flush=True)
