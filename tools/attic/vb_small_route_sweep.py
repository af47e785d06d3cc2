"""Variable base (Encodings and Elements) and fixed base: a quad of lanes per element (tuning key small_max) against a lane per
element by batch size, warm clocks.  Dev tool: profiles/r05_vb_small_route_sweep.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
nmax = 1 << 17
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
pm, _ = ctx.decompress(enc)
o32 = torch.empty((nmax, 32), dtype=torch.uint8, device=dev); o1 = torch.empty((nmax,), dtype=torch.uint8, device=dev)
ops = {
    "scalar_mul_var": lambda n: ctx.scalar_mul_var(enc[:n], k[:n], outs=[o32[:n], o1[:n]]),
    "scalar_mul_var_element": lambda n: ctx.scalar_mul_var_element(pm[:n], k[:n]),
    "scalar_mul_base": lambda n: ctx.scalar_mul_base(k[:n], outs=[o32[:n]]),
}
def t(fn, n, quads, reps):
    with ctx.tuning(small_max=((1 << 24) if quads else 1100)):
        for _ in range(reps): fn(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn(n)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, fn in ops.items():
    for n in [8192, 12288, 16384, 20480, 24576, 32768, 40960, 49152, 65536, 98304, 131072]:
        a = [t(fn, n, True, 20), t(fn, n, False, 20), t(fn, n, True, 20), t(fn, n, False, 20)]
        print("%-22s n=%6d  quads %7.1f %7.1f   lanes %7.1f %7.1f   quads/lanes %.3f" % (name, n, a[0], a[2], a[1], a[3], (a[0] + a[2]) / (a[1] + a[3])), flush=True)
