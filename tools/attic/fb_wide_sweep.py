"""Fixed base: narrow against wide launch (tuning key fb_wide) by batch size, warm clocks.  Dev tool: profiles/r05_fb_wide_sweep.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
nmax = 1 << 22
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
o = torch.empty((nmax, 32), dtype=torch.uint8, device=dev)
def t(n, wide, reps):
    with ctx.tuning(fb_wide=wide):
        for _ in range(reps): ctx.scalar_mul_base(k[:n], outs=[o[:n]])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ctx.scalar_mul_base(k[:n], outs=[o[:n]])
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# keep the clocks up
for _ in range(50): ctx.scalar_mul_base(k, outs=[o])
for n in [65536, 98304, 131072, 196608, 262144, 393216, 524288, 655360, 786432, 917504, 1048576, 1310720, 1572864, 1835008, 2097152, 3145728, 4194304]:
    reps = max(10, min(200, (1 << 24) // n))
    a = [t(n, 0, reps), t(n, 1, reps), t(n, 0, reps), t(n, 1, reps)]
    print("n=%8d  narrow %8.1f %8.1f   wide %8.1f %8.1f   wide/narrow %.3f" % (n, a[0], a[2], a[1], a[3], (a[1] + a[3]) / (a[0] + a[2])), flush=True)
