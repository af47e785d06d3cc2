"""MSM: a quad of lanes per point (k_msm_small) against the bucket method by batch size (tuning key msm_small_max), warm
clocks, both input forms.  Dev tool: profiles/r05_msm_small_route_sweep.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
nmax = 1 << 15
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
pm, _ = ctx.decompress(enc)
def t(pts, n, quads, reps):
    with ctx.tuning(msm_small_max=((1 << 24) if quads else 1100)):
        for _ in range(reps): ctx.msm(pts[:n], k[:n])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ctx.msm(pts[:n], k[:n])
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, pts in (("Elements", pm), ("Encodings", enc)):
    for n in [1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768]:
        a = [t(pts, n, True, 30), t(pts, n, False, 30), t(pts, n, True, 30), t(pts, n, False, 30)]
        print("%-9s n=%6d  quads %7.1f %7.1f   buckets %7.1f %7.1f   buckets/quads %.3f" % (name, n, a[0], a[2], a[1], a[3], (a[1] + a[3]) / (a[0] + a[2])), flush=True)
