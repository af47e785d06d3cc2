"""MSM on Encodings: the decoding pass on the wide grid against its chunked form (tuning key msm_enc_chunked_min) by
batch size, warm clocks.  Dev tool: profiles/r05_decompress_route_sweep.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(5)
nmax = 1 << 22
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
def t(n, chunked, reps):
    with ctx.tuning(msm_enc_chunked_min=(1 if chunked else (1 << 24))):
        for _ in range(reps): ctx.msm(enc[:n], k[:n])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ctx.msm(enc[:n], k[:n])
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(10): ctx.msm(enc, k)
for n in [32768, 65536, 131072, 196608, 262144, 393216, 524288, 786432, 1048576, 1310720, 1572864, 2097152, 3145728, 4194304]:
    reps = max(5, min(50, (1 << 23) // n))
    a = [t(n, False, reps), t(n, True, reps), t(n, False, reps), t(n, True, reps)]
    print("n=%8d  wide grid %8.1f %8.1f   chunked %8.1f %8.1f   chunked/wide %.3f" % (n, a[0], a[2], a[1], a[3], (a[1] + a[3]) / (a[0] + a[2])), flush=True)
