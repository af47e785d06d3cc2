import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for lg in (16, 20):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    k1 = torch.randint(0, 256, (1, 32), dtype=torch.uint8, device=dev, generator=g)
    for name, k in (("random", torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)), ("all equal", k1.expand(n, 32).contiguous())):
        ctx.msm(P, k); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.msm(P, k); torch.cuda.synchronize()
        print("n=2^%d %-10s %9.3f ms" % (lg, name, (time.perf_counter() - t0) * 1e3))
