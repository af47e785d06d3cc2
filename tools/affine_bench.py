import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, decaf377_amd as d
ctx = d.Context([0]); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for lg in (16, 20, 22):
    n = 1 << lg
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, _ = ctx.decompress(ctx.encode_to_curve(r0))
    out = torch.empty((n, 8), dtype=torch.int64, device=dev)
    ctx.to_affine(P, outs=[out]); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ctx.to_affine(P, outs=[out])
    b.record(); torch.cuda.synchronize()
    print("blocks/CU", os.environ.get("D377_AFFINE_BLOCKS_PER_CU"), "n=2^%d" % lg, "%.3f ms" % (a.elapsed_time(b) / 5))
