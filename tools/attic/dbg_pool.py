import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, decaf377_amd as d
dev = torch.device("cuda:0")
c = d.Context([0])
g = torch.Generator(device=dev).manual_seed(77)
n = 70000
r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
want = c.encode_to_curve(r0); torch.cuda.synchronize()
print("health0", c.health(), "reset0", c.reset_scratch())
c._debug_poison_pool()
print("poisoned", c.health())
out = torch.zeros_like(want)
t0 = time.time()
c.encode_to_curve(r0, outs=[out])
print("enqueued %.3f" % (time.time() - t0))
for i in range(40):
    t1 = time.time()
    h = c.health()
    print("t=%.3f health call %.3f s -> %s" % (time.time() - t0, time.time() - t1, h), flush=True)
    if h[2]: break
    time.sleep(0.25)
print("reset", c.reset_scratch(), "%.3f" % (time.time() - t0))
torch.cuda.synchronize()
print("done %.3f" % (time.time() - t0), c.health())
