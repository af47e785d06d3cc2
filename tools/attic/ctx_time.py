import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, decaf377_amd as d
torch.cuda.init(); torch.zeros(1, device="cuda")
for i in range(3):
    t0 = time.time(); c = d.Context([0]); torch.cuda.synchronize(); t1 = time.time(); c.close()
    print("ctx create %.1f ms" % ((t1 - t0) * 1e3))
