"""Per-workgroup timeline of a chunked kernel (developer variant: tools/build_variant.sh wgtimes -DD377_WG_TIMES, run with
D377_LIB=build/variants/wgtimes.so).  For each operation and size: the spread of the workgroups' start and end times and of
their durations, by XCD -- is a one-generation launch as long as its slowest workgroup, and what makes that one slow?
usage: D377_LIB=build/variants/wgtimes.so python tools/wg_times.py [op,op,...] [log2n,...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import decaf377_amd as d
from decaf377_amd import _native
lib = _native.load()
fn = lib.d377_debug_wg_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
ctx = d.Context([0]); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(7)
ops = (sys.argv[1] if len(sys.argv) > 1 else "sqrt_ratio_zeta,encode_to_curve,scalar_mul_var").split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "20,22").split(",")]
nmax = 1 << max(sizes)
r0 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
r1 = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device=dev, generator=g)
enc = ctx.encode_to_curve(r0)
calls = {"sqrt_ratio_zeta": lambda n: ctx.sqrt_ratio_zeta(r0[:n], r1[:n]), "encode_to_curve": lambda n: ctx.encode_to_curve(r0[:n]),
         "hash_to_curve": lambda n: ctx.hash_to_curve(r0[:n], r1[:n]), "scalar_mul_var": lambda n: ctx.scalar_mul_var(enc[:n], r1[:n]),
         "scalar_mul_base": lambda n: ctx.scalar_mul_base(r1[:n]), "decompress": lambda n: ctx.decompress(enc[:n])}
def pct(a, q): return float(np.percentile(a, q))
for op in ops:
    for lg in sizes:
        n = 1 << lg
        for _ in range(6): calls[op](n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); calls[op](n); e1.record(); torch.cuda.synchronize()
        wgs = min(16384, (n + 2047) // 2048)
        buf = np.zeros((wgs, 6), dtype=np.uint64)
        assert fn(buf.ctypes.data, wgs) == 0
        t0, t1, t2 = [buf[:, k].astype(np.int64) for k in range(3)]
        xcc = (buf[:, 3] >> np.uint64(32)).astype(np.int64) & 15
        hw = buf[:, 3].astype(np.int64) & 0xffffffff
        cu = (hw >> 8) & 15; se = (hw >> 13) & 7
        base = t0.min()
        us = lambda x: x / 100.0
        dur = us(t2 - t1)
        print("%s n=2^%d  kernel %.1f us (events)  workgroups %d  first start -> last end %.1f us" % (op, lg, e0.elapsed_time(e1) * 1e3, wgs, us(t2.max() - base)))
        print("   start after first: p50 %.1f  p99 %.1f  max %.1f us | claim p50 %.2f max %.2f us | duration min %.1f p1 %.1f p50 %.1f p99 %.1f max %.1f us (max/p50 %.3f)"
              % (pct(us(t0 - base), 50), pct(us(t0 - base), 99), us(t0 - base).max(), pct(us(t1 - t0), 50), us(t1 - t0).max(),
                 dur.min(), pct(dur, 1), pct(dur, 50), pct(dur, 99), dur.max(), dur.max() / pct(dur, 50)))
        print("   end before last:   p50 %.1f  p1 %.1f  earliest %.1f us" % (pct(us(t2.max() - t2), 50), pct(us(t2.max() - t2), 99), us(t2.max() - t2).max()))
        # by XCD: mean duration of its workgroups, and the end of its last one
        line = []
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            mhz = ((buf[m, 5].astype(np.int64) - buf[m, 4].astype(np.int64)) / np.maximum(1, (t2 - t1)[m]) * 100.0).mean()
            line.append("xcd%d: %d wg, mean %.1f, last end %.1f, clock64/wall %.0f MHz" % (x, int(m.sum()), dur[m].mean(), us(t2[m].max() - base), mhz))
        print("   " + " | ".join(line))
        # generations: how many workgroups started within 2 % of the kernel's start, and order of durations by start time
        first_gen = us(t0 - base) < 0.02 * us(t2.max() - base)
        print("   started in the first 2 %% of the kernel: %d; their mean duration %.1f us, the others' %.1f us" %
              (int(first_gen.sum()), dur[first_gen].mean(), dur[~first_gen].mean() if (~first_gen).any() else float("nan")), flush=True)
