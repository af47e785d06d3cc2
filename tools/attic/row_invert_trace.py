#!/usr/bin/env python3
"""The rounds of one wave inversion on the GPU (row_ops.hpp fe_invert_wave_impl<true>) against the same rounds on Python
integers: prints the first round whose state differs.  Dev tool (needs build/row_proto.so).  usage: row_invert_trace.py [c]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "build", "row_proto.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
Q = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
R = 1 << 261
M30 = (1 << 30) - 1

def s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x
def ctz(x): return (x & -x).bit_length() - 1
def divsteps_var(eta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g, i = f0, g0, 30
    while True:
        zeros = ctz((g | (0xFFFFFFFF << i)) & 0xFFFFFFFF)
        g >>= zeros; u = (u << zeros) & 0xFFFFFFFF; v = (v << zeros) & 0xFFFFFFFF; eta -= zeros; i -= zeros
        if i == 0: break
        if eta < 0:
            eta = -eta
            f, g = g, (-f) & 0xFFFFFFFF
            u, q = q, (-u) & 0xFFFFFFFF
            v, r = r, (-v) & 0xFFFFFFFF
        limit = min(eta + 1, i)
        mask = (0xFFFFFFFF >> (32 - limit)) & 255
        fi = f
        fi = (fi * ((2 - f * fi) & 0xFFFFFFFF)) & 0xFFFFFFFF
        fi = (fi * ((2 - f * fi) & 0xFFFFFFFF)) & 0xFFFFFFFF
        w = ((-(g * fi)) & 0xFFFFFFFF) & mask
        g = (g + f * w) & 0xFFFFFFFF; q = (q + u * w) & 0xFFFFFFFF; r = (r + v * w) & 0xFFFFFFFF
    return eta, s32(u), s32(v), s32(q), s32(r)
def limbs(x):
    out = []
    for i in range(8):
        out.append(x & M30); x >>= 30
    return out + [x]

def main():
    # the integer the divsteps see is the element's residue in the internal (Montgomery) form: the limbs themselves, mod q
    c = int(sys.argv[1], 0) if len(sys.argv) > 1 else Q - (1 << 54)
    assert 0 <= c < min(Q, 1 << 252)
    a = np.array([(c >> (29 * i)) & ((1 << 29) - 1) for i in range(9)], np.uint32)
    out = np.zeros(25 * 80 + 16, np.uint32)
    assert lib.row_proto_invert_trace(P(a), P(out)) == 0
    QL = limbs(Q)
    st = [limbs(Q), limbs(c), limbs(0), limbs(1)]
    eta = -1
    for it in range(25):
        f, g, d, e = st
        f0, g0 = f[0] & M30, g[0] & M30
        d0, e0 = d[0] & 0xFFFFFFFF, e[0] & 0xFFFFFFFF
        eta, u, v, q, r = divsteps_var(eta, f0, g0)
        md = -((u * d0 + v * e0) & M30); me = -((q * d0 + r * e0) & M30)
        def upd(a_, X, b_, Y, mm):
            cs = [a_ * X[j] + b_ * Y[j] + mm * QL[j] for j in range(9)] + [0]
            lo = [cc & M30 for cc in cs]; h = [cc >> 30 for cc in cs]
            mid = [(h[j] & M30) if j != 8 else h[j] for j in range(9)]
            tp = [(h[j] >> 30) if j != 8 else 0 for j in range(9)]
            t1 = [lo[j + 1] + mid[j] for j in range(9)]
            keep = [t1[j] & M30 if j != 8 else t1[j] for j in range(9)]
            up = [((t1[j] >> 30) + tp[j]) if j != 8 else 0 for j in range(9)]
            return [keep[j] + (up[j - 1] if j > 0 else 0) for j in range(9)]
        st = [upd(u, f, v, g, 0), upd(q, f, r, g, 0), upd(u, d, v, e, md), upd(q, d, r, e, me)]
        gpu = [[s32(int(out[it * 80 + 16 * rr + j])) for j in range(9)] for rr in range(4)]
        gs = [s32(int(out[it * 80 + 64 + i])) for i in range(9)]
        want_s = [eta, u, v, q, r, md, me, f0, g0]
        if gpu != st or gs[:7] != want_s[:7]:
            print("round %d differs" % it)
            print("  scalars gpu ", gs)
            print("  scalars want", want_s)
            for rr, name in enumerate("fgde"):
                if gpu[rr] != st[rr]:
                    print("  %s gpu  %s" % (name, gpu[rr])); print("  %s want %s" % (name, st[rr]))
            return 1
    print("all 25 rounds equal to the model's; d = %.2f q at the end" % (sum(v << (30 * i) for i, v in enumerate(st[2])) / Q))
    return 0

if __name__ == "__main__":
    sys.exit(main())
