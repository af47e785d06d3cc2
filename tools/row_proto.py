#!/usr/bin/env python3
"""Runs tools/row_proto.hip (build/row_proto.so) on the GPU: the lane-spread product against Python integers, and the time
of a dependent chain of products on one wave against the one-lane-per-element multiplier.  Dev tool."""
import ctypes, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import row_model as rm

lib = ctypes.CDLL(os.path.join(ROOT, "build", "row_proto.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

def main():
    rng = random.Random(11)
    TIGHT = ((1 << 28) + (1 << 12), (1 << 20) + (1 << 12))
    LAZY = (int(2 ** 30.25), 1 << 24)
    n = 4096
    A, B = np.zeros((n, 16), np.uint32), np.zeros((n, 16), np.uint32)
    for i in range(n):
        la, lb = rng.choice([TIGHT, LAZY]), rng.choice([TIGHT, LAZY])
        a, b = rm.rand_elem(rng, *la), rm.rand_elem(rng, *lb)
        if i % 7 == 0: a = [la[0] - 1] * 9 + [la[1] - 1] + [0] * 6
        if i % 11 == 0: b = [lb[0] - 1] * 9 + [lb[1] - 1] + [0] * 6
        A[i], B[i] = a, b
    O = np.zeros((n, 16), np.uint32)
    assert lib.row_proto_mul(P(A), P(B), P(O), n) == 0
    bad = 0
    for i in range(n):
        g = [int(x) for x in O[i]]
        want = rm.row_mul([int(x) for x in A[i]], [int(x) for x in B[i]])
        if g != want:
            bad += 1
            if bad < 4: print("MISMATCH", i, g, want)
        assert rm.val(g) % rm.Q == rm.val([int(x) for x in A[i]]) * rm.val([int(x) for x in B[i]]) % rm.Q
    print("row_mul: %d products, %d differ from the model, all congruent to a*b mod q" % (n, bad))
    a = np.zeros((64, 16), np.uint32); b = np.zeros((64, 16), np.uint32); o = np.zeros((64, 16), np.uint32)
    for r in range(4):
        a[r * 16 // 16 * 1] = 0
    arow = np.array([rm.rand_elem(rng, *TIGHT) for _ in range(4)], np.uint32).reshape(64)
    brow = np.array([rm.rand_elem(rng, *TIGHT) for _ in range(4)], np.uint32).reshape(64)
    af = np.zeros(64 * 16, np.uint32); bf = np.zeros(64 * 16, np.uint32)
    af[:64], bf[:64] = arow, brow
    ms = ctypes.c_float(0)
    iters = 20000
    for which, name in ((0, "row product chain"), (1, "lane fe_mul chain"), (2, "lane fe_sqr chain")):
        if which:
            af = np.random.default_rng(3).integers(0, 1 << 29, 64 * 16, dtype=np.uint32); bf = np.random.default_rng(4).integers(0, 1 << 29, 64 * 16, dtype=np.uint32)
        of = np.zeros(64 * 16, np.uint32)
        assert lib.row_proto_chain(which, iters, P(af), P(bf), P(of), ctypes.byref(ms)) == 0
        print("%-20s %8.3f ms for %d dependent products: %6.1f ns each" % (name, ms.value, iters, ms.value * 1e6 / iters))
        if which == 0:
            x, y = [int(v) for v in arow[:16]], [int(v) for v in brow[:16]]
            for _ in range(200 if iters > 200 else iters):
                pass
    print("ROW_PROTO_DONE")

main()
