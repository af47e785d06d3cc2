#!/usr/bin/env python3
"""GPU check of the row-form group operations (row_ops.hpp rq_double_neg / rq_add) against the same formulas on Python
integers, and the time of a chain of doublings on rows against the chain on quads.  Dev tool (needs build/row_proto.so)."""
import ctypes, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import row_model as rm
Q = rm.Q
lib = ctypes.CDLL(os.path.join(ROOT, "build", "row_proto.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

def rows(vals):
    o = np.zeros(64, np.uint32)
    for r, x in enumerate(vals):
        for j in range(10):
            o[16 * r + j] = (x >> (28 * j)) & rm.M28
    return o

def unrows(o):
    return [sum(int(o[16 * r + j]) << (28 * j) for j in range(10)) % Q for r in range(4)]

def dbl(X, Y, Z, T):
    A, B, ZZ, TZ = X * X % Q, Y * Y % Q, Z * Z % Q, T * Z % Q
    G, H = (A - B) % Q, (A + B) % Q
    F, E = (G + 2 * ZZ) % Q, 2 * TZ % Q
    return [E * F % Q, G * H % Q, F * G % Q, E * H % Q]

def add(X, Y, Z, T, q, neg):
    q0, q1, q2, q3 = q
    if neg:
        q0, q1 = q1, q0
    a, b, c, d = (Y - X) * q0 % Q, (Y + X) * q1 % Q, T * q2 % Q, 2 * Z * q3 % Q
    E, H = (b - a) % Q, (b + a) % Q
    F, G = ((d + c) % Q, (d - c) % Q) if neg else ((d - c) % Q, (d + c) % Q)
    return [E * F % Q, G * H % Q, F * G % Q, E * H % Q]

def main():
    rng = random.Random(3)
    bad = 0
    for it in range(200):
        v = [rng.randrange(Q) for _ in range(4)]
        q = [rng.randrange(Q) for _ in range(4)]
        if it == 0: v = [Q - 1] * 4; q = [Q - 1] * 4
        if it == 1: v = [0, 1, 1, 0]
        out = np.zeros(256, np.uint32)
        assert lib.row_proto_point(P(rows(v)), P(rows(q)), P(out)) == 0
        got = [unrows(out[64 * k:64 * k + 64]) for k in range(4)]
        want = [dbl(*v), add(*v, q, False), add(*v, q, True), dbl(*dbl(*v))]
        for k in range(4):
            if got[k] != want[k]:
                bad += 1
                if bad < 5: print("MISMATCH it", it, "op", k)
        assert all(int(out[64 * k + 16 * r + j]) == 0 for k in range(4) for r in range(4) for j in range(10, 16))
        assert all(int(out[64 * k + 16 * r + j]) < (1 << 28) + (1 << 12) for k in range(4) for r in range(4) for j in range(9))
    print("row point ops: 200 random inputs x (double, add, sub, double twice): %d mismatches" % bad)
    ms = ctypes.c_float(0)
    iters = 5000
    a = np.random.default_rng(3).integers(0, 1 << 28, 64 * 16, dtype=np.uint32)
    a.reshape(64, 16)[:, 10:] = 0
    av = rows([rng.randrange(Q) for _ in range(4)])
    af = np.zeros(64 * 16, np.uint32); af[:64] = av
    o = np.zeros(64 * 16, np.uint32)
    for which, name, src in ((0, "rows", af), (1, "quads", np.random.default_rng(5).integers(0, 1 << 29, 64 * 16, dtype=np.uint32))):
        assert lib.row_proto_dbl_chain(which, iters, P(src), P(o), ctypes.byref(ms)) == 0
        print("doubling chain on %-6s %8.3f ms for %d doublings: %7.1f ns each" % (name, ms.value, iters, ms.value * 1e6 / iters))
    print("ROW_POINT_%s" % ("OK" if bad == 0 else "FAILED"))

main()
