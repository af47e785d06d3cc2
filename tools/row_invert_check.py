#!/usr/bin/env python3
"""GPU check of the wave's inversion (row_ops.hpp fe_invert_wave) against the lane's (fe_invert) and against x * (1/x) = 1 on
random, small and edge values, and the time of either on a lone wave.  Dev tool (needs build/row_proto.so)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "build", "row_proto.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
Q = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
R = 1 << 261

def limbs29(x):
    return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]

def main():
    rng = np.random.default_rng(11)
    n = 3072
    a = rng.integers(0, 1 << 29, (n, 9), dtype=np.uint32)
    # edge values: 0, R (the field's one), small residues, q - 1, q, q + 1 (lazy forms of 0 and 1), all-ones limbs
    edge = [0, R % Q, 1, 2, 3, Q - 1, Q, Q + 1, (1 << 253) - 1, (R * (Q - 1)) % Q]
    # the integer the divsteps see is the element's residue in the internal form, i.e. the limbs themselves mod q: structured
    # values (powers of two, q minus them, small odd numbers, (q - 1) / 2), and the same divided by R (random-looking)
    rinv = pow(R, -1, Q)
    cs = [1 << k for k in range(1, 252)] + [Q - (1 << k) for k in range(0, 252)] + list(range(3, 64, 2)) + [(Q - 1) // 2, (Q + 1) // 2]
    edge += [c % Q for c in cs] + [c % Q * rinv % Q for c in cs[::3]] + [pow(c, -1, Q) for c in cs[::2]]     # ... and their inverses
    for k, x in enumerate(edge):
        a[k] = limbs29(x)
    a[len(edge)] = [(1 << 29) - 1] * 8 + [(1 << 20) - 1]
    assert len(edge) + 1 < n
    out = np.zeros((n, 64), np.uint32)
    assert lib.row_proto_invert(P(a), P(out), n) == 0
    am = a.copy(); am[:, 8] &= 0xFFFFF
    differ = int((out[:, 0:9] != out[:, 9:18]).any(axis=1).sum())
    zero = np.array([sum(int(v) << (29 * i) for i, v in enumerate(row)) % Q == 0 for row in am])
    prod_ok = (out[:, 18:27] == out[:, 27:36]).all(axis=1)
    bad_prod = int((~prod_ok & ~zero).sum())
    zero_ok = bool((out[zero][:, 0:9] == 0).all())
    for i in np.nonzero((out[:, 0:9] != out[:, 9:18]).any(axis=1))[0][:60]:
        c = sum(int(v) << (29 * k) for k, v in enumerate(am[i])) % Q
        print("  differs: element %d, residue %#x (bit length %d)" % (i, c, c.bit_length()))
    print("wave inversion: %d elements (%d zero): %d differ from the lane's, %d with x * (1/x) != 1, zero -> zero %s"
          % (n, int(zero.sum()), differ, bad_prod, zero_ok))
    ms = ctypes.c_float(0)
    o = np.zeros(64 * 9, np.uint32)
    iters = 200
    for which, name in ((0, "wave"), (1, "lane")):
        assert lib.row_proto_invert_chain(which, iters, P(a[100]), P(o), ctypes.byref(ms)) == 0
        print("  %s: %.2f us per inversion (chain of %d on one wave)" % (name, ms.value * 1000 / iters, iters))
    if differ == 0 and bad_prod == 0 and zero_ok:
        print("ROW_INVERT_OK")
        return 0
    return 1

if __name__ == "__main__":
    sys.exit(main())
