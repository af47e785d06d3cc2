#!/usr/bin/env python3
"""The wave's inversion (row_ops.hpp fe_invert_wave) on Python integers, step for step: variable-time divsteps on the low words
(the batched form, runs of zeros and up to eight bits of g cancelled at a time), the 2 x 2 matrix applied to lazily carried
signed 30-bit limbs with the kernel's two carry passes, no sign test, one lift by 32 q at the end.  Every intermediate is checked
against the width the kernel gives it (int64 products, uint32 sums, int32 limbs), d and e against the range the comments
claim, the exact divisions by 2^30 against the integers, and the result against pow(c, -1, q) -- on structured residues
(powers of two, q minus them, small numbers, their inverses: the values whose inverses are SMALL, which is where a sign test
on lazily carried limbs goes wrong) and random ones.  Prints INV_WAVE_MODEL_OK.  tools/attic/row_invert_trace.py compares the same
rounds with the GPU's."""
import random
import sys

Q = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
M30 = (1 << 30) - 1
ROUNDS = 25


def s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def ctz(x):
    return (x & -x).bit_length() - 1


def divsteps_30_var(eta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g, i = f0, g0, 30
    while True:
        zeros = ctz((g | (0xFFFFFFFF << i)) & 0xFFFFFFFF)
        g >>= zeros
        u = (u << zeros) & 0xFFFFFFFF
        v = (v << zeros) & 0xFFFFFFFF
        eta -= zeros
        i -= zeros
        if i == 0:
            break
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
        g = (g + f * w) & 0xFFFFFFFF
        q = (q + u * w) & 0xFFFFFFFF
        r = (r + v * w) & 0xFFFFFFFF
    u, v, q, r = s32(u), s32(v), s32(q), s32(r)
    assert abs(u) + abs(v) <= 1 << 30 and abs(q) + abs(r) <= 1 << 30
    return eta, u, v, q, r


def limbs(x):
    out = []
    for _ in range(8):
        out.append(x & M30)
        x >>= 30
    return out + [x]


def val(l):
    return sum(v << (30 * i) for i, v in enumerate(l))


QL = limbs(Q)


def apply(a, X, b, Y, mm):
    """one row of the round: lane j computes a X_j + b Y_j + mm Q_j and the two carry passes"""
    cs = [a * X[j] + b * Y[j] + mm * QL[j] for j in range(9)] + [0]
    assert all(-(1 << 63) <= c < (1 << 63) for c in cs), "a product sum leaves int64"
    lo = [c & M30 for c in cs]
    h = [c >> 30 for c in cs]
    assert lo[0] == 0, "the division by 2^30 is not exact"
    assert -(1 << 31) <= h[8] < (1 << 31), "the top limb leaves int32"
    mid = [(h[j] & M30) if j != 8 else h[j] for j in range(9)]
    tp = [(h[j] >> 30) if j != 8 else 0 for j in range(9)]
    t1 = [lo[j + 1] + mid[j] for j in range(9)]
    assert all(0 <= t1[j] < (1 << 32) for j in range(8))
    keep = [t1[j] & M30 if j != 8 else t1[j] for j in range(9)]
    up = [((t1[j] >> 30) + tp[j]) if j != 8 else 0 for j in range(9)]
    out = [keep[j] + (up[j - 1] if j > 0 else 0) for j in range(9)]
    assert all(-4 < o < (1 << 30) + 5 for o in out[:8]) and -(1 << 31) <= out[8] < (1 << 31), "a limb leaves its range"
    return out


def inv_wave(c):
    f, g, d, e = limbs(Q), limbs(c), limbs(0), limbs(1)
    eta = -1
    for it in range(ROUNDS):
        f0, g0 = f[0] & M30, g[0] & M30
        d0, e0 = d[0] & 0xFFFFFFFF, e[0] & 0xFFFFFFFF
        eta, u, v, q, r = divsteps_30_var(eta, f0, g0)
        md = -((u * d0 + v * e0) & M30)
        me = -((q * d0 + r * e0) & M30)
        nf, ng, nd, ne = apply(u, f, v, g, 0), apply(q, f, r, g, 0), apply(u, d, v, e, md), apply(q, d, r, e, me)
        assert val(nf) << 30 == u * val(f) + v * val(g) and val(ng) << 30 == q * val(f) + r * val(g)
        assert ((val(nd) << 30) - (u * val(d) + v * val(e))) % Q == 0 and ((val(ne) << 30) - (q * val(d) + r * val(e))) % Q == 0
        f, g, d, e = nf, ng, nd, ne
        assert -(it + 2) * Q < val(d) <= Q and -(it + 2) * Q < val(e) <= Q, "d or e leaves (-(round + 2) q, q]"
    assert val(g) == 0, "g has not reached zero in %d rounds" % ROUNDS
    fv = val(f)
    assert fv in (1, -1) or (c % Q == 0 and fv in (Q, -Q))
    negate = (f[0] & 2) != 0
    assert negate == (fv < 0)
    y, carry = [], 0
    for i in range(9):
        t = (-d[i] if negate else d[i]) + 32 * QL[i] + carry
        y.append(t & M30 if i < 8 else t)
        carry = t >> 30
    assert 0 <= y[8] < (1 << 19) and 0 < val(y) < 60 * Q
    return val(y) % Q


def main():
    R = 1 << 261
    rinv = pow(R, -1, Q)
    cs = [1 << k for k in range(0, 252)] + [Q - (1 << k) for k in range(0, 252)] + list(range(3, 256, 2)) + [(Q - 1) // 2, (Q + 1) // 2, Q - 1, Q - 2]
    cs += [c * rinv % Q for c in cs] + [pow(c, -1, Q) for c in cs]
    rng = random.Random(5)
    cs += [rng.randrange(1, Q) for _ in range(4000)] + [rng.randrange(1, 1 << rng.randrange(1, 253)) for _ in range(2000)]
    for c in cs:
        assert inv_wave(c) * c % Q == 1, hex(c)
    assert inv_wave(0) == 0
    print("wave inversion model: %d residues (structured, their inverses, random, random short), every width and range held" % len(cs))
    print("INV_WAVE_MODEL_OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
