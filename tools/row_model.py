#!/usr/bin/env python3
"""Python model of row_ops.hpp (the lane-spread Fq product: 10 x 28-bit limbs over the lanes of a 16-lane row).
Dev tool / test infrastructure: (1) random products through the same steps as the kernel, checked against a*b mod q;
(2) the worst case of every 64-bit accumulator for operands at their contract bounds (tight / lazy).
    python tools/row_model.py
"""
import random
import sys

Q = 725501752471715841 | 6461107452199829505 << 64 | 6968279316240510977 << 128 | 1345280370688173398 << 192
RW, RL = 28, 10
M28 = (1 << RW) - 1
FOLD = [[(pow(2, RW * (RL + m), Q) >> (RW * j)) & M28 for j in range(RL)] + [0] * 6 for m in range(11)]
F272 = [(pow(2, 272, Q) >> (RW * j)) & M28 for j in range(RL)] + [0] * 6
KEEP = [M28] * 9 + [0xFFFFFFFF] + [0] * 6
KEEP2 = [0xFFFFFFFF] * 9 + [(1 << 20) - 1] + [0] * 6
CMASK = [0xFFFFFFFF] * 9 + [0] * 7
VALID = [0xFFFFFFFF] * 10 + [0] * 6
U64 = (1 << 64) - 1
worst = {}


def note(name, vals, limit):
    m = max(vals)
    worst[name] = max(worst.get(name, 0), m)
    assert m < limit, (name, m.bit_length())


def shr(v, n):
    return [v[j - n] if j - n >= 0 else 0 for j in range(16)]


def shl(v, n):
    return [v[j + n] if j + n < 16 else 0 for j in range(16)]


def split3(x):
    lo = [t & M28 for t in x]
    mid = [(t >> RW) & M28 for t in x]
    top = [t >> (2 * RW) for t in x]
    note("split3 top", top, 1 << 32)
    return [lo[j] + shr(mid, 1)[j] + shr(top, 2)[j] for j in range(16)]


def row_mul(a, b):
    # lane j of L takes column j for all 16 lanes; H (lanes 6..8) takes columns 16..18 from the steps i = 7, 8, 9
    L, H = [0] * 16, [0] * 16
    for i in range(RL):
        bl = shr(b, i)
        for j in range(16):
            L[j] += a[i] * bl[j]
    for i in (7, 8, 9):
        bh = shl(b, RL - i)
        for j in range(16):
            H[j] += a[i] * bh[j]
    note("L", L, 1 << 64)
    note("H", H, 1 << 64)
    U = [shl(L, 10)[j] if j < 6 else H[j] for j in range(16)]   # columns 10..18 as lanes 0..8
    L = [L[j] if j < RL else 0 for j in range(16)]
    H = U
    h = split3(H)
    assert all(h[j] == 0 for j in range(11, 16))
    for m in range(11):
        for j in range(16):
            L[j] += h[m] * FOLD[m][j]
    note("L + fold", L, 1 << 64)
    n = split3(L)
    assert all(n[j] == 0 for j in range(12, 16))
    R = [(n[j] & VALID[j]) + n[10] * FOLD[0][j] + n[11] * FOLD[1][j] for j in range(16)]
    note("fold 2", R, 1 << 64)
    c = [(t >> RW) & CMASK[j] for j, t in enumerate(R)]
    note("carry A", c, 1 << 32)
    f = [(R[j] & KEEP[j]) + shr(c, 1)[j] for j in range(16)]
    note("pass A", f, 1 << 32)
    t = f[9] >> 20
    R3 = [(f[j] & KEEP2[j]) + t * F272[j] for j in range(16)]
    note("fold 3", R3, 1 << 64)
    c = [(x >> RW) & CMASK[j] for j, x in enumerate(R3)]
    note("carry B", c, 1 << 32)
    g = [(R3[j] & KEEP[j]) + shr(c, 1)[j] for j in range(16)]
    note("out limbs 0..8", g[:9], (1 << 28) + (1 << 12))
    note("out limb 9", g[9:10], (1 << 20) + (1 << 12))
    assert all(x == 0 for x in g[10:])
    return g


def val(v):
    return sum(x << (RW * j) for j, x in enumerate(v[:RL]))


def rand_elem(rng, lim_low, lim_top):
    return [rng.randrange(lim_low) for _ in range(9)] + [rng.randrange(lim_top)] + [0] * 6


def main():
    rng = random.Random(5)
    TIGHT = ((1 << 28) + (1 << 12), (1 << 20) + (1 << 12))
    LAZY = (int(2 ** 30.25), 1 << 24)
    for it in range(3000):
        la, lb = rng.choice([TIGHT, LAZY]), rng.choice([TIGHT, LAZY])
        a, b = rand_elem(rng, *la), rand_elem(rng, *lb)
        if it % 7 == 0:
            a = [la[0] - 1] * 9 + [la[1] - 1] + [0] * 6
        if it % 11 == 0:
            b = [lb[0] - 1] * 9 + [lb[1] - 1] + [0] * 6
        g = row_mul(a, b)
        assert val(g) % Q == val(a) * val(b) % Q, it
    # worst case: every limb at the lazy bound
    a = [LAZY[0] - 1] * 9 + [LAZY[1] - 1] + [0] * 6
    row_mul(a, a)
    for k, v in worst.items():
        print("%-16s max 2^%.3f" % (k, __import__("math").log2(v) if v else 0))
    print("ROW_MODEL_OK")


if __name__ == "__main__":
    main()
