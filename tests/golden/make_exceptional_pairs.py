#!/usr/bin/env python3
"""Regenerates tests/golden/hash_exceptional_pairs.json: input pairs (r1, r2) of hash_to_curve whose two Elligator images
(s1, t1), (s2, t2) on the Jacobi quartic satisfy s1 s2 = +-1 -- the exceptional case of the quartic's addition law, for
which the kernels take the reference's own route (Edwards addition, generic compression: curve.hpp
ge_dcb_from_jacobi_sum, d377.hip hash_exceptional_pair).  No random input takes that route, so the pairs are constructed
with the big-integer model (oracle/d377_model.py, pinned by reference_kats.json): for a random r1 with image s1, the
target s2 = +-1 / s1 (the non-negative one: the map returns a non-negative s exactly when its ratio was a square,
src/ark_curve/elligator.rs:25-45), then the map inverted in that case --
    s^2 = num / den,   num = (r + 1)(a - 2d),   den = (d r - (d - a))((d - a) r - d),   r = zeta r0^2
is a quadratic in r; a root r with r / zeta a square gives r0.  About one r1 in four has such a partner.
The file holds inputs and the model's expected encodings only.  Run from the repo root: python tests/golden/make_exceptional_pairs.py"""
import json, os, random, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import d377_model as m  # noqa: E402

Q, A, D, ZETA = m.Q, m.COEFF_A, m.COEFF_D, m.ZETA


def fsqrt(x):
    """a square root of x mod q, or None"""
    ok, r = m.sqrt_ratio_zeta(x % Q, 1)
    return r if ok else None


def jacobi_s(r0):
    """the s coordinate of the Elligator image of r0 (elligator.rs:20-45), and whether the ratio was a square"""
    r = ZETA * r0 % Q * r0 % Q
    den = (D * r - (D - A)) % Q * (((D - A) * r - D) % Q) % Q
    num = (r + 1) * (A - 2 * D) % Q
    iss, isri = m.sqrt_ratio_zeta(1, num * den % Q)
    if not iss:
        isri = isri * r0 % Q
    s = isri * num % Q
    if m.is_negative(s) == iss:
        s = (-s) % Q
    return s, iss


def partner(s2):
    """r0 whose image has the s coordinate s2 (non-negative), through the square case of the map; or None"""
    ss = s2 * s2 % Q
    qa = ss * D % Q * (D - A) % Q
    qb = (-(ss * ((D * D + (D - A) * (D - A)) % Q) + (A - 2 * D))) % Q
    qc = (qa - (A - 2 * D)) % Q
    disc = fsqrt((qb * qb - 4 * qa * qc) % Q)
    if disc is None or qa == 0:
        return None
    for sign in (1, -1):
        r = (-qb + sign * disc) * pow(2 * qa, -1, Q) % Q
        r0 = fsqrt(r * pow(ZETA, -1, Q) % Q)
        if r0 is None:
            continue
        for cand in (r0, (-r0) % Q):
            s, iss = jacobi_s(cand)
            if iss and s == s2:
                return cand
    return None


def main():
    rng = random.Random(377)
    pairs = []
    tries = 0
    while len(pairs) < 12:
        tries += 1
        r1b = bytes(rng.getrandbits(8) for _ in range(32))
        r1 = m.fq_from_le_bytes_mod_order(r1b)
        s1, _ = jacobi_s(r1)
        if s1 == 0:
            continue
        inv = pow(s1, -1, Q)
        s2 = inv if not m.is_negative(inv) else (-inv) % Q
        r2 = partner(s2)
        if r2 is None:
            continue
        s2b, _ = jacobi_s(r2)
        assert s1 * s2b % Q in (1, Q - 1)
        r2b = m.fq_to_bytes(r2)
        enc = m.compress(m.hash_to_curve(r1, r2))
        enc_swapped = m.compress(m.hash_to_curve(r2, r1))
        assert enc == enc_swapped
        pairs.append({"r1": r1b.hex(), "r2": r2b.hex(), "s1_s2": "+1" if s1 * s2b % Q == 1 else "-1", "encoding": bytes(enc).hex()})
    out = {"generator": "tests/golden/make_exceptional_pairs.py", "seed": 377, "tries": tries, "pairs": pairs}
    with open(os.path.join(ROOT, "tests", "golden", "hash_exceptional_pairs.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("%d exceptional pairs from %d candidates" % (len(pairs), tries))


if __name__ == "__main__":
    main()
