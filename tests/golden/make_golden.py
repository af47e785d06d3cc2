#!/usr/bin/env python3
"""Regenerates tests/golden/model_vectors.json from the big-integer model
(oracle/d377_model.py), which is itself pinned by reference_kats.json.

Run from the repo root:  python tests/golden/make_golden.py
Inputs follow SURVEY.md section 8d: raw 32-byte strings reduced mod q / mod r the way
tests/operations.rs:6-17 of the reference builds its proptest strategies, plus edge cases.
The file holds inputs and expected outputs only (hex); no reference source text.
"""
import json, os, random, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import d377_model as m  # noqa: E402

SEED = 666  # benches/sqrt.rs:28 uses ChaChaRng::seed_from_u64(666); we reuse the constant


def rb(rng):
    return bytes(rng.getrandbits(8) for _ in range(32))


def main():
    rng = random.Random(SEED)
    out = {"seed": SEED, "generator": "tests/golden/make_golden.py"}

    # sqrt_ratio_zeta: edge pairs then random (num, den)
    pairs = [(0, 1), (1, 0), (0, 0), (1, 1), (1 << 248, 1 << 248), (m.Q - 1, 1), (1, m.Q - 1), (m.ZETA, 1)]
    cases = []
    for u, v in pairs:
        cases.append((u.to_bytes(32, "little"), v.to_bytes(32, "little")))
    for _ in range(56):
        cases.append((rb(rng), rb(rng)))
    sq = []
    for nb, db in cases:
        ws, r = m.sqrt_ratio_zeta(m.fq_from_le_bytes_mod_order(nb), m.fq_from_le_bytes_mod_order(db))
        sq.append({"num": nb.hex(), "den": db.hex(), "was_square": int(ws), "root": m.fq_to_bytes(r).hex()})
    out["sqrt_ratio_zeta"] = sq

    # encode_to_curve -> encoding, with extended Montgomery coordinates of the image
    el = []
    r0s = [bytes(32), (1).to_bytes(32, "little"), (m.Q - 1).to_bytes(32, "little"), b"\xff" * 32]
    r0s += [rb(rng) for _ in range(44)]
    for b in r0s:
        p = m.encode_to_curve(m.fq_from_le_bytes_mod_order(b))
        el.append({"r0": b.hex(), "enc": m.compress(p).hex(),
                   "xyzt_mont": [m.to_mont_limbs(c) for c in p]})
    out["encode_to_curve"] = el

    # decompress of raw strings (mostly invalid) + valid encodings, with status and coordinates
    dec = []
    raws = [rb(rng) for _ in range(40)]
    raws += [bytes([b] + [0] * 31) for b in range(0, 20)]
    raws += [bytes.fromhex(e["enc"]) for e in el[:24]]
    raws += [(m.Q).to_bytes(32, "little"), (m.Q - 1).to_bytes(32, "little"), (m.Q + 2).to_bytes(32, "little"),
             bytes([0] * 31 + [0x20]), bytes([0] * 31 + [0x80]), b"\xff" * 32]
    for b in raws:
        p = m.decompress(b)
        if p is None:
            dec.append({"enc": b.hex(), "status": 1})
        else:
            dec.append({"enc": b.hex(), "status": 0, "xyzt_mont": [m.to_mont_limbs(c) for c in p],
                        "recompressed": m.compress(p).hex()})
    out["decompress"] = dec

    # scalar multiplication: fixed base and variable base
    r = m.R_ORDER
    ks = [bytes(32), (1).to_bytes(32, "little"), (2).to_bytes(32, "little"), (r - 1).to_bytes(32, "little"),
          r.to_bytes(32, "little"), (r + 1).to_bytes(32, "little"), b"\xff" * 32,
          (8).to_bytes(32, "little"), (1 << 250).to_bytes(32, "little"),
          int("8" * 63, 16).to_bytes(32, "little"), int("7" * 63, 16).to_bytes(32, "little"),
          int("f" * 62, 16).to_bytes(32, "little")]
    ks += [rb(rng) for _ in range(20)]
    fb = []
    for kb in ks:
        k = m.fr_from_le_bytes_mod_order(kb)
        fb.append({"scalar": kb.hex(), "enc": m.compress(m.scalar_mul(m.GENERATOR, k)).hex()})
    out["scalar_mul_base"] = fb
    vb = []
    pts = [bytes.fromhex(e["enc"]) for e in el[4:36]]
    pts[0] = bytes(32)                       # identity as the base point
    pts[1] = bytes([8] + [0] * 31)           # generator
    for i, pb in enumerate(pts):
        kb = ks[i % len(ks)]
        p = m.decompress(pb)
        k = m.fr_from_le_bytes_mod_order(kb)
        vb.append({"point": pb.hex(), "scalar": kb.hex(), "status": 0,
                   "enc": m.compress(m.scalar_mul(p, k)).hex()})
    for pb in raws[:6]:                      # invalid points -> status 1, zero output
        if m.decompress(pb) is None:
            vb.append({"point": pb.hex(), "scalar": ks[13].hex(), "status": 1, "enc": bytes(32).hex()})
    out["scalar_mul_var"] = vb

    # hash_to_curve (two maps + add)
    h2c = []
    for _ in range(12):
        a, b = rb(rng), rb(rng)
        p = m.hash_to_curve(m.fq_from_le_bytes_mod_order(a), m.fq_from_le_bytes_mod_order(b))
        h2c.append({"r1": a.hex(), "r2": b.hex(), "enc": m.compress(p).hex()})
    out["hash_to_curve"] = h2c

    # Fr reduction of raw bytes
    out["fr_mod_order"] = [{"bytes": kb.hex(), "reduced": m.fr_from_le_bytes_mod_order(kb).to_bytes(32, "little").hex()}
                           for kb in ks]

    path = os.path.join(ROOT, "tests", "golden", "model_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
