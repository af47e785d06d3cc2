#!/usr/bin/env python3
"""Randomised cross-check of the size-dependent routes: every operation that picks a kernel or a launch shape by batch size
is run at random sizes around its thresholds on both routes (d377_ctx_set_tuning forces the other one) and the bytes are
compared; a sample of each result also goes to the oracle.  Test infrastructure (the oracle is the checker).
  variable base (Encodings / Elements):  one wave / one quad per element |  one lane per element      tiny_max, small_max
  MSM (Elements / Encodings):            one wave / one quad per point  |  Pippenger                  msm_tiny_max, msm_small_max
  fixed base:                            2 workgroups per CU, K = 8     |  3 per CU, K = 16           fb_wide / fb_k
                                         one wave per scalar            |  one lane per scalar        tiny_max
  sqrt_ratio_zeta, decompress, compress, round trip, encode_to_curve, hash_to_curve:
                                         four elements per wave         |  one lane per element       tiny_max
  small sums (msm_small), 1..8 terms:    one wave per sum               |  one lane per sum           tiny_max
usage: python tools/route_stress.py [rounds=40] [seed=1]   -> summary lines, exit 1 on any mismatch"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import decaf377_amd as d  # noqa: E402
from _oracle import Oracle  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    orc = Oracle(native=True)
    ctx = d.Context([0])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda n: torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    one_gen = cus * 64
    bad = 0
    nmax = 5 * one_gen
    r0, k = rnd(nmax), rnd(nmax)
    enc_all = ctx.encode_to_curve(r0)
    enc_all[3::97, 31] |= 0x80                                     # invalid encodings sprinkled in
    valid_all = ctx.encode_to_curve(rnd(nmax))
    P_all, _ = ctx.decompress(valid_all)

    def sizes(lo, hi, marks):
        s = [int(v) for v in rng.integers(lo, hi, rounds)]
        for m in marks:
            s += [m - 1, m, m + 1]
        return sorted(set(v for v in s if v >= 1))

    # variable base, both forms
    cnt = 0
    for n in sizes(1, 3 * one_gen, [16, one_gen, 7 * cus * 16, 8 * cus * 16]):
        with ctx.tuning(small_max=1 << 24, tiny_max=0):
            q = ctx.scalar_mul_var(enc_all[:n], k[:n])
            qe = ctx.compress(ctx.scalar_mul_var_element(P_all[:n], k[:n]))
        if n <= 2500:                                              # one wave per element, forced beyond its size
            with ctx.tuning(small_max=1 << 24, tiny_max=1 << 24):
                qw = ctx.scalar_mul_var(enc_all[:n], k[:n])
                qwe = ctx.compress(ctx.scalar_mul_var_element(P_all[:n], k[:n]))
            if not (torch.equal(qw[0], q[0]) and torch.equal(qw[1], q[1]) and torch.equal(qwe, qe)):
                bad += 1
                print("MISMATCH variable base (waves) n = %d" % n, flush=True)
        with ctx.tuning(small_max=0):
            l = ctx.scalar_mul_var(enc_all[:n], k[:n])
            le_ = ctx.compress(ctx.scalar_mul_var_element(P_all[:n], k[:n]))
        ok = torch.equal(q[0], l[0]) and torch.equal(q[1], l[1]) and torch.equal(qe, le_)
        sel = np.unique(rng.integers(0, n, 6))
        o, st = orc.scalar_mul_var(enc_all[:n][sel].cpu().numpy(), k[:n][sel].cpu().numpy())
        ok = ok and (q[0][sel].cpu().numpy() == o).all() and (q[1][sel].cpu().numpy() == st).all()
        bad += 0 if ok else 1
        cnt += 1
        if not ok:
            print("MISMATCH variable base n = %d" % n, flush=True)
    print("variable base: %d sizes in [1, %d], quads == lanes == oracle sample: %s" % (cnt, 3 * one_gen, "ok" if not bad else "FAILED"), flush=True)

    # MSM, both input forms
    bad0, cnt = bad, 0
    for n in sizes(1, 5 * one_gen, [16, 17, 128 * 16, one_gen]):
        with ctx.tuning(msm_small_max=1 << 24, msm_tiny_max=0):
            a = bytes(ctx.msm(P_all[:n], k[:n])[0])
            ae = ctx.msm(enc_all[:n], k[:n])
        if n <= 3000:                                              # one wave per point (lane-spread arithmetic), forced beyond its size
            with ctx.tuning(msm_small_max=1 << 24, msm_tiny_max=1 << 24):
                aw = bytes(ctx.msm(P_all[:n], k[:n])[0])
                awe = ctx.msm(enc_all[:n], k[:n])
            if aw != a or bytes(awe[0]) != bytes(ae[0]) or not torch.equal(torch.as_tensor(awe[2]), torch.as_tensor(ae[2])):
                bad += 1
                print("MISMATCH msm (waves) n = %d" % n, flush=True)
        with ctx.tuning(msm_small_max=0):
            b = bytes(ctx.msm(P_all[:n], k[:n])[0])
            be = ctx.msm(enc_all[:n], k[:n])
        ok = a == b and bytes(ae[0]) == bytes(be[0]) and torch.equal(torch.as_tensor(ae[2]), torch.as_tensor(be[2]))
        if n <= 300:
            ok = ok and a == bytes(orc.msm(P_all[:n].cpu().numpy(), k[:n].cpu().numpy())[0])
        bad += 0 if ok else 1
        cnt += 1
        if not ok:
            print("MISMATCH msm n = %d" % n, flush=True)
    print("msm: %d sizes in [1, %d], quads == buckets (Elements and Encodings, statuses too), oracle below 300 points: %s"
          % (cnt, 5 * one_gen, "ok" if bad == bad0 else "FAILED"), flush=True)

    # fixed base around the wide-launch threshold (and small sizes)
    bad0, cnt = bad, 0
    big = rnd((2 << 20) + 70000)
    for n in sizes(1, 200000, []) [: rounds // 2] + sizes((2 << 20) - 60000, (2 << 20) + 60000, [2 << 20])[: rounds // 2 + 3]:
        with ctx.tuning(fb_wide=0, fb_k=8):
            a = ctx.scalar_mul_base(big[:n])
        with ctx.tuning(fb_wide=1, fb_k=16):
            b = ctx.scalar_mul_base(big[:n])
        c = ctx.scalar_mul_base(big[:n])
        ok = torch.equal(a, b) and torch.equal(a, c)
        sel = np.unique(rng.integers(0, n, 6))
        ok = ok and (a[sel].cpu().numpy() == orc.scalar_mul_base(big[:n][sel].cpu().numpy())).all()
        bad += 0 if ok else 1
        cnt += 1
        if not ok:
            print("MISMATCH fixed base n = %d" % n, flush=True)
    print("fixed base: %d sizes (small, and around 2^21), narrow == wide == default == oracle sample: %s" % (cnt, "ok" if bad == bad0 else "FAILED"), flush=True)
    # the smallest batches: one scalar per wave (fixed base), four elements per wave (the square-root family), forced beyond
    # their sizes, against the lane-per-element kernels
    bad0, cnt = bad, 0
    r1 = rnd(6000)
    for n in sizes(1, 6000, [4, cus * 4, cus * 16]):
        if n > 6000:
            continue
        res = []
        for kv in (dict(small_max=0), dict(tiny_max=0, small_max=1 << 24), dict(tiny_max=1 << 24)):   # lanes, quads (fixed base), waves
            with ctx.tuning(**kv):
                P, st = ctx.decompress(enc_all[:n])
                res.append([ctx.scalar_mul_base(k[:n]), *ctx.sqrt_ratio_zeta(r0[:n], r1[:n]), *ctx.sqrt_ratio_zeta(r0[:n], r1[:n], root="min_curve"),
                            P, st, *ctx.roundtrip(enc_all[:n]), ctx.compress(P_all[:n]), ctx.encode_to_curve(r0[:n]), ctx.hash_to_curve(r0[:n], r1[:n])])
        ok = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(*res))
        sel = np.unique(rng.integers(0, n, 4))
        ok = ok and (res[2][-1][sel].cpu().numpy() == orc.hash_to_curve(r0[:n][sel].cpu().numpy(), r1[:n][sel].cpu().numpy())).all()
        ok = ok and (res[1][0][sel].cpu().numpy() == res[2][0][sel].cpu().numpy()).all() and (res[2][0][sel].cpu().numpy() == orc.scalar_mul_base(k[:n][sel].cpu().numpy())).all()
        bad += 0 if ok else 1
        cnt += 1
        if not ok:
            print("MISMATCH smallest batches n = %d" % n, flush=True)
    print("smallest batches: %d sizes in [1, 6000], fixed base / sqrt (both roots) / decompress / round trip / compress / encode_to_curve / hash_to_curve, "
          "waves == lanes == oracle sample: %s" % (cnt, "ok" if bad == bad0 else "FAILED"), flush=True)
    # small sums (d377_batch_msm_small): a wave per sum | a lane per sum, random term counts, Elements and Encodings (invalid
    # ones sprinkled in: reported and left out of their sums), against the composition they replace and an oracle sample
    bad0, cnt = bad, 0
    for n in sizes(1, 3 * one_gen // 4, [cus * 16 - 1, cus * 16, cus * 16 + 1])[: rounds + 9]:
        m = int(rng.integers(1, 9))
        n = max(1, min(n, nmax // m))
        t = n * m
        res = []
        for kv in (dict(tiny_max=0), dict(tiny_max=1 << 20), dict()):                                   # lanes, waves, default
            if "tiny_max" in kv and kv["tiny_max"] and n > 3000:
                continue
            with ctx.tuning(**kv):
                e_el, x_el = ctx.msm_small(P_all[:t], k[:t], m, elements=True)
                e_en, x_en, st_en = ctx.msm_small(enc_all[:t], k[:t], m, elements=True)
                res.append((e_el, e_en, st_en))
                # the Element form of the sums: representatives differ between the kernels, their Encodings do not
                if not (torch.equal(ctx.compress(x_el), e_el) and torch.equal(ctx.compress(x_en), e_en)):
                    bad += 1
                    print("MISMATCH small sums (Element records) n = %d m = %d" % (n, m), flush=True)
        ok = all(torch.equal(a, b) for r in res[1:] for a, b in zip(res[0], r))
        acc = ctx.scalar_mul_var_element(P_all[:t][0::m].contiguous(), k[:t][0::m].contiguous())
        for j in range(1, m):
            acc = ctx.add(acc, ctx.scalar_mul_var_element(P_all[:t][j::m].contiguous(), k[:t][j::m].contiguous()))
        ok = ok and torch.equal(res[0][0], ctx.compress(acc))
        st_ref = ctx.decompress(enc_all[:t])[1]
        ok = ok and torch.equal(res[0][2], st_ref)
        for i in np.unique(rng.integers(0, n, 3)):                   # the Encodings form on the oracle: the fold over the valid terms
            e, kk = enc_all[i * m:(i + 1) * m].cpu().numpy(), k[i * m:(i + 1) * m].cpu().numpy()
            prod, st = orc.scalar_mul_var(e, kk)
            accp = ctx.identity()[None, :].copy()
            for q in orc.decompress(prod[st == 0])[0]:
                accp = orc.add_xyzt(accp, q[None, :])
            want = orc.compress(accp)[0]
            ok = ok and (res[0][1][i].cpu().numpy() == want).all()
        bad += 0 if ok else 1
        cnt += 1
        if not ok:
            print("MISMATCH small sums n = %d m = %d" % (n, m), flush=True)
    print("small sums: %d (n, m) pairs, n in [1, %d], m in [1, 8], waves == lanes == default == composition (Elements), Element records encode to the same bytes, "
          "statuses == decompress's, Encodings == oracle fold sample: %s" % (cnt, 3 * one_gen // 4, "ok" if bad == bad0 else "FAILED"), flush=True)
    print("ROUTE_STRESS_%s" % ("OK" if bad == 0 else "FAILED"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
