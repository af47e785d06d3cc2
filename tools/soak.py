#!/usr/bin/env python3
"""Large-sample parity run on the GPU box: every hot-path operation on millions of seeded random inputs (plus
structured ones: invalid encodings, zero / tiny / near-r scalars, small field elements) compared byte for byte with the
C restatement of the reference (oracle/, all host threads).  Test infrastructure, like tests/: the oracle is the
checker.  usage: python tools/soak.py [log2n_var=22] [seed=1]   -> one summary line per operation, exit 1 on mismatch"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import decaf377_amd as d  # noqa: E402
from _oracle import Oracle  # noqa: E402
from _kat_inputs import groth16_regression_inputs  # noqa: E402

R_ORDER = 2111115437357092606062206234695386632838870926408408195193685246394721360383


def le(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    orc = Oracle(native=True)
    threads = len(os.sched_getaffinity(0))
    ctx = d.Context([0])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda n: torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    bad = 0

    def check(name, gpu_outs, op, a, b):
        nonlocal bad
        t0 = time.time()
        o, st, _ = orc.run_threads(op, a.cpu().numpy(), None if b is None else b.cpu().numpy(), threads)
        ok = bool((gpu_outs[0].cpu().numpy() == o).all())
        if len(gpu_outs) > 1:
            ok = ok and bool((gpu_outs[1].cpu().numpy() == st).all())
        print("%-18s n = %9d  %s  (oracle %.1f s on %d threads)" % (name, a.shape[0], "bit-exact" if ok else "MISMATCH", time.time() - t0, threads), flush=True)
        bad += 0 if ok else 1

    n = 1 << lg
    # the reference's shrunk gadget inputs (tests/groth16_gadgets.proptest-regressions:7-15) ride along in every batch
    kat = {k_: torch.from_numpy(v.copy()).to(dev) for k_, v in groth16_regression_inputs().items()}
    # Elligator on random field elements and on small ones
    r0 = rnd(2 * n)
    r0[:4096] = 0
    r0[:4096, 0] = torch.arange(4096, device=dev).to(torch.uint8)
    r0[:4096, 1] = (torch.arange(4096, device=dev) >> 8).to(torch.uint8)
    r0[4096:4098] = kat["fq"]
    enc = ctx.encode_to_curve(r0)
    check("encode_to_curve", (enc,), "encode_to_curve", r0, None)
    # variable base: valid points, with invalid encodings and special scalars sprinkled in
    pts = enc[:n].clone()
    k = rnd(n)
    pts[5::1009] = rnd(pts[5::1009].shape[0])                      # raw strings: almost all invalid
    pts[7::4099] = 0                                               # the identity
    specials = [0, 1, 2, 3, R_ORDER - 1, R_ORDER, R_ORDER + 1, (R_ORDER - 1) // 2, (R_ORDER + 1) // 2, 2**251 - 1, 2**256 - 1]
    for j, v in enumerate(specials):
        k[11 + j::8191] = torch.from_numpy(le(v % 2**256).copy()).to(dev)
    pts[6000:6016] = kat["points"].repeat_interleave(4, dim=0)     # each reference point x each reference scalar
    k[6000:6016] = kat["scalars"].repeat(4, 1)
    out, st = ctx.scalar_mul_var(pts, k)
    check("scalar_mul_var", (out, st), "scalar_mul_var", pts, k)
    # the same (point, scalar) pairs as three-term sums (d377_batch_msm_small on Encodings; an invalid one drops out of its sum):
    # the oracle folds its own products -- a failed product is the zero string, the identity's Encoding
    ns = min(n // 3, 1 << 18)
    t0 = time.time()
    se, sx, sst = ctx.msm_small(pts[: 3 * ns], k[: 3 * ns], 3, elements=True)
    prod, pst, _ = orc.run_threads("scalar_mul_var", pts[: 3 * ns].cpu().numpy(), k[: 3 * ns].cpu().numpy(), threads)
    terms = orc.decompress(prod)[0]
    fold = orc.compress(orc.add_xyzt(orc.add_xyzt(terms[0::3], terms[1::3]), terms[2::3]))
    ok = bool((se.cpu().numpy() == fold).all()) and bool((sst.cpu().numpy() == pst).all()) and bool(torch.equal(ctx.compress(sx), se))
    print("%-18s n = %9d  %s  (three-term sums, Encodings and Element records; oracle %.1f s)" % ("msm_small", ns, "bit-exact" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
    kb = k[: n // 2]
    check("scalar_mul_base", (ctx.scalar_mul_base(kb),), "scalar_mul_base", kb, None)
    raw = torch.cat([enc[: n // 2], rnd(n // 2)])
    rt, st = ctx.roundtrip(raw)
    check("roundtrip", (rt, st), "roundtrip", raw, None)
    num, den = rnd(n), rnd(n)
    den[3::997] = 0
    num[4::997] = 0
    num[5:7] = kat["fq"]
    den[5:7] = kat["fq"].flip(0)
    root, ws = ctx.sqrt_ratio_zeta(num, den)
    check("sqrt_ratio_zeta", (root, ws), "sqrt_ratio_zeta", num, den)
    # small batches take other kernels (one element per quad of lanes up to 16 384 elements; inversion-free roots below 3
    # elements per lane): many calls of awkward sizes, checked as one concatenated batch
    sizes = [1, 2, 3, 15, 16, 17, 63, 64, 65, 255, 1000, 4097, 16383, 16384, 16385, 40000] * 8
    off = 0
    outs, sts, encs = [], [], []
    for sz in sizes:
        if off + sz > n:
            break
        p_, k_ = pts[off: off + sz], k[off: off + sz]
        o_, s_ = ctx.scalar_mul_var(p_.contiguous(), k_.contiguous())
        outs.append(o_); sts.append(s_)
        encs.append(ctx.encode_to_curve(r0[off: off + sz].contiguous()))
        off += sz
    check("var-base, small calls", (torch.cat(outs), torch.cat(sts)), "scalar_mul_var", pts[:off], k[:off])
    check("Elligator, small calls", (torch.cat(encs),), "encode_to_curve", r0[:off], None)
    nh = min(n // 2, 1 << 18)                                      # the oracle's hash_to_curve is single-threaded
    h = ctx.hash_to_curve(r0[:nh], r0[n: n + nh])
    ho = orc.hash_to_curve(r0[:nh].cpu().numpy(), r0[n: n + nh].cpu().numpy())
    ok = bool((h.cpu().numpy() == ho).all())
    print("%-18s n = %9d  %s" % ("hash_to_curve", nh, "bit-exact" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
