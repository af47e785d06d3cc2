"""One process driving every GPU of the node through ONE d377_ctx (the C ABI's multi-device paths), checked against
the oracle.  Started by tools/multigpu_selftest.py; lives under tests/ because it uses the oracle as its checker.

  * peer access between the context's devices (hipDeviceEnablePeerAccess in d377_ctx_create) as torch sees it
  * every device's tables: the same `_dev` call on a tensor resident on each device returns the same bytes
  * d377_batch_sharded_dev: an HBM-resident batch split over the devices by peer copies, every op code, rooted
    on every device in turn
  * the host-pointer entry points sliced over the devices (one host thread per device)
  * d377_msm / d377_msm_encoded over several devices (per-device partial sums combined on device 0)

With one physical GPU the device list is [0, 0] (slicing, staging and event ordering still run; peer copies do not)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import decaf377_amd as d  # noqa: E402
from _oracle import Oracle  # noqa: E402


def same(a, b):
    a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    b = b.cpu().numpy() if hasattr(b, "cpu") else np.asarray(b)
    return a.shape == b.shape and bool((a.view(np.uint8) == b.view(np.uint8)).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=16)
    ap.add_argument("--devices", type=str, default="")
    ap.add_argument("--require-distinct", action="store_true",
                    help="fail unless the context spans >= 2 physical GPUs with peer access enabled between all of them")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    G = torch.cuda.device_count()
    ids = [int(x) for x in args.devices.split(",")] if args.devices else (list(range(G)) if G > 1 else [0, 0])
    distinct = sorted(set(ids))
    print("devices", ids, "(%d distinct)" % len(distinct), flush=True)
    orc = Oracle()
    n = (1 << args.log2n) + 37                          # ragged: slices of different sizes
    rng = np.random.default_rng(9000)
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ns = min(n, 2048)                                   # oracle sample
    enc_o = orc.encode_to_curve(r0[:ns])
    raw_mark = slice(0, None, 7)

    for a in distinct:
        for b in distinct:
            if a != b:
                print("peer %d -> %d: %s" % (a, b, torch.cuda.can_device_access_peer(a, b)), flush=True)

    ctx = d.Context(ids)
    assert ctx.device_ids == ids
    # what d377_ctx_create arranged between the context's devices (d377_ctx_peer_access): the lines of the library that
    # only run between DISTINCT devices (hipDeviceEnablePeerAccess, hipMemcpyPeerAsync across ids, cross-device
    # hipStreamWaitEvent in run_sharded_dev) are covered by this leg iff some pair reports 2
    pairs = [(a, b, ctx.peer_access(a, b)) for a in range(len(ids)) for b in range(len(ids)) if a != b]
    n_peer = sum(1 for _, _, v in pairs if v == 2)
    print("peer access inside the context: %d of %d ordered pairs enabled, %d pairs are one GPU listed twice"
          % (n_peer, len(pairs), sum(1 for _, _, v in pairs if v == 1)), flush=True)
    if args.require_distinct:
        if len(distinct) < 2 or n_peer != len(pairs):
            print("REQUIRE_DISTINCT_FAILED devices=%s pairs=%s: the distinct-device paths were NOT exercised" % (ids, pairs), flush=True)
            sys.exit(3)
    single = d.Context([distinct[0]])
    dev0 = torch.device("cuda", distinct[0])
    t0 = lambda x: torch.from_numpy(x).to(dev0)
    enc_ref = single.encode_to_curve(t0(r0))
    assert same(enc_ref[:ns], enc_o), "encode_to_curve differs from the oracle"
    raw = enc_ref.clone()
    raw[raw_mark, 31] |= 0x80                           # invalid encodings travel through every slice
    out_ref, st_ref = single.scalar_mul_var(raw, t0(k))
    o_out, o_st = orc.scalar_mul_var(raw[:ns].cpu().numpy(), k[:ns])
    assert same(out_ref[:ns], o_out) and same(st_ref[:ns], o_st), "scalar_mul_var differs from the oracle"
    base_ref = single.scalar_mul_base(t0(k))
    rt_ref = single.roundtrip(raw)
    hash_ref = single.hash_to_curve(t0(r0), t0(r1))
    sq_ref = single.sqrt_ratio_zeta(t0(r0), t0(r1))
    xyzt_ref, dst_ref = single.decompress(enc_ref)
    torch.cuda.synchronize(dev0)

    # every device's own tables and scratch: the same launch on each device
    for slot, g in enumerate(ids):
        dv = torch.device("cuda", g)
        tg = lambda x: x.to(dv)
        o, s = ctx.scalar_mul_var(tg(raw), tg(t0(k)))
        assert same(o, out_ref) and same(s, st_ref), "scalar_mul_var on device %d" % g
        assert same(ctx.scalar_mul_base(tg(t0(k))), base_ref), "scalar_mul_base on device %d" % g
        assert same(ctx.encode_to_curve(tg(t0(r0))), enc_ref), "encode_to_curve on device %d" % g
        r_, s_ = ctx.roundtrip(tg(raw))
        assert same(r_, rt_ref[0]) and same(s_, rt_ref[1]), "roundtrip on device %d" % g
        e_, x_, _ = ctx.msm(tg(enc_ref[:5000]), tg(t0(k[:5000])))
        torch.cuda.synchronize(dv)
        if slot == 0:
            msm_first = e_.cpu()
        assert same(e_, msm_first), "msm on device %d" % g
        print("device %d: per-device launches ok" % g, flush=True)

    # d377_batch_sharded_dev, rooted on every device in turn
    for root in distinct:
        dv = torch.device("cuda", root)
        tr = lambda x: x.to(dv)
        assert same(ctx.sharded("encode_to_curve", tr(t0(r0)))[0], enc_ref)
        o, s = ctx.sharded("scalar_mul_var", tr(raw), tr(t0(k)))
        assert same(o, out_ref) and same(s, st_ref)
        o, s = ctx.sharded("roundtrip", tr(raw))
        assert same(o, rt_ref[0]) and same(s, rt_ref[1])
        x, s = ctx.sharded("decompress", tr(enc_ref))
        assert same(x, xyzt_ref) and same(s, dst_ref)
        assert same(ctx.sharded("compress", x)[0], enc_ref)
        assert same(ctx.sharded("scalar_mul_base", tr(t0(k)))[0], base_ref)
        assert same(ctx.sharded("hash_to_curve", tr(t0(r0)), tr(t0(r1)))[0], hash_ref)
        o, s = ctx.sharded("sqrt_ratio_zeta", tr(t0(r0)), tr(t0(r1)))
        assert same(o, sq_ref[0]) and same(s, sq_ref[1])
        el = ctx.sharded("scalar_mul_var_element", x, tr(t0(k)))[0]
        assert same(ctx.compress(el), single.compress(single.scalar_mul_var_element(xyzt_ref, t0(k))))
        torch.cuda.synchronize(dv)
        print("sharded_dev rooted on device %d ok" % root, flush=True)

    # host-pointer path sliced over the devices
    h_out, h_st = ctx.scalar_mul_var(raw.cpu().numpy(), k)
    assert same(h_out, out_ref) and same(h_st, st_ref)
    assert same(ctx.encode_to_curve(r0), enc_ref)
    assert same(ctx.scalar_mul_base(k), base_ref)
    print("host path over %d devices ok" % len(ids), flush=True)

    # multi-device MSM (host path): per-device Pippenger, partial sums combined on device 0
    for m in (1, 37, 5000, n):
        pts = xyzt_ref[:m].cpu().numpy().view(np.uint64)
        e_multi, x_multi, _ = ctx.msm(pts, k[:m])
        e_single, _, _ = single.msm(pts, k[:m])
        assert bytes(e_multi) == bytes(e_single), "multi-device msm, n = %d" % m
        e_enc, _, st = ctx.msm(raw[:m].cpu().numpy(), k[:m])
        e_enc1, _, st1 = single.msm(raw[:m].cpu().numpy(), k[:m])
        assert bytes(e_enc) == bytes(e_enc1) and same(st, st1), "multi-device msm_encoded, n = %d" % m
        if m <= 5000:
            assert bytes(e_multi) == bytes(orc.msm(pts, k[:m])[0]), "msm vs oracle, n = %d" % m
    print("multi-device msm ok", flush=True)
    ctx.close()
    single.close()
    print("CTX_LEG_OK devices=%s distinct_devices=%d peer_pairs_enabled=%d%s" %
          (ids, len(distinct), n_peer, "" if n_peer else "  (same-device run: peer copies and cross-device events NOT covered)"), flush=True)


if __name__ == "__main__":
    main()
