"""Worker for tests/test_sharding.py: run under torch.distributed.run with the gloo backend.
The per-shard computation is the CPU oracle (this file lives under tests/: the oracle is the
checker here, the thing under test is the sharding / scatter / gather / timing plumbing)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from decaf377_amd import sharding  # noqa: E402
from _oracle import Oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = Oracle()
    dev = torch.device("cpu")
    for n in (0, 1, 5, 64, 257):
        if rank == 0:
            rng = np.random.default_rng(1000 + n)
            r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            pts = orc.encode_to_curve(r0) if n else np.zeros((0, 32), np.uint8)
            full_p, full_k = torch.from_numpy(pts), torch.from_numpy(k)
        else:
            full_p = full_k = None

        def op(p, s):
            if p.shape[0] == 0:
                return torch.zeros((0, 32), dtype=torch.uint8), torch.zeros((0,), dtype=torch.uint8)
            o, st = orc.scalar_mul_var(p.numpy(), s.numpy())
            return torch.from_numpy(o), torch.from_numpy(st)

        out, st = sharding.map_from_root(op, [(full_p, (32,), torch.uint8), (full_k, (32,), torch.uint8)],
                                         n, [None, None], dev)
        if rank == 0:
            if n:
                e_out, e_st = orc.scalar_mul_var(pts, k)
                assert (out.numpy() == e_out).all() and (st.numpy() == e_st).all(), n
            else:
                assert out.shape == (0, 32)
        # every record is owned by exactly one rank
        lo, hi = sharding.shard_bounds(n, world, rank)
        cnt = torch.tensor([hi - lo], dtype=torch.int64)
        dist.all_reduce(cnt)
        assert int(cnt.item()) == n
    # sharded MSM exchange step: all-gather of per-rank partial sums, then a local sum
    n = 300
    rng = np.random.default_rng(77)            # same stream on every rank
    P = orc.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    lo, hi = sharding.shard_bounds(n, world, rank)
    _, part = orc.msm(P[lo:hi], k[lo:hi], threads=1)
    allp = sharding.allgather_partials(torch.from_numpy(part.view(np.int64).copy()))
    assert allp.shape == (world, 16)
    acc = allp[0:1].numpy().view(np.uint64)
    for r in range(1, world):
        acc = orc.add_xyzt(acc, allp[r:r + 1].numpy().view(np.uint64))
    assert bytes(orc.compress(acc)[0]) == bytes(orc.msm(P, k, threads=1)[0])
    # bench.py's timed region in its three modes, with the oracle standing in for the kernels
    def make_inputs(count, rk):
        rng = np.random.default_rng(500 + rk)
        r0 = rng.integers(0, 256, (count, 32), dtype=np.uint8)
        pts = orc.encode_to_curve(r0) if count else np.zeros((0, 32), np.uint8)
        return torch.from_numpy(pts), torch.from_numpy(rng.integers(0, 256, (count, 32), dtype=np.uint8))

    def compute(p, s):
        if p.shape[0] == 0:
            return torch.zeros((0, 32), dtype=torch.uint8), torch.zeros((0,), dtype=torch.uint8)
        o, st = orc.scalar_mul_var(p.numpy(), s.numpy())
        return torch.from_numpy(o), torch.from_numpy(st)

    n = 41
    for mode in ("weak", "strong", "from-root"):
        res = sharding.run_job(mode, n, 2, 1, make_inputs, compute, dev)
        assert res["world"] == world and res["elapsed_s"] > 0
        if mode == "weak":
            assert res["units"] == n * world * 2 and res["per_rank"] == n and res["collective_s"] == 0
        else:
            lo, hi = sharding.shard_bounds(n, world, rank)
            assert res["units"] == n * 2 and res["per_rank"] == hi - lo
        if mode == "from-root":
            assert res["collective_s"] > 0
            if rank == 0:
                fp, fk = make_inputs(n, 0)
                e_out, e_st = orc.scalar_mul_var(fp.numpy(), fk.numpy())
                assert (res["outputs"][0].numpy() == e_out).all() and (res["outputs"][1].numpy() == e_st).all()
            else:
                assert res["outputs"] == [None, None]
    t = sharding.max_over_ranks(0.5 + rank, dev)
    assert abs(t - (0.5 + world - 1)) < 1e-9
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
