"""Worker for tests/test_gpu_parity.py::test_rccl_sharding_single_rank: run under torch.distributed.run with the
real `nccl` backend (= RCCL) on however many GPUs the launcher gives it (one on the builder's boxes).  Every
collective the path has -- scatter of input records, gather of outputs, all-gather of MSM partial sums, the MAX
all-reduce of the timings -- runs on HBM tensors through RCCL and is checked against the oracle (the checker)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import decaf377_amd as d  # noqa: E402
from decaf377_amd import sharding  # noqa: E402
from _oracle import Oracle  # noqa: E402


def main():
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = Oracle()
    ctx = d.Context([local])
    for n in (1, 5, 1000, 4099):
        if rank == 0:
            rng = np.random.default_rng(2000 + n)
            pts = orc.encode_to_curve(rng.integers(0, 256, (n, 32), dtype=np.uint8))
            pts[::11, 31] |= 0x80
            k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            full_p, full_k = torch.from_numpy(pts).to(dev), torch.from_numpy(k).to(dev)
        else:
            full_p = full_k = None
        out, st = sharding.map_from_root(lambda p, s: ctx.scalar_mul_var(p, s) if p.shape[0] else
                                         (torch.zeros((0, 32), dtype=torch.uint8, device=dev), torch.zeros((0,), dtype=torch.uint8, device=dev)),
                                         [(full_p, (32,), torch.uint8), (full_k, (32,), torch.uint8)], n, [None, None], dev)
        torch.cuda.synchronize()
        if rank == 0:
            e_out, e_st = orc.scalar_mul_var(pts, k)
            assert (out.cpu().numpy() == e_out).all() and (st.cpu().numpy() == e_st).all(), n
    # the MSM exchange step over RCCL
    n = 3000
    rng = np.random.default_rng(78)
    enc = orc.encode_to_curve(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    lo, hi = sharding.shard_bounds(n, world, rank)
    e = sharding.msm_sharded(ctx, torch.from_numpy(enc[lo:hi]).to(dev), torch.from_numpy(k[lo:hi]).to(dev))
    torch.cuda.synchronize()
    xyzt, _ = orc.decompress(enc)
    assert bytes(e.cpu().numpy()) == bytes(orc.msm(xyzt, k)[0])
    t = sharding.max_over_ranks(0.25 + rank, dev)
    assert abs(t - (0.25 + world - 1)) < 1e-9
    dist.barrier()
    if rank == 0:
        print("RCCL_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
