"""Multi-GPU layout of the hot path: one process per GPU, independent contiguous shards.

Every batch operation is a pure map over elements, so N GPUs need no
data-path collective: rank g owns records [g*n/G, (g+1)*n/G).  The only communication the
path ever has is moving records to and from the rank that happens to hold them --
`scatter_records` / `gather_records`, which ride RCCL over xGMI when the tensors live in HBM
(backend "nccl") and gloo on CPU tensors (tests).  The reference has no counterpart: it is a
single-threaded library (src/lib.rs:1).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous slice [lo, hi) of rank `rank` out of n records over `world` ranks; sizes
    differ by at most one and cover [0, n) exactly."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _pad_rows(n, world):
    return (n + world - 1) // world


def scatter_records(full, n, row_shape, dtype, device, src=0, group=None):
    """Rank `src` holds `full` ([n, *row_shape]); every rank returns its shard_bounds slice.
    Other ranks pass full=None.  Uses one dist.scatter of equal chunks: views of `full` when n divides evenly over the
    ranks (no copy on the root: the collective is all the step pays for), zero-padded copies otherwise."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = _pad_rows(n, world)
    recv = torch.empty((rows,) + tuple(row_shape), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        assert full is not None and full.shape[0] == n
        # views go to the collective as they are, so they must already be what the peers' `recv` is (the padded path
        # converts through its copies)
        if n % world == 0 and full.is_contiguous() and full.dtype == dtype and full.device == torch.device(device) \
                and tuple(full.shape[1:]) == tuple(row_shape):
            chunks = [full[r * rows:(r + 1) * rows] for r in range(world)]
        else:
            chunks = []
            for r in range(world):
                lo, hi = shard_bounds(n, world, r)
                c = torch.zeros((rows,) + tuple(row_shape), dtype=dtype, device=device)
                c[: hi - lo] = full[lo:hi]
                chunks.append(c)
    dist.scatter(recv, chunks, src=src, group=group)
    lo, hi = shard_bounds(n, world, rank)
    return recv[: hi - lo]


def gather_records(local, n, dst=0, group=None):
    """Inverse of scatter_records: rank `dst` returns the [n, ...] tensor, others None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = _pad_rows(n, world)
    lo, hi = shard_bounds(n, world, rank)
    assert local.shape[0] == hi - lo
    if n % world == 0:
        # even shards: every rank sends its tensor as it is, and the root receives straight into the slices of the result
        send = local.contiguous()
        out = bufs = None
        if rank == dst:
            out = torch.empty((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            bufs = [out[r * rows:(r + 1) * rows] for r in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return out
    send = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    send[: hi - lo] = local
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        parts.append(bufs[r][: b - a])
    return torch.cat(parts, dim=0)


def max_over_ranks(seconds, device, group=None):
    """The bench contract's timing rule: the slowest rank's elapsed time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def map_from_root(op, inputs, n, out_specs, device, src=0, group=None):
    """Records start on rank `src`: scatter them, run `op(*local_inputs) -> tuple of tensors`
    on every rank's shard, gather the outputs back to `src`.
    inputs: list of ([n, ...] tensor on src | None elsewhere, row_shape, dtype)."""
    locs = [scatter_records(t, n, rs, dt, device, src, group) for (t, rs, dt) in inputs]
    outs = op(*locs)
    if not isinstance(outs, (tuple, list)):
        outs = (outs,)
    assert len(outs) == len(out_specs)
    return [gather_records(o, n, src, group) for o in outs]


def allgather_partials(partial, group=None):
    """The one exchange step of a sharded multi-scalar multiplication: every rank contributes its
    partial sum (one 128-byte Element record, int64[16]) and receives all of them, [world, 16]."""
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(partial) for _ in range(world)]
    dist.all_gather(bufs, partial.contiguous(), group=group)
    return torch.stack(bufs, dim=0)


def msm_sharded(ctx, points, scalars, group=None):
    """vartime_multiscalar_mul over points sharded across ranks: local Pippenger MSM on this
    rank's shard, all-gather of the per-rank partial sums (RCCL over xGMI for HBM tensors),
    then every rank adds the `world` partial points and compresses.  Returns enc[32] (same on
    every rank)."""
    _, partial, _ = ctx.msm(points, scalars)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        enc, _ = ctx.sum_elements(partial.reshape(1, 16))
        return enc
    allp = allgather_partials(partial, group)
    enc, _ = ctx.sum_elements(allp)
    return enc


def run_job(mode, n, steps, warmup, make_inputs, compute, device, sync=lambda: None, red_device=None, group=None,
            coll_device=None):
    """The timed region of bench.py, for any backend (RCCL on GPUs; gloo + a CPU stand-in in the tests).

    mode  "weak":      every rank owns n records of its own (per-GPU work fixed as ranks are added);
          "strong":    n records in total, rank g owns shard_bounds(n, world, g) of them;
          "from-root": n records in total that live on rank 0: every step scatters the inputs, runs the
                       shard on every rank and gathers the outputs back (the only collectives the path
                       has); their share of the step is returned as collective_s.
    make_inputs(count, rank) -> tuple of [count, ...] tensors on `device` (the rank's synthetic records);
    compute(*inputs) -> tuple of output tensors; sync() drains the device; coll_device: where the collectives'
    tensors must live (the compute device for RCCL; the CPU when HBM tensors are moved by gloo in plumbing tests).
    Returns dict(elapsed_s (max over ranks, `steps` timed steps), units (records all ranks processed in
    them), per_rank (records of this rank per step), collective_s, outputs (last step: this rank's, or
    rank 0's gathered ones in from-root mode))."""
    import time
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if mode not in ("weak", "strong", "from-root"):
        raise ValueError("mode must be weak, strong or from-root")
    if mode == "weak":
        lo, hi = 0, n
        total = n * world
    else:
        lo, hi = shard_bounds(n, world, rank)
        total = n
    from_root = mode == "from-root"
    cdev = coll_device if coll_device is not None else device
    if from_root:
        full = make_inputs(n, 0) if rank == 0 else None
        specs = [(tuple(t.shape[1:]), t.dtype) for t in make_inputs(1, 0)]
    else:
        local = make_inputs(hi - lo, rank)
    coll = 0.0
    outs = None

    def barrier():
        sync()
        if world > 1:
            dist.barrier(group)
        sync()

    def one_step(timed):
        nonlocal coll, outs
        if from_root and dist.is_initialized():          # also at world size 1: the collectives then run through the backend
            t0 = time.perf_counter()
            ins = [scatter_records(full[i].to(cdev) if rank == 0 else None, n, specs[i][0], specs[i][1], cdev, 0,
                                   group).to(device) for i in range(len(specs))]
            sync()
            t1 = time.perf_counter()
            res = compute(*ins)
            sync()
            t2 = time.perf_counter()
            outs = [gather_records(o.to(cdev), n, 0, group) for o in res]
            if rank == 0:
                outs = [o.to(device) for o in outs]
            sync()
            if timed:
                coll += (t1 - t0) + (time.perf_counter() - t2)
        elif from_root:
            outs = list(compute(*full))
        else:
            outs = list(compute(*local))

    for _ in range(warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    rd = red_device if red_device is not None else device
    if world > 1:
        elapsed = max_over_ranks(elapsed, rd, group)
        coll = max_over_ranks(coll, rd, group)
    return {"elapsed_s": elapsed, "units": total * steps, "per_rank": hi - lo, "collective_s": coll, "outputs": outs,
            "world": world, "rank": rank}
