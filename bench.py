#!/usr/bin/env python3
"""bench.py -- throughput of the decaf377 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the headline workload over one batch of synthetic input that is
already resident in HBM: BASELINE.json configs[3], 2^22 variable-base scalar multiplications
(random Element x random Fr: decompress, [k]P, compress).  Rank 0 prints ONE JSON line.

  --scaling weak      (default) 2^22 elements PER GPU: every rank owns its own shard, no data-path collective
  --scaling strong    configs[3] as written: 2^22 elements IN TOTAL, rank g owns n/G of them
  --from-root         (with either total) the batch lives on rank 0: every step scatters the inputs over
                      RCCL, runs the shards, gathers the outputs back; `collective_ms` reports their share

Besides the contract keys the line carries
  roofline       HBM view of the dominant kernel (k_scalar_mul_var): algorithmic bytes / launch
                 duration (HIP events on the launch stream) against the 8 TB/s peak;
  roofline_valu  the view that actually binds: integer MACs/s against the v_mad_u64_u32 issue
                 ceiling (one wave-instruction per 4 cycles per SIMD; tools/valu_mix.hip, profiles/r02_valu_mix*.txt);
  cpu_baseline   the C restatement of the reference algorithm (oracle/, kind "port") timed on the
                 host cores of this box on the same workload (rank 0, N = 1 only);
  extra          encodes/s of the other batch operations (round trip, Elligator, fixed base, sqrt, decompress, compress,
                 hash_to_curve, MSM), each with its own roofline_valu (BASELINE.json configs[1], [2], [4] are the round
                 trip, fixed base and Elligator entries).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def csrc_sha256():
    """Identity of the kernel sources (the same hash tools/pmc_summarize.py stores in profiles/pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "decaf377_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()
sys.path.insert(0, ROOT)

# algorithmic bytes and reference-algorithm work per unit
ALGO_BYTES = {"scalar_mul_var": 97, "roundtrip": 65, "encode_to_curve": 64, "scalar_mul_base": 64,
              "sqrt_ratio_zeta": 97}
# Field products (M, 153 v_mad_u64_u32 each) and squarings (S, 117) one element executes, counted by the
# instrumented host build of the same headers (tests/test_host_sim.py::test_bench_mac_counts):
# variable base = 1 square root (decompression, handed the inverse of its denominator: 242 S + 75 M with its extras)
# + the 9-entry table + 63 windows x (4 doublings of 3 S + 4 M, + 1 M for T, + a 7 M cached addition) on k/2,
# + the state of the final doubling (3 S + 5 M) + the square-root-free compression (7 M) + the two batched inversions
# of a chunk (denominators, compressor: 3-4 M per element each, and one divsteps inversion per lane per 8 elements
# each: inv30.hpp, 20 rounds x 90 signed 64-bit MACs on 30-bit limbs, + 2 M).  The 2^20 extras are counted at the
# 8 elements per lane they have.
KERNEL_OPS = {"scalar_mul_var": (1668.5, 1009.0), "roundtrip": (177, 580), "scalar_mul_base": (85.25, 3.0),
              "sqrt_ratio_zeta": (75.25, 241.0), "encode_to_curve": (102.5, 243.0), "hash_to_curve": (202.75, 491.0),
              "decompress": (93.0, 291.0), "compress": (92.0, 289.0),
              # decompress from 2 elements per resident lane (262 144 on 256 CUs): chunks with batched inverses (d377.hip)
              "decompress_chunked": (88.25, 246.0), "compress_chunked": (91.25, 243.0), "roundtrip_chunked": (168.5, 489.0)}
# divsteps inversions per element (one per lane per 8 elements and per batched-inversion pass of the kernel)
KERNEL_INVERSIONS = {"scalar_mul_var": 2 / 8.0, "scalar_mul_base": 1 / 8.0, "sqrt_ratio_zeta": 1 / 8.0, "encode_to_curve": 2 / 8.0,
                     "hash_to_curve": 3 / 8.0, "decompress_chunked": 1 / 8.0, "compress_chunked": 1 / 8.0, "roundtrip_chunked": 2 / 8.0}
MACS_PER_MUL, MACS_PER_SQR = 153, 117
DIVSTEP_MACS_PER_INVERSION = 20 * 90                       # v_mad_i64_i32: update_fg_30 (36) + update_de_30 (54) per round
KERNEL_MACS = {k: m * MACS_PER_MUL + s * MACS_PER_SQR + KERNEL_INVERSIONS.get(k, 0.0) * DIVSTEP_MACS_PER_INVERSION
               for k, (m, s) in KERNEL_OPS.items()}                                           # scalar_mul_var: 373784
MSM_MACS_PER_ADDITION = 7 * MACS_PER_MUL                   # one mixed addition per point and window: 7 field products


def valu_view(macs_per_element, n, kernel_ms):
    """roofline_valu of one extra: executed MACs / s against the v_mad_u64_u32 issue ceiling."""
    rate = macs_per_element * n / (kernel_ms * 1e-3)
    return {"bound": "valu_int32_mac", "macs_per_element": macs_per_element, "achieved": rate / 1e12, "peak": VALU_MAC_PEAK / 1e12,
            "unit": "TMAC/s", "frac": rate / VALU_MAC_PEAK}
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
# v_mad_u64_u32 issues one wave-instruction per 4 cycles per SIMD (16 lanes / cycle): 256 CUs x 4 SIMDs x 16
# lanes x 2.4 GHz.  Measured on this chip: 3.74-3.80e13/s = 95-97 % of it, because the sustained clock under this
# load is ~2.3 GHz (tools/valu_mix.hip, tools/valu_mix2.hip -> profiles/r02_valu_mix.txt, r02_valu_mix2.txt).
# Round 1 quoted 3.28e13 from a loop with 8 instructions per branch; that figure was the benchmark's, not the chip's.
VALU_MAC_PEAK = 256 * 4 * 16 * 2.4e9  # 3.93e13 MAC/s
VALU_MAC_PEAK_MEASURED = 3.80e13


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=22, help="2^log2n elements per step: per GPU (weak) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--from-root", action="store_true",
                    help="the batch starts and ends on rank 0: scatter -> shards -> gather inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--launch-check", action="store_true",
                    help="plumbing test of the self-launcher: ranks rendezvous over gloo on the CPU, rank 0 prints "
                         "{launch_check, ranks_seen}, nothing touches a GPU")
    ap.add_argument("--same-device", action="store_true",
                    help="plumbing test: every rank uses GPU 0 (needs --backend gloo; not a measurement)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here.  This parent makes no GPU
    call (it never imports torch): a process that has opened the GPU must not spawn programs on this pool, so the
    ranks are started first and the parent only relays rank 0's JSON line and the exit codes."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "D377_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    rc = 0
    try:
        # rank 0's stdout is small (one line): read it to the end, then reap everyone; a rank that dies takes the others
        # down with it instead of leaving them in a collective
        import threading
        def drain():
            nonlocal out0
            out0 = procs[0].stdout.read()
        th = threading.Thread(target=drain, daemon=True)
        th.start()
        live = set(range(args.gpus))
        deadline = time.time() + float(os.environ.get("D377_BENCH_LAUNCH_TIMEOUT_S", "3600"))
        while live:
            if time.time() > deadline:
                sys.stderr.write("bench.py: ranks still running after the launch timeout; stopping them\n")
                rc = rc or 124
                for o in live:
                    procs[o].kill()
                break
            for r in list(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if c != 0 and rc == 0:
                        rc = c if c > 0 else 1
                        sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
                        for o in live:
                            procs[o].terminate()
            time.sleep(0.05)
        th.join(timeout=10)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    raise SystemExit(rc)


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            c = max(1, min(c, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(c, 256))


def time_op(torch, fn, steps, warmup):
    """Returns (avg kernel ms via HIP events on the current stream, wall ms per step).  The extras that last milliseconds or
    less are given 5-10 untimed calls: after a stretch of small kernels the clock needs that long to come back up."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / steps
    ker = sum(a.elapsed_time(b) for a, b in evs) / steps
    return ker, wall


def main():
    args = parse()
    # The checker's libraries are built BEFORE anything opens the GPU (and before the ranks exist when this process is
    # the launcher): later on Oracle(build=False) only loads them -- no rank starts a program once it holds a GPU.
    if not args.launch_check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _oracle import prebuild
        prebuild(native=not args.no_cpu_baseline and int(os.environ.get("RANK", "0")) == 0)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args)                          # never returns
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "ranks_seen": dist.get_world_size(), "rank_sum": int(t.item()),
                              "self_launched": bool(os.environ.get("D377_BENCH_SELF_LAUNCHED"))}), flush=True)
        dist.destroy_process_group()
        return
    import decaf377_amd as d
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    if args.same_device:
        local = 0
    ndev = torch.cuda.device_count()
    if ndev and local >= ndev:
        if os.environ.get("D377_BENCH_SELF_LAUNCHED") and not args.same_device:
            raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s) (--same-device --backend gloo is the "
                             "plumbing mode for one GPU)" % (args.gpus, ndev))
        local = local % ndev            # a launcher that masks devices per rank leaves one visible GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run, also with a single rank (RCCL smoke path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    # the 5.9 GB fixed-base comb is built by the first fixed-base call (the `extra` legs), not at context creation: a rank
    # whose job has no fixed-base leg (--no-extra, the multi-GPU headline) never pays for it (d377_ctx_create_ex)
    ctx = d.Context([local], comb_lazy=True)
    from decaf377_amd import sharding
    n = 1 << args.log2n
    mode = "from-root" if args.from_root else args.scaling
    if args.from_root and args.scaling == "weak":
        n_job = n * world                # the root holds world x 2^log2n records
    else:
        n_job = n

    # synthetic records, generated on the device and resident before timing
    def make_inputs(count, rk):
        g = torch.Generator(device=dev).manual_seed(666 + rk)
        r0_ = torch.randint(0, 256, (count, 32), dtype=torch.uint8, device=dev, generator=g)
        k_ = torch.randint(0, 256, (count, 32), dtype=torch.uint8, device=dev, generator=g)
        return ctx.encode_to_curve(r0_), k_      # valid encodings, strategy of tests/operations.rs:6-11

    kernel_events = []                           # every launch of compute(), in order (the sub-jobs below slice it)
    last_local = []                              # this rank's last launch: inputs and outputs (parity sample below)

    def compute(points_, scalars_):
        cnt = int(points_.shape[0])
        o = torch.empty((cnt, 32), dtype=torch.uint8, device=dev)
        st = torch.empty((cnt,), dtype=torch.uint8, device=dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()                               # torch's current stream is the launch stream of the _dev call
        if cnt:
            ctx.scalar_mul_var(points_, scalars_, outs=[o, st])
        b.record()
        kernel_events.append((a, b))
        last_local[:] = [points_, scalars_, o, st]
        return o, st

    res = sharding.run_job(mode, n_job, args.steps, args.warmup, make_inputs, compute, dev,
                           sync=torch.cuda.synchronize, red_device=red_dev, coll_device=red_dev)
    elapsed = res["elapsed_s"]
    timed = kernel_events[-args.steps:]
    kernel_ms = sum(a.elapsed_time(b) for a, b in timed) / max(1, len(timed))
    n_launch = res["per_rank"]                   # records one launch of this rank processed
    out, status = res["outputs"] if rank == 0 else (None, None)
    if rank == 0 and out is not None:
        assert int(status.sum().item()) == 0, "valid inputs must all decode"
    # Every rank checks a sample of ITS OWN last launch against the oracle (the checker; after the timed region):
    # 256 records spread evenly over the shard.  MIN over ranks, so one wrong rank fails the line.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle
    checker = Oracle(build=False)

    def parity_of_last_launch():
        """(ok, records checked): 256 records spread over this rank's last launch against the oracle; MIN over ranks."""
        ok, cnt_ = 1, 0
        if last_local and int(last_local[0].shape[0]) > 0:
            lp, lk, lo_, lst = last_local
            cnt = int(lp.shape[0])
            sel = torch.linspace(0, cnt - 1, steps=min(256, cnt), device=dev).round().to(torch.int64).unique()
            o_out, o_st = checker.scalar_mul_var(lp[sel].cpu().numpy(), lk[sel].cpu().numpy())
            ok = int(bool((lo_[sel].cpu().numpy() == o_out).all() and (lst[sel].cpu().numpy() == o_st).all()))
            cnt_ = int(sel.numel())
        if dist.is_initialized() and world > 1:
            tt = torch.tensor([ok, cnt_], dtype=torch.int64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            ok, cnt_ = int(tt[0].item()), int(tt[1].item())
        return ok, cnt_

    def per_rank_list(value):
        """`value` of every rank, in rank order (a float each)."""
        if not (dist.is_initialized() and world > 1):
            return [float(value)]
        mine = torch.tensor([float(value)], dtype=torch.float64, device=red_dev)
        bufs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(bufs, mine)
        return [float(b.item()) for b in bufs]

    parity_ok, parity_cnt = parity_of_last_launch()
    head_local = list(last_local)                # the headline's last launch on this rank (the cpu_baseline leg compares against it)
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1
    # the extra ops and the CPU baseline below reuse rank 0's records
    g = torch.Generator(device=dev).manual_seed(666 + rank)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    scalars = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    points = ctx.encode_to_curve(r0)

    value = res["units"] / elapsed
    line = {
        "metric": "decaf377 var-base scalar-mults/sec + encodes/sec at 1/2/4/8 MI355X",
        "value": value,
        "unit": "scalar-mults/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "2^%d variable-base scalar mult (random Element x random Fr -> Encoding) %s, "
                        "BASELINE.json configs[3]" % (args.log2n, "per GPU" if args.scaling == "weak" else "in total"),
            "mode": mode,
            "elements_per_gpu": n_launch if mode != "weak" else n,
            "elements_total": res["units"] // args.steps,
            "sharding": ("the batch lives on rank 0: scatter of inputs, independent shards, gather of outputs (RCCL)"
                         if args.from_root else
                         "independent contiguous shards, one process per GPU, no data-path collective"),
            "inputs": "points = encode_to_curve(rand32), scalars = rand32 (reduced mod r on the GPU), seed 666+rank",
        },
    }
    line["ranks_seen"] = ranks_seen              # size of the process group the timed region ran in
    line["backend"] = (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if dist.is_initialized() else "none (one process)"
    line["parity_sample_ok"] = bool(parity_ok)   # every rank: >= parity_sample_per_rank outputs of its last launch == oracle
    line["parity_sample_per_rank"] = parity_cnt
    if args.from_root:
        line["collective_ms"] = res["collective_s"] * 1e3 / args.steps
    n = max(n_launch, 1) if mode != "weak" else n
    algo_bytes = ALGO_BYTES["scalar_mul_var"] * n
    ach = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source, pmc = None, None, {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            pmc = rec.get("k_scalar_mul_var") or {}
            if rec.get("_sources", {}).get("csrc_sha256") != csrc_sha256():
                # counters of other kernels than the ones that just ran are not replayed
                pmc, traffic_source = {}, "stale: profiles/pmc_traffic.json was collected on different kernel sources (tools/collect_pmc.sh)"
            elif pmc.get("elements") == n:
                traffic = pmc["hbm_bytes_per_launch"]     # PMC counters of separate rocprofv3 passes, not of this run
                traffic_source = "profiles/pmc_traffic.json (" + pmc.get("source", "rocprofv3 --pmc passes") + ")"
        except Exception:
            traffic, pmc = None, {}
    line["roofline"] = {
        "kernel": "k_scalar_mul_var",
        "bound": "hbm",
        "achieved": ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "traffic_measured_in_this_run": False,    # a replay of separate rocprofv3 --pmc passes on the builder's box, gated on the kernel sources' hash
        # what the memory system really moves (per-lane window tables in global scratch), against the same peak; an
        # upper bound on HBM bytes: the counters also see Infinity-Cache hits
        "achieved_measured": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
        "frac_measured": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "kernel_ms": kernel_ms,
        "note": "integer-ALU-bound kernel: 3.7e5 32-bit MACs per 97 algorithmic bytes; see roofline_valu",
    }
    # beyond 8 elements per resident lane (2 workgroups x 256 lanes per CU) a chunk shares its two inversions among up to 16
    # elements per lane (dcb.hpp DCB_K_LONG): the divsteps' share of an element is 2/16 instead of 2/8 there
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    macs_el = KERNEL_MACS["scalar_mul_var"]
    if n > 8 * 512 * cus:
        macs_el -= (2 / 8.0 - 2 / 16.0) * DIVSTEP_MACS_PER_INVERSION
    macs = macs_el * n / (kernel_ms * 1e-3)
    line["roofline_valu"] = {
        "bound": "valu_int32_mac",
        "achieved": macs / 1e12,
        "peak": VALU_MAC_PEAK / 1e12,
        "unit": "TMAC/s",
        "frac": macs / VALU_MAC_PEAK,
        "peak_measured": VALU_MAC_PEAK_MEASURED / 1e12,
        "frac_of_measured": macs / VALU_MAC_PEAK_MEASURED,
        "macs_per_element": macs_el,
    }
    if pmc.get("valu_insts_per_element"):
        # every VALU instruction of this stream costs one issue slot: MACs / all VALU instructions is the ceiling
        # of `frac` for this instruction stream, and instructions/s against the issue rate says how full the pipe is
        vi = pmc["valu_insts_per_element"]
        line["roofline_valu"]["valu_insts_per_mac"] = vi / macs_el
        # VALU instructions/s against one wave-instruction per 4 cycles per SIMD AT THE NOMINAL 2.4 GHz (DESIGN.md section 5's
        # "issue-slot utilisation" divides by the cycles the chip really ran, GRBM_GUI_ACTIVE, and reads ~0.04 higher)
        line["roofline_valu"]["valu_issue_frac_at_nominal_clock"] = vi * n / (kernel_ms * 1e-3) / VALU_MAC_PEAK
        line["roofline_valu"]["pmc_source"] = pmc.get("source")

    # ---- N > 1: everything else the first multi-GPU node can give, in the same run -------------------------------
    # The headline above is the weak line the contract asks for (2^log2n per GPU).  BASELINE configs[3] as written is
    # 2^22 pairs IN TOTAL sharded over the GPUs, with the scatter / gather over RCCL when the batch lives on one of them,
    # and configs[4] is the Elligator batch at 2^20 in total: measured here with the same barrier + MAX-over-ranks rule.
    multi = {}
    if world > 1 and not args.no_extra and mode == "weak":
        sub_steps = max(1, min(args.steps, 10))
        for key, sub_mode in (("strong_2^%d_total" % args.log2n, "strong"), ("from_root", "from-root")):
            first_event = len(kernel_events)
            r_ = sharding.run_job(sub_mode, n, sub_steps, 1, make_inputs, compute, dev,
                                  sync=torch.cuda.synchronize, red_device=red_dev, coll_device=red_dev)
            evs = kernel_events[first_event:][-sub_steps:]
            k_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
            ok_, cnt_ = parity_of_last_launch()
            parity_ok = min(parity_ok, ok_)
            if sub_mode == "from-root" and rank == 0:
                assert int(r_["outputs"][1].sum().item()) == 0 and int(r_["outputs"][0].shape[0]) == n
            multi[key] = {
                "mode": sub_mode, "elements_total": n, "steps": sub_steps, "ms_per_step": r_["elapsed_s"] * 1e3 / sub_steps,
                "value": r_["units"] / r_["elapsed_s"], "unit": "scalar-mults/s",
                "kernel_ms_per_rank": per_rank_list(k_ms), "elements_per_rank": [int(v) for v in per_rank_list(r_["per_rank"])],
                "parity_sample_ok": bool(ok_), "parity_sample_per_rank": cnt_,
            }
            if sub_mode == "from-root":
                multi[key]["collective_ms"] = r_["collective_s"] * 1e3 / sub_steps
                multi[key]["note"] = "the batch lives on rank 0: scatter of (Encoding, scalar), shards, gather of (Encoding, status) over " + args.backend
        # configs[4]: 2^20 Elligator maps in total, contiguous shards, no collective
        n_ell = min(1 << 20, n)
        lo_e, hi_e = sharding.shard_bounds(n_ell, world, rank)
        ge = torch.Generator(device=dev).manual_seed(4242 + rank)
        r_e = torch.randint(0, 256, (hi_e - lo_e, 32), dtype=torch.uint8, device=dev, generator=ge)
        o_e = torch.empty_like(r_e)
        torch.cuda.synchronize()
        dist.barrier()
        k_e, _ = time_op(torch, lambda: ctx.encode_to_curve(r_e, outs=[o_e]), 3, 1)
        ks = per_rank_list(k_e)
        sel = torch.linspace(0, hi_e - lo_e - 1, steps=min(64, hi_e - lo_e), device=dev).round().to(torch.int64).unique()
        ok_e = int(bool((o_e[sel].cpu().numpy() == checker.encode_to_curve(r_e[sel].cpu().numpy())).all()))
        if dist.is_initialized():
            tt = torch.tensor([ok_e], dtype=torch.int64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            ok_e = int(tt.item())
        parity_ok = min(parity_ok, ok_e)
        multi["encode_to_curve_2^20_total"] = {"elements_total": n_ell, "kernel_ms_per_rank": ks, "value": n_ell / (max(ks) * 1e-3),
                                               "unit": "encodes/s", "parity_sample_ok": bool(ok_e),
                                               "note": "BASELINE configs[4]: contiguous shards, slowest rank's kernel time"}
        line["parity_sample_ok"] = bool(parity_ok)

    if not args.no_extra:
        extra = dict(multi)
        ne = min(1 << 20, n)
        # from 2 elements per resident lane (2 workgroups x 256 lanes per CU) decompress / compress / round trip run in chunks with
        # batched inverses (d377.hip): fewer products per element, priced as such below
        chunked_min = 2 * 512 * torch.cuda.get_device_properties(dev).multi_processor_count
        enc1 = points[:ne]
        o1 = torch.empty((ne, 32), dtype=torch.uint8, device=dev)
        s1 = torch.empty((ne,), dtype=torch.uint8, device=dev)
        for name, fn in [
            ("roundtrip", lambda: ctx.roundtrip(enc1, outs=[o1, s1])),
            ("encode_to_curve", lambda: ctx.encode_to_curve(r0[:ne], outs=[o1])),
            ("scalar_mul_base", lambda: ctx.scalar_mul_base(scalars[:ne], outs=[o1])),
            ("sqrt_ratio_zeta", lambda: ctx.sqrt_ratio_zeta(r0[:ne], scalars[:ne], outs=[o1, s1])),
        ]:
            ker, _ = time_op(torch, fn, 5, 5)
            ker_all = ker
            if world > 1:                       # whole-job rate: slowest rank's kernel time
                tt = torch.tensor([ker], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ker_all = float(tt.item())
            macs = KERNEL_MACS["roundtrip_chunked"] if name == "roundtrip" and ne >= chunked_min else KERNEL_MACS[name]
            extra[name] = {"n": ne, "kernel_ms": ker, "per_sec": ne / (ker * 1e-3),
                           "per_sec_all_gpus": ne * world / (ker_all * 1e-3),
                           "algo_GBps": ALGO_BYTES[name] * ne / (ker * 1e-3) / 1e9,
                           "roofline_valu": valu_view(macs, ne, ker)}
        # vartime_multiscalar_mul (Pippenger MSM), 2^20 Elements -> one Encoding
        pm, _ = ctx.decompress(enc1)
        ker, _ = time_op(torch, lambda: ctx.msm(pm, scalars[:ne]), 5, 5)
        msm_w = 18 if ne >= (1 << 20) else None             # 14-bit windows from 2^19 points, 16-bit (16 windows) from 3 x 2^20 (msm.hip pick_window)
        extra["msm_2^20"] = {"n": ne, "ms": ker, "per_sec": ne / (ker * 1e-3)}
        if msm_w:
            extra["msm_2^20"]["roofline_valu"] = valu_view(msm_w * MSM_MACS_PER_ADDITION, ne, ker)
            extra["msm_2^20"]["roofline_valu"]["note"] = "one 7-product mixed addition per point and window (18 windows); whole call"
        if n >= (1 << 22):                                  # BASELINE config 5 at its full size: Elements decoded once, outside the timing
            n22 = 1 << 22
            pm22, _ = ctx.decompress(points[:n22])
            ker, _ = time_op(torch, lambda: ctx.msm(pm22, scalars[:n22]), 3, 1)
            extra["msm_2^22"] = {"n": n22, "ms": ker, "per_sec": n22 / (ker * 1e-3),
                                 "roofline_valu": valu_view(16 * MSM_MACS_PER_ADDITION, n22, ker)}
            extra["msm_2^22"]["roofline_valu"]["note"] = (
                "one 7-product mixed addition per point and window, 16 windows (16-bit windows from 3 x 2^20 points: the call does "
                "16/18 of the additions the 14-bit windows of earlier rounds did, so the same time reads as a lower fraction); whole call")
            # beyond the advertised size: 2^23 and 2^24 points (fresh points: Elligator images, decoded once outside the timing)
            for lg in (23, 24):
                nl = 1 << lg
                gl = torch.Generator(device=dev).manual_seed(2400 + lg)
                rl = torch.randint(0, 256, (nl, 32), dtype=torch.uint8, device=dev, generator=gl)
                kl = torch.randint(0, 256, (nl, 32), dtype=torch.uint8, device=dev, generator=gl)
                pl, _ = ctx.decompress(ctx.encode_to_curve(rl))
                del rl
                ker, _ = time_op(torch, lambda: ctx.msm(pl, kl), 3, 1)
                extra["msm_2^%d" % lg] = {"n": nl, "ms": ker, "per_sec": nl / (ker * 1e-3),
                                          "roofline_valu": valu_view(16 * MSM_MACS_PER_ADDITION, nl, ker)}
                del pl, kl
                torch.cuda.empty_cache()
            o22 = torch.empty((n22, 32), dtype=torch.uint8, device=dev)
            ker, _ = time_op(torch, lambda: ctx.scalar_mul_base(scalars[:n22], outs=[o22]), 3, 1)
            extra["scalar_mul_base_2^22"] = {"n": n22, "kernel_ms": ker, "per_sec": n22 / (ker * 1e-3),
                                             # wide launch beyond 2^20 elements: one inversion per 16 elements, not 8
                                             "roofline_valu": valu_view(KERNEL_MACS["scalar_mul_base"] - DIVSTEP_MACS_PER_INVERSION / 16.0, n22, ker)}
            del pm22, o22
        # small batches: a call lasts as long as one element's dependency chain (one quad of lanes per element / point)
        ns = 1 << 12
        small = {}
        for name, fn in [
            ("msm", lambda: ctx.msm(pm[:ns], scalars[:ns])),
            ("msm_encoded", lambda: ctx.msm(enc1[:ns], scalars[:ns])),
            ("scalar_mul_var", lambda: ctx.scalar_mul_var(enc1[:ns], scalars[:ns], outs=[o1[:ns], s1[:ns]])),
            ("scalar_mul_var_element", lambda: ctx.scalar_mul_var_element(pm[:ns], scalars[:ns])),
            ("scalar_mul_base", lambda: ctx.scalar_mul_base(scalars[:ns], outs=[o1[:ns]])),
        ]:
            ker, _ = time_op(torch, fn, 10, 10)       # ten untimed calls first: a call this short is otherwise timed at the clock the previous route left behind
            small[name] = ker
        extra["small_batch_2^12_ms_per_call"] = small
        # the smallest batches (up to one element per SIMD: 4 x the CUs) take one WAVE per element / point, lane-spread arithmetic
        nt = 1 << 8
        tiny = {}
        for name, fn in [
            ("msm", lambda: ctx.msm(pm[:nt], scalars[:nt])),
            ("msm_encoded", lambda: ctx.msm(enc1[:nt], scalars[:nt])),
            ("scalar_mul_var", lambda: ctx.scalar_mul_var(enc1[:nt], scalars[:nt], outs=[o1[:nt], s1[:nt]])),
            ("scalar_mul_var_element", lambda: ctx.scalar_mul_var_element(pm[:nt], scalars[:nt])),
            # one scalar per wave; the square-root family: four elements per wave (up to 16 x the CUs)
            ("scalar_mul_base", lambda: ctx.scalar_mul_base(scalars[:nt], outs=[o1[:nt]])),
            ("sqrt_ratio_zeta", lambda: ctx.sqrt_ratio_zeta(r0[:nt], scalars[:nt], outs=[o1[:nt], s1[:nt]])),
            ("decompress", lambda: ctx.decompress(enc1[:nt])),
            ("compress", lambda: ctx.compress(pm[:nt], outs=[o1[:nt]])),
            ("encode_to_curve", lambda: ctx.encode_to_curve(r0[:nt], outs=[o1[:nt]])),
            ("hash_to_curve", lambda: ctx.hash_to_curve(r0[:nt], scalars[:nt], outs=[o1[:nt]])),
        ]:
            ker, _ = time_op(torch, fn, 10, 10)
            tiny[name] = ker
        extra["tiny_batch_2^8_ms_per_call"] = tiny
        # mid-size MSMs (what a batch verifier holds): whole call
        mid = {}
        for lg in (16, 18):
            if ne >= (1 << lg):
                ker, _ = time_op(torch, lambda: ctx.msm(pm[:1 << lg], scalars[:1 << lg]), 10, 10)
                mid["2^%d" % lg] = ker
        extra["msm_mid_ms_per_call"] = mid
        aff = torch.empty((ne, 8), dtype=torch.int64, device=dev)
        ker, _ = time_op(torch, lambda: ctx.to_affine(pm, outs=[aff]), 5, 5)
        extra["to_affine"] = {"n": ne, "kernel_ms": ker, "per_sec": ne / (ker * 1e-3)}
        # `Element * Fr` with the reference's own signature (Elements in and out: no square root at either end) and
        # Fr products on 32-byte scalars (the one HBM-priced op here: 96 algorithmic bytes per product)
        pm2 = torch.empty_like(pm)
        ker, _ = time_op(torch, lambda: ctx.scalar_mul_var_element(pm, scalars[:ne], outs=[pm2]), 5, 5)
        extra["scalar_mul_var_element"] = {"n": ne, "kernel_ms": ker, "per_sec": ne / (ker * 1e-3)}
        # many small multiscalar sums at once (d377_batch_msm_small): 2^20 (or n) independent 3-term sums, the shape of the
        # reference's own multiscalar test (tests/operations.rs:44-60), against the composition the call replaces
        g3 = torch.Generator(device=dev).manual_seed(9000 + rank)
        r3 = torch.randint(0, 256, (3 * ne, 32), dtype=torch.uint8, device=dev, generator=g3)
        k3 = torch.randint(0, 256, (3 * ne, 32), dtype=torch.uint8, device=dev, generator=g3)
        p3, _ = ctx.decompress(ctx.encode_to_curve(r3))
        o3 = torch.empty((ne, 32), dtype=torch.uint8, device=dev)
        ker, _ = time_op(torch, lambda: ctx.msm_small(p3, k3, 3, outs=[o3]), 3, 2)
        cols = [p3[j::3].contiguous() for j in range(3)]
        kcols = [k3[j::3].contiguous() for j in range(3)]

        def composed3():
            acc = ctx.scalar_mul_var_element(cols[0], kcols[0])
            for j in (1, 2):
                acc = ctx.add(acc, ctx.scalar_mul_var_element(cols[j], kcols[j]))
            return ctx.compress(acc)

        ker_c, _ = time_op(torch, composed3, 2, 1)
        same3 = bool(torch.equal(composed3(), ctx.msm_small(p3, k3, 3)))
        extra["msm_small_3_terms"] = {"sums": ne, "terms": 3, "ms": ker, "sums_per_sec": ne / (ker * 1e-3), "terms_per_sec": 3 * ne / (ker * 1e-3),
                                      "composition_ms": ker_c, "composition_over_msm_small": ker_c / ker, "equal_to_composition": same3,
                                      "note": "one Straus chain per sum; composition = 3 x scalar_mul_var_element + 2 x add + compress"}
        del r3, p3, cols
        ker, _ = time_op(torch, lambda: ctx.fr_op("mul", scalars[:ne], r0[:ne], outs=[o1, s1]), 5, 5)
        extra["fr_mul"] = {"n": ne, "kernel_ms": ker, "per_sec": ne / (ker * 1e-3), "algo_GBps": 97 * ne / (ker * 1e-3) / 1e9}
        # the remaining group-level entry points of the path, for the record (same 2^20 records)
        xy = torch.empty((ne, 16), dtype=torch.int64, device=dev)
        for name, fn in [
            ("decompress", lambda: ctx.decompress(enc1, outs=[xy, s1])),
            ("compress", lambda: ctx.compress(pm, outs=[o1])),
            ("hash_to_curve", lambda: ctx.hash_to_curve(r0[:ne], scalars[:ne], outs=[o1])),
        ]:
            ker, _ = time_op(torch, fn, 5, 10)                      # ten untimed calls: they follow the HBM-priced Fr products
            macs = KERNEL_MACS[name]
            if name in ("decompress", "compress") and ne >= chunked_min:
                macs = KERNEL_MACS[name + "_chunked"]               # the route the entry point takes at this size
            extra[name] = {"n": ne, "kernel_ms": ker, "per_sec": ne / (ker * 1e-3),
                           "roofline_valu": valu_view(macs, ne, ker)}
        extra["encodes_per_sec"] = extra["roundtrip"]["per_sec_all_gpus"]          # whole job, all GPUs
        extra["elligator_encodes_per_sec"] = extra["encode_to_curve"]["per_sec_all_gpus"]
        line["extra"] = extra

    # The CPU leg runs on rank 0 at every N (the other ranks wait at the barrier below): the same host cores, the same run.
    cpu_leg_error = None
    if rank == 0 and not args.no_cpu_baseline and head_local and int(head_local[0].shape[0]) >= 4096:
        try:
            orc = Oracle(native=True, build=False)
            cores = usable_cores()
            # pilot on one thread to size a sample worth ~12 s of wall time on all cores
            pilot = 512
            n_head = int(head_local[0].shape[0])
            p_h = head_local[0].cpu().numpy()            # rank 0's last headline launch: inputs ...
            k_h = head_local[1].cpu().numpy()
            out, n = head_local[2], n_head               # ... and outputs, compared below on the sample
            t0 = time.perf_counter()
            orc.run_threads("scalar_mul_var", p_h[:pilot], k_h[:pilot], 1)
            per_thread = pilot / (time.perf_counter() - t0)
            one_thread_rate = per_thread
            ns = int(min(n, max(4096, per_thread * cores * 12.0)))
            p_h, k_h = p_h[:ns], k_h[:ns]
            t0 = time.perf_counter()
            o_out, o_st, used = orc.run_threads("scalar_mul_var", p_h, k_h, cores)
            dt = time.perf_counter() - t0
            same = bool((o_out == out[:ns].cpu().numpy()).all() and not o_st.any())
            line["cpu_baseline"] = {
                "value": ns / dt,
                "unit": "scalar-mults/s",
                "cores": used,
                "kind": "port",
                "sample": "first %d of the %d (point, scalar) pairs of this run, %d pthreads over contiguous slices; "
                          "C restatement of the reference algorithm (Sarkar sqrt, 256-step double-and-add, 4x64 "
                          "Montgomery), gcc %s" % (ns, n, used, orc.flags),
                "seconds": dt,
                "matches_gpu_output": same,
                "value_1_thread": one_thread_rate,
            }
            if not same:
                raise AssertionError("GPU output differs from the oracle on the cpu_baseline sample")
            # BASELINE.json configs[0] (the reference's own CPU-runnable case, shape of benches/sqrt.rs):
            # 2^16 Fq::sqrt_ratio_zeta on the CPU restatement, one thread, plus the GPU on the same pairs
            nsq = min(1 << 16, int(r0.shape[0]))
            num_h, den_h = r0[:nsq].cpu().numpy(), scalars[:nsq].cpu().numpy()
            t0 = time.perf_counter()
            o_root, o_ws, _ = orc.run_threads("sqrt_ratio_zeta", num_h, den_h, 1)
            dts = time.perf_counter() - t0
            g_root, g_ws = ctx.sqrt_ratio_zeta(r0[:nsq], scalars[:nsq])
            ok = bool((g_root.cpu().numpy() == o_root).all() and (g_ws.cpu().numpy() == o_ws).all())
            if not ok:
                raise AssertionError("GPU sqrt_ratio_zeta differs from the oracle")
            line["cpu_baseline"]["config0_sqrt_ratio_zeta_2^16"] = {
                "cpu_ns_per_call_1_thread": dts / nsq * 1e9, "cpu_per_sec_1_thread": nsq / dts, "matches_gpu_output": ok}
        except Exception as e:                      # every rank must still reach the barrier below: report, then fail
            cpu_leg_error = "%s: %s" % (type(e).__name__, e)
            line["cpu_baseline_error"] = cpu_leg_error
            parity_ok = False

    if dist.is_initialized() and world > 1:
        # rank 0 may have spent ~15 s in the CPU leg; its verdict (a failed check there fails the job on every rank)
        # travels with the MIN all-reduce, which is also the barrier
        flag = torch.tensor([1 if parity_ok else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity_ok = bool(flag.item())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if cpu_leg_error:
        raise SystemExit("bench.py: the CPU-baseline leg failed: " + cpu_leg_error)
    if not parity_ok:
        raise SystemExit("bench.py: a rank's outputs differ from the oracle on its parity sample (or rank 0's CPU leg failed)")


if __name__ == "__main__":
    main()
