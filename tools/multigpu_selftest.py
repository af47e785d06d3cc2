#!/usr/bin/env python3
"""First contact with a multi-GPU node: every multi-device path of the engine, exit code 0 only if all of it is right.

    python tools/multigpu_selftest.py [--gpus G] [--log2n 16] [--require-distinct]

This launcher never touches a GPU itself (a process that has opened the GPU must not start programs on this pool):
it only starts the legs below as child processes and checks their exit codes.

  leg 1  tests/multigpu_worker.py   ONE process, one d377_ctx over all G devices: peer access, per-device tables,
                                    d377_batch_sharded_dev rooted on each device, the host path sliced over the devices,
                                    multi-device d377_msm -- all against the oracle / the single-device results
  leg 2  tests/dist_worker_gpu.py   G processes, one per GPU, backend nccl (= RCCL over xGMI): scatter of input records,
                                    gather of outputs, all-gather of MSM partial sums, MAX all-reduce of the timings
  leg 3  bench.py --gpus G          the driver's command at a small size, self-launched; the line must report
                                    ranks_seen == G and parity_sample_ok

With one GPU visible: leg 1 lists GPU 0 twice, leg 2 runs one rank, leg 3 runs two ranks on GPU 0 over gloo."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    # asked of a short-lived child, so that this process stays GPU-free whatever the runtime does to count devices
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                       timeout=600)
    return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(script, world, extra_env=None, timeout=900):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script], env=env, cwd=ROOT))
    t0 = time.time()
    rc = 0
    live = set(range(world))
    while live:
        for r in list(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    for o in live:
                        procs[o].terminate()
        if time.time() - t0 > timeout:
            rc = rc or 124
            for o in live:
                procs[o].kill()
            break
        time.sleep(0.1)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="GPUs to use (default: all visible)")
    ap.add_argument("--log2n", type=int, default=16)
    ap.add_argument("--require-distinct", action="store_true",
                    help="exit non-zero unless every leg ran on >= 2 physical GPUs (a one-GPU run lists GPU 0 twice and "
                         "covers slicing, staging and ordering, but no peer copy, no cross-device event, no RCCL transfer)")
    args = ap.parse_args()
    G = args.gpus or gpu_count()
    if G < 1:
        print("multigpu_selftest: no GPU visible")
        return 2
    print("multigpu_selftest: %d GPU(s)" % G, flush=True)
    if args.require_distinct and G < 2:
        print("multigpu_selftest: --require-distinct: only %d GPU visible -- the distinct-device paths cannot be covered here" % G)
        return 3
    devs = ",".join(str(g) for g in range(G)) if G > 1 else "0,0"

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multigpu_worker.py"), "--log2n", str(args.log2n), "--devices", devs]
                       + (["--require-distinct"] if args.require_distinct else []), cwd=ROOT, timeout=1500)
    if r.returncode != 0:
        print("multigpu_selftest: leg 1 (one context over %s) FAILED rc=%d" % (devs, r.returncode))
        return 1
    print("leg 1 ok", flush=True)

    rc = run_ranks(os.path.join(ROOT, "tests", "dist_worker_gpu.py"), G)
    if rc != 0:
        print("multigpu_selftest: leg 2 (%d RCCL ranks) FAILED rc=%d" % (G, rc))
        return 1
    print("leg 2 ok", flush=True)

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log2n", str(args.log2n),
           "--no-cpu-baseline", "--no-extra"]
    cmd += ["--gpus", str(G)] if G > 1 else ["--gpus", "2", "--same-device", "--backend", "gloo"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        print("multigpu_selftest: leg 3 (bench.py) FAILED rc=%d\n%s\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-3000:]))
        return 1
    line = json.loads(lines[0])
    want = G if G > 1 else 2
    if line.get("ranks_seen") != want or line.get("n_gpus") != want or not line.get("parity_sample_ok"):
        print("multigpu_selftest: leg 3 line is wrong: %s" % lines[0])
        return 1
    print("leg 3 ok: %d ranks, %.3g scalar-mults/s, parity ok" % (want, line["value"]), flush=True)
    print("MULTIGPU_SELFTEST_OK gpus=%d distinct_device_paths=%s" % (G, "covered" if G > 1 else "NOT covered (one GPU listed twice)"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
