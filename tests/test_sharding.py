"""N > 1 path on CPU: world_size-2 gloo run of the shard / scatter / gather / max-timing
plumbing (decaf377_amd/sharding.py), plus in-process checks of the partition."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    from decaf377_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 9, 1 << 22, (1 << 22) + 5):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = shard_bounds(n, world, r)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == n and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_scatter_map_gather(world):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST_OK world=%d" % world in r.stdout


def test_bench_self_launch_plumbing():
    """`python bench.py --gpus N` with no launcher starts its own N ranks (the parent makes no GPU call); the ranks
    rendezvous and rank 0's line comes back through the parent.  --launch-check keeps the children off the GPU."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launch_check": True, "ranks_seen": 3, "rank_sum": 3, "self_launched": True}
    # the same command under the driver's launcher: bench.py must not start ranks of its own
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--launch-check"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen"] == 2 and line["self_launched"] is False


def test_bench_self_launch_propagates_failure():
    """A rank that dies (here: no GPU in the CPU container, or --gpus above the GPUs present) ends the whole job with a
    non-zero exit code instead of leaving the other ranks waiting in a collective."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log2n", "10",
                        "--no-cpu-baseline", "--no-extra"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
