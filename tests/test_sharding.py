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
