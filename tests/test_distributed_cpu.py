"""N > 1 path on CPU: gloo, one process per rank, the same sharding/gather code
that runs over RCCL on the GPUs (int8inferenceengine_amd/sharding.py)."""
import os
import subprocess
import sys

import pytest

from int8inferenceengine_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_partition():
    for n in (1000, 1001, 7, 8, 3):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == sharding.max_shard(n, world)
    assert sharding.shard_bounds(1000, 7, 8) == (875, 1000)
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


@pytest.mark.parametrize("world,n_total", [(2, 1000), (3, 1000), (2, 7)])
def test_gather_over_gloo(world, n_total):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world * 10 + n_total % 7),
           os.path.join(HERE, "dist_worker.py"), str(n_total)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok rows ") == world


def test_failed_submit_releases_its_ring_slot():
    """A forward that raises inside submit() must not leave the slot claimed: the error a caller sees later is the
    forward's, not 'would overwrite the buffers of batch ...' (no process group: the forward raises before any gather)."""
    calls = {"n": 0}

    def forward(x):
        calls["n"] += 1
        raise ValueError("forward failed")

    r = sharding.ShardedRunner(forward, 8, 10, rank=0, world=1, host_copies=True, depth=2)
    for _ in range(5):  # more failures than slots: each one surfaces as itself
        with pytest.raises(ValueError, match="forward failed"):
            r.submit(None)
    assert calls["n"] == 5 and r._owner == [None, None] and r._tick == 0
