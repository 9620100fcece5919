"""Worker for tests/test_distributed_cpu.py: one rank of a gloo group on CPU.
Exercises the exact sharding + logits all-gather code bench.py uses with RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["I8IE_NO_TORCH_PRELOAD"] = "1"
from int8inferenceengine_amd import sharding  # noqa: E402


def main():
    n_total = int(sys.argv[1])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    start, stop = sharding.shard_bounds(n_total, rank, world)
    # "logits" every rank can recompute: row i = [i*10 + j]
    full = (np.arange(n_total, dtype=np.float32)[:, None] * 10 + np.arange(10, dtype=np.float32)[None, :])
    local = torch.from_numpy(full[start:stop].copy())
    got = sharding.gather_rows(local, n_total).numpy()
    assert got.shape == (n_total, 10), got.shape
    assert np.array_equal(got, full), "rank %d: gathered logits differ" % rank
    # timing reduction used by bench.py: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    # labels gather (single column) and the top-1 count on the gathered result
    lab = torch.from_numpy((np.arange(start, stop) % 10).astype(np.float32)[:, None])
    labs = sharding.gather_rows(lab, n_total).numpy()[:, 0].astype(np.int64)
    assert np.array_equal(labs, np.arange(n_total) % 10)
    pred = sharding.centred_argmax(got, np.zeros(10, np.float32))
    assert (pred == 9).all()
    # the package's runner (host mode): a "network" every rank can recompute, pipelined two deep like bench.py
    def net(x):
        return x @ np.arange(40, dtype=np.float32).reshape(4, 10) + 1.0

    runner = sharding.ShardedRunner(net, n_total, 10, host_copies=True)
    assert (runner.start, runner.stop) == (start, stop) and runner.rows == stop - start
    batches = [np.random.default_rng(s).normal(size=(n_total, 4)).astype(np.float32) for s in range(3)]
    pending, outs = None, []
    for xb in batches:
        h = runner.submit(runner.shard(xb))
        if pending is not None:
            outs.append(runner.result(pending))
        pending = h
    outs.append(runner.result(pending))
    for xb, o in zip(batches, outs):
        if rank == 0:
            assert o.shape == (n_total, 10) and np.array_equal(o, net(xb)), "runner: gathered logits differ"
        else:
            assert o is None
    assert (runner.run(runner.shard(batches[0])) is None) == (rank != 0)
    # the depth contract is enforced: a third outstanding batch at depth 2 is refused BEFORE it issues a collective
    # (every rank refuses alike, so the group stays matched), a collected handle cannot be read twice, and a deeper
    # ring (depth 3) takes three outstanding batches and returns each one's own logits
    h0 = runner.submit(runner.shard(batches[0]))
    h1 = runner.submit(runner.shard(batches[1]))
    try:
        runner.submit(runner.shard(batches[2]))
        raise AssertionError("a third outstanding batch at depth 2 must be refused")
    except RuntimeError as e:
        assert "outstanding" in str(e)
    r0, r1 = runner.result(h0), runner.result(h1)
    if rank == 0:
        assert np.array_equal(r0, net(batches[0])) and np.array_equal(r1, net(batches[1]))
    try:
        runner.result(h0)
        raise AssertionError("a handle must not be readable twice")
    except RuntimeError as e:
        assert "stale handle" in str(e)
    deep = sharding.ShardedRunner(net, n_total, 10, host_copies=True, depth=3)
    hs = [deep.submit(deep.shard(xb)) for xb in batches]
    for xb, h in zip(batches, hs):
        o = deep.result(h)
        assert (o is None) if rank != 0 else np.array_equal(o, net(xb))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d ok rows [%d,%d)" % (rank, world, start, stop))


if __name__ == "__main__":
    main()
