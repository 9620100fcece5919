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
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d ok rows [%d,%d)" % (rank, world, start, stop))


if __name__ == "__main__":
    main()
