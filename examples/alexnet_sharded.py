#!/usr/bin/env python3
"""AlexNet INT8 over the GPUs of one node: one process per GPU, the batch sharded across the ranks, one all-gather
of the logits (int8inferenceengine_amd.sharding.ShardedRunner -- the same runner bench.py times).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 \\
        examples/alexnet_sharded.py --batch 1000 --batches 20

The reference's batch loop is an OpenMP `parallel for` over independent images (src/conv2d.cc:125); here each rank
takes a contiguous slice of the batch and holds a full replica of the quantised weights.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1000, help="global batch")
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--graph", action="store_true", help="replay this rank's forward as one HIP graph (small shards)")
    ap.add_argument("--check", action="store_true", help="rank 0: compare the gathered logits with an unsharded forward")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    import int8inferenceengine_amd  # noqa: F401
    import _CXX_i8ie as cx
    import i8ie
    from int8inferenceengine_amd import sharding
    from int8inferenceengine_amd import workloads as wl

    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL on ROCm
    torch.cuda.set_stream(torch.cuda.Stream())  # not the legacy default stream (no graph capture there, implicit syncs)
    cx.use_stream(torch.cuda.current_stream().cuda_stream, local_rank)  # engine kernels on torch's stream

    net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet"))  # identical on every rank (seeded)
    runner = sharding.ShardedRunner(net, args.batch, 10)
    x_all = wl.synthetic_input("alexnet", args.batch, seed=1234)
    x_dev = i8ie.tensor(np.ascontiguousarray(runner.shard(x_all))).prefetch()
    if args.graph:
        from int8inferenceengine_amd.graph import GraphedForward

        graphed = GraphedForward(net, x_dev)
        runner.forward = lambda x: graphed()

    runner.run(x_dev)  # warm-up
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    pending = None
    for _ in range(args.batches):  # depth-2 pipeline: batch i + 1 is queued before batch i's logits are awaited
        h = runner.submit(x_dev)
        if pending is not None:
            logits = runner.result(pending)
        pending = h
    logits = runner.result(pending)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print("%d GPU(s): %.0f images/s, logits %s, classes hit %d" % (world, args.batch * args.batches / dt, logits.shape,
                                                                      len(np.unique(logits.argmax(1)))))
    if args.check and rank == 0:
        want = net(i8ie.tensor(x_all)).numpy()  # the whole batch on one GPU, eagerly
        assert np.array_equal(logits.view(np.uint32), want.view(np.uint32)), "sharded logits differ from the unsharded forward"
        print("check ok: gathered logits identical to the unsharded forward")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
