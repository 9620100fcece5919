"""Batch sharding across the GPUs of one node, one process per GPU.

The reference is single-process CPU code; its batch loop is an OpenMP
`parallel for` over independent images (src/conv2d.cc:125).  Images stay
independent on the GPU, and every quantisation parameter is a per-tensor
constant fixed at convert(), so the batch shards with no data-path exchange.
The only collective is one all-gather of the per-shard logits ([n/G, 10] fp32,
5 KB per rank at n=1000, G=8: latency-bound, so a single direct all-gather, not
a bucketed ring schedule).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo"
runs the same code on CPU tensors (tests/test_distributed_cpu.py).
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous [start, stop) of rank's images; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_shard(n, world):
    return (n + world - 1) // world


def gather_rows(local, n_total, group=None):
    """All-gather row blocks of a [rows_local, C] torch tensor (CPU/gloo or GPU/RCCL) into
    [n_total, C] on every rank, in rank order.  Ragged shards are padded to the largest."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, stop = shard_bounds(n_total, rank, world)
    if local.shape[0] != stop - start:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], stop - start))
    cap = max_shard(n_total, world)
    cols = local.shape[1]
    send = local
    if local.shape[0] != cap:
        send = torch.zeros((cap, cols), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * cap, cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if n_total == world * cap:
        return recv
    parts = []
    for r in range(world):
        s, e = shard_bounds(n_total, r, world)
        parts.append(recv[r * cap: r * cap + (e - s)])
    return torch.cat(parts, 0)


def centred_argmax(logits, centre):
    """Top-1 on random-init weights: subtract a fixed per-class centre first, otherwise every
    image lands in the same class and agreement is vacuous."""
    return np.argmax(np.asarray(logits, np.float32) - np.asarray(centre, np.float32)[None, :], axis=1)
