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


class ShardedRunner:
    """The N > 1 data path of the drop-in package: one process per GPU, this rank's contiguous shard of every batch
    through `forward`, ONE all-gather of the per-shard logits, the gathered [n_total, C] logits on rank 0.

    The reference's analogue is the OpenMP loop over independent images (src/conv2d.cc:125); nothing is exchanged
    between shards until the logits.  Two modes, same control flow:

    * device mode (default; `torch.distributed` backend "nccl" = RCCL over xGMI): `forward(x)` returns an `i8ie`
      tensor on the GPU.  Its logits are copied into one of two stage buffers on the compute stream; the all-gather
      and rank 0's copy to pinned host memory run on a SIDE stream behind an event, so the collective's latency and
      the wait for the slowest rank overlap the next batch's kernels.  A stage buffer is reused only after the
      gather that read it has signalled.  Make a `torch.cuda.Stream()` current (not the legacy default stream: it
      cannot be captured into a HIP graph and serialises with blocking streams) and call
      `_CXX_i8ie.use_stream(torch.cuda.current_stream().cuda_stream, dev)`
      before the first op so that the engine's kernels and these copies share torch's stream.
    * host mode (`host_copies=True`; backend "gloo"): `forward(x)` returns an ndarray (or an `i8ie` tensor that is
      read back); the gather runs on CPU tensors.  This is what the CPU tests and the one-GPU rehearsal use (RCCL
      refuses two ranks on one device).

    `submit(x)` queues one batch and returns a handle without waiting for the GPU; `result(handle)` returns the
    gathered logits as an ndarray on rank 0 (None elsewhere).  Submitting batch i + 1 before asking for batch i's
    result is the depth-2 pipeline bench.py times.  Every rank must make the same sequence of submit() calls
    (each contains a collective).

    `depth` (default 2) is the number of batches that may be outstanding (submitted, result() not yet taken): the
    stage / receive / pinned buffers are rings of that many slots.  A submit() that would reuse the slot of a batch
    whose result has not been collected raises RuntimeError instead of overwriting it, and result() refuses a handle
    whose slot has since been handed to another batch (on every rank, also where result() returns None).
    """

    def __init__(self, forward, n_total, n_classes, rank=None, world=None, host_copies=False, group=None, depth=2):
        import torch.distributed as dist

        self.forward = forward
        self.n_total = int(n_total)
        self.n_classes = int(n_classes)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.start, self.stop = shard_bounds(self.n_total, self.rank, self.world)
        self.host_copies = bool(host_copies)
        self.depth = int(depth)
        if self.depth < 1:
            raise ValueError("ShardedRunner: depth must be >= 1")
        self._tick = 0
        self._owner = [None] * self.depth  # tick of the uncollected batch that occupies slot k, or None
        self._stage = None

    # ---- which images are mine ---------------------------------------------------------------------------
    def shard(self, x_all):
        """This rank's rows of a global batch (any array-like indexed along axis 0)."""
        return x_all[self.start:self.stop]

    @property
    def rows(self):
        return self.stop - self.start

    # ---- device-mode plumbing ----------------------------------------------------------------------------
    def _device_setup(self):
        import torch

        self._torch = torch
        D = self.depth
        self._stage = [torch.empty((self.rows, self.n_classes), dtype=torch.float32, device="cuda") for _ in range(D)]
        self._side = torch.cuda.Stream()
        self._gathered = [None] * D
        # equal shards: the gather's receive buffers are allocated once (no allocator traffic per batch)
        self._even = self.n_total % self.world == 0
        self._recv = [torch.empty((self.n_total, self.n_classes), dtype=torch.float32, device="cuda") for _ in range(D)] if self._even else None
        self._host = None
        if self.rank == 0:
            self._host = [torch.empty((self.n_total, self.n_classes), dtype=torch.float32).pin_memory() for _ in range(D)]

    def _claim_slot(self):
        tick = self._tick
        k = tick % self.depth
        if self._owner[k] is not None:
            raise RuntimeError("ShardedRunner: batch %d would overwrite the buffers of batch %d, whose result() has not "
                               "been taken (depth=%d: at most that many batches may be outstanding)"
                               % (tick, self._owner[k], self.depth))
        self._owner[k] = tick
        self._tick += 1
        return k, tick

    def _release_slot(self, k, tick):
        if self._owner[k] != tick:
            raise RuntimeError("ShardedRunner: stale handle (batch %d): its result was already taken or its slot reused"
                               % tick)
        self._owner[k] = None

    def submit(self, x):
        # (claimed before the forward: a refused submit must not have issued this rank's part of a collective)
        k, tick = self._claim_slot()
        try:
            return self._submit_claimed(x, k, tick)
        except BaseException:
            # a forward or collective that raised must not leave the slot taken: the next submit on it would then fail with
            # the "would overwrite" error and hide this one
            if self._owner[k] == tick:
                self._owner[k] = None
                self._tick = tick
            raise

    def _submit_claimed(self, x, k, tick):
        y = self.forward(x)
        if self.host_copies:
            import torch

            local = y.numpy() if hasattr(y, "numpy") and not isinstance(y, np.ndarray) else np.asarray(y)
            full = gather_rows(torch.from_numpy(np.ascontiguousarray(local, np.float32)), self.n_total, self.group)
            return ("host", k, tick, full.numpy() if self.rank == 0 else None)
        if self._stage is None:
            self._device_setup()
        import _CXX_i8ie as cx

        torch = self._torch
        main = torch.cuda.current_stream()
        if self._gathered[k] is not None:
            main.wait_event(self._gathered[k])  # the gather of `depth` batches ago has finished reading stage[k]
        cx.copy_to_ptr(y.data, self._stage[k].data_ptr())
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._side):     # off the compute stream: the next batch does not wait for the collective
            self._side.wait_event(ready)
            if self._even:  # RCCL all-gather of the per-shard logits, straight into the preallocated buffer
                import torch.distributed as dist

                full = self._recv[k]
                dist.all_gather_into_tensor(full, self._stage[k], group=self.group)
            else:
                full = gather_rows(self._stage[k], self.n_total, self.group)
            if self.rank == 0:
                self._host[k].copy_(full, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._gathered[k] = ev
        return ("dev", k, tick, ev)

    def result(self, handle):
        kind, k, tick, payload = handle
        self._release_slot(k, tick)
        if kind == "host":
            return payload
        if self.rank != 0:
            return None
        payload.synchronize()
        return self._host[k].numpy().copy()  # (copied out: the pinned slot is free for the next batch)

    def run(self, x):
        """submit + result: one batch, no pipelining."""
        return self.result(self.submit(x))
