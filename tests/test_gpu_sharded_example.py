"""examples/alexnet_sharded.py end to end on one GPU: one rank, backend "nccl" (RCCL), ShardedRunner in DEVICE mode
(stage buffers, side stream, all-gather of device tensors, pinned read-back on rank 0), eager and with the forward
replayed as a HIP graph; the gathered logits must equal an unsharded forward bit for bit.  (Two ranks on one device
are refused by RCCL: the two-rank control flow is rehearsed over gloo in test_gpu_bench_rehearsal.py and on the CPU
in test_distributed_cpu.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra", [[], ["--graph"]])
def test_sharded_example_single_rank_rccl(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "examples/alexnet_sharded.py", "--batch", "96", "--batches", "4", "--check"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "check ok" in r.stdout and "1 GPU(s)" in r.stdout, r.stdout[-2000:]
