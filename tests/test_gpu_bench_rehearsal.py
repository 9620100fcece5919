"""bench.py's N > 1 control flow (one process per GPU, sharded batch, logits all-gather, depth-2 pipeline,
matched collectives in every pass) rehearsed with two ranks on ONE GPU: RCCL refuses two ranks on a device, so
I8IE_BENCH_REHEARSE=1 maps every rank to device 0 and runs the collectives over gloo on host copies.  The
gathered logits must score exactly like a single-process run of the same global batch."""
import json
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


def _last_json(text):
    return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_equal_one_rank():
    common = ["bench.py", "--steps", "3", "--warmup", "1", "--batch", "66", "--no-cpu-baseline"]
    env = dict(os.environ, I8IE_BENCH_NO_PREWARM="1")
    one = subprocess.run([sys.executable] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + common + ["--gpus", "2"],
                         cwd=ROOT, env=dict(env, I8IE_BENCH_REHEARSE="1"), capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["config"]["per_gpu_batch"] == 33
    assert b["scaling"] == "strong" and b["config"]["global_batch"] == 66
    assert a["top1_vs_fp32_teacher"] == b["top1_vs_fp32_teacher"]
    assert "REHEARSAL" in b and b["value"] > 0


def test_bench_starts_its_own_ranks_when_typed_without_a_launcher():
    """`python bench.py --gpus 2 ...` with no launcher: bench.py starts torch.distributed.run as a child process (before it
    imports torch or touches HIP), relays rank 0's ONE JSON line and the child's exit code."""
    env = dict(os.environ, I8IE_BENCH_NO_PREWARM="1", I8IE_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "66", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["per_gpu_batch"] == 33 and b["steps"] == 3 and b["value"] > 0
