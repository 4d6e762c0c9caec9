"""Multi-GPU legs that need MORE THAN ONE GPU on the box (skipped on the one-GPU boxes the round's own runs get; they run wherever the
driver's GPU tier sees two or more devices): RCCL over xGMI for the gradient exchange of the training iteration."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(backend, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", S2D_TEST_BACKEND=backend)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_overlap_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    ok = [l for l in r.stdout.splitlines() if l.startswith("RCCL_OVERLAP_OK")]
    assert ok, r.stdout[-2000:]
    print(ok[0])
    return ok[0]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_overlapped_allreduce_two_ranks_rccl_matches_one_shot_exchange():
    """forward_backward(grad_ready=ex.ready) + finish() on two ranks over RCCL, each child process initialising its own GPU,
    against optimizer.allreduce_grads() on the same gradients, bit for bit (tests/_rccl_overlap_worker.py)"""
    assert "async=True" in _run("nccl", 29671)


def test_overlapped_allreduce_worker_two_ranks_share_the_gpu_gloo():
    """the same worker with both ranks on this GPU over gloo (the exchange then happens in finish(): no overlap, same result) -- keeps
    the worker itself exercised on one-GPU boxes"""
    _run("gloo", 29672)
