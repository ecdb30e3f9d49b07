"""N > 1 on the MI355X: two data-parallel ranks (fresh child processes, gloo, both on the one GPU of the box)
running the HIP kernels, the bucketed reduction overlapped with backward and the per-bucket fused AdamW, against
one process that accumulates both batches (tools/check_ddp_gpu.py).  RCCL with more than one rank needs more than one
GPU (the driver's multi-GPU bench); its calls and stream semantics are exercised here in a ONE-rank nccl group with the
bucket collectives forced (tools/check_rccl_one_rank.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_process_on_both_batches():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "check_ddp_gpu.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert "ok" in r.stdout.splitlines()[-1], r.stdout[-2000:]


@pytest.mark.timeout(600)
def test_rccl_one_rank_group_leaves_training_unchanged():
    """nccl backend (RCCL) on this box's one GPU: communicator creation, async bucket all-reduces (fp32 and bf16
    buckets) under backward, the side stream's waits, per-bucket AdamW -- three steps equal to the trainer without a
    process group."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    env.pop("VY_DDP_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rccl_one_rank.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1] == "ok", r.stdout[-2000:]
