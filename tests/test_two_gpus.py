"""The multi-rank RCCL paths with MORE THAN ONE rank: skipped on a one-GPU box -- the round's GPU box has one
MI355X, so there the collectives only ever run with one forced rank (tests/test_frames_comm.py) -- and run wherever two
devices are visible: slam3d_icp_dense_run over a 2-rank communicator must be bit-identical to the unsharded run on every
rank (both exchange forms), and slam3d_pose_gather must return the table in rank order."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs (RCCL wants one device per rank)")
def test_dense_run_and_pose_gather_over_two_ranks():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_two_gpu_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0 and "TWO_GPU_RESULT True" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
