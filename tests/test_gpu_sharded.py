"""
Two ranks, TOD sharded at noise-block boundaries, on the one GPU of the test box (gloo
transport; RCCL needs one GPU per rank): the reduced matvec with the chunked, overlapped
all-reduce against the single all-reduce and against the same problem on one rank, and a
PCG solve with the residual norm synchronised across ranks.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port nobody listens on right now (a fixed number collides with a lingering rendezvous or a
    parallel job on the same box)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_one_gpu():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_sharded_worker.py")]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "SHARDED-OK" in res.stdout, res.stdout[-2000:]
    assert "ROWSHARDED-OK" in res.stdout, res.stdout[-2000:]


def test_one_rank_rccl_group():
    """The "nccl" (RCCL) backend itself, with one rank on the one GPU of the test box: every
    collective of sharding.py is forced through the transport (tests/_nccl_worker.py) and the
    results must be those of the single-process run, bit for bit."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_worker.py")],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "RCCL-1RANK-OK" in res.stdout, res.stdout[-2000:]
