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
    assert "PERSISTENT-OK" in res.stdout, res.stdout[-2000:]


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


def _build_c_demo(tmp_path, rccl):
    exe = str(tmp_path / ("pcg_sharded_demo" + ("_rccl" if rccl else "")))
    libdir = os.path.join(ROOT, "cosmomap2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__"]
                          + (["-DUSE_RCCL"] if rccl else [])
                          + ["-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                             os.path.join(ROOT, "tests", "c_abi", "pcg_sharded_demo.c"),
                             "-L" + libdir, "-lcosmomap2_hip", "-L/opt/rocm/lib", "-lamdhip64"]
                          + (["-lrccl"] if rccl else [])
                          + ["-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("layout", [0, 1])
def test_c_host_shards_over_two_processes(tmp_path, layout):
    """A C99 host on two ranks (tests/c_abi/pcg_sharded_demo.c): the TOD cut at a noise-block boundary,
    cm2_pcg_sharded with the host's collective as callbacks -- here a file-backed exchange between the
    two processes that share the box's GPU --, replicated (0) and row-sharded (1) map vectors.  Every
    rank checks iteration count and solution against its own un-sharded cm2_pcg solve."""
    exe = _build_c_demo(tmp_path, rccl=False)
    n = 3 * 12 * 16 * 16
    xch = str(tmp_path / "exchange")
    with open(xch, "wb") as f:
        f.write(b"\0" * (64 + 8 * 2 * n))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([exe, str(r), "2", str(layout), xch], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
        assert "C-SHARDED-OK" in so and ("rows" if layout else "replicated") in so, so


@pytest.mark.parametrize("layout", [0, 1])
def test_c_host_collectives_through_rccl(tmp_path, layout):
    """The same program with its collectives compiled against rccl.h (ncclAllReduce, ncclAllGather,
    ncclReduceScatter on the solve's stream), one rank: the call path a production C host uses."""
    exe = _build_c_demo(tmp_path, rccl=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([exe, "0", "1", str(layout), "-"], env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "C-SHARDED-OK" in res.stdout and "RCCL" in res.stdout, res.stdout
