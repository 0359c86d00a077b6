"""
TOD sharding across the GPUs of one node (one process per GPU, RCCL over xGMI through
``torch.distributed``; backend "nccl" IS RCCL on ROCm, "gloo" in the CPU tests).

The reference is single-process; what makes sharding exact is the block
structure of N: ``P``, ``N^-1`` and ``P^T`` act on independent samples / independent
noise blocks (interfaces/blkop.py:195-206), so contiguous sample ranges cut AT BLOCK
BOUNDARIES can be processed by different ranks and only the pixel-domain result
couples them:

    A x = sum_ranks  P_k^T N_k^-1 P_k x          (one all-reduce of pol*npix doubles)

Map-domain vectors (x, r, p, z, Z) are replicated on every rank, so the two CG dot
products and the preconditioners need no communication; after the all-reduce every
rank holds bit-identical data and takes the same branches.  As a guard against a
non-bitwise-identical collective, :func:`make_sync` max-reduces the 8-byte ||r||^2 the
stop test reads.  Per PCG iteration the wire traffic is one all-reduce of the map
(18.9 MB at nside 256, 75.5 MB at nside 512).
"""
import numpy as np

from . import device as D
from . import linop as lp

torch = D.torch

__all__ = ["shard_blocks", "allreduce_sum_", "ShardedLO", "make_sync", "world"]


def world(group=None):
    """(rank, world_size) of the default / given process group, (0, 1) when
    torch.distributed is not initialised."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_blocks(block_sizes, world_size, rank):
    """
    Contiguous, block-aligned partition of the TOD: returns ``(b0, b1, s0, s1)`` -- rank
    ``rank`` owns noise blocks ``b0 <= b < b1`` = samples ``s0 <= t < s1``.  Cuts are placed
    at the block boundary nearest to the ideal equal-sample split, never inside a block
    (a Toeplitz block must not be split).  Ranks may get no block when there are fewer
    blocks than ranks.
    """
    sizes = np.asarray(block_sizes, dtype=np.int64)
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %r of %r" % (rank, world_size))
    off = np.concatenate([[0], np.cumsum(sizes)])
    total = int(off[-1])
    cuts = [0]
    for k in range(1, world_size):
        ideal = total * k / float(world_size)
        j = int(np.argmin(np.abs(off - ideal)))
        cuts.append(max(j, cuts[-1]))
    cuts.append(len(sizes))
    b0, b1 = cuts[rank], cuts[rank + 1]
    return b0, b1, int(off[b0]), int(off[b1])


def allreduce_sum_(t, group=None):
    """In-place sum over ranks of a tensor (HBM: RCCL; host: gloo).  No-op for one rank."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    if isinstance(t, np.ndarray):
        buf = torch.from_numpy(t)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def make_sync(group=None):
    """Callable for ``cg(..., sync=)``: every rank continues with the max over ranks of
    its ||r||^2, so that all ranks stop at the same iteration."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    backend = dist.get_backend(group)

    def sync(value):
        dev = D.dev() if backend == "nccl" else "cpu"
        buf = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        return float(buf.item())
    return sync


class ShardedLO(lp.LinearOperator):
    """
    ``sum over ranks of A_local``: wraps this rank's ``P_k^T N_k^-1 P_k`` (any symmetric
    map-domain operator) and all-reduces its output.  Vectors are replicated.
    """

    def __init__(self, local_op, group=None):
        self.local_op = local_op
        self.group = group
        n = local_op.shape[0]
        super(ShardedLO, self).__init__(n, n, self._mult, symmetric=True,
                                        device_ok=lp.supports_device(local_op))

    def _overlapped(self, x):
        """Tile-order chain with a fused noise operator: reduce tile groups while the next
        ones are still being back-projected (CM2_ALLREDUCE_CHUNKS groups, default 4; 0 or 1
        = one all-reduce after the matvec)."""
        import os
        dist = torch.distributed
        chunks = int(os.environ.get("CM2_ALLREDUCE_CHUNKS", "4"))
        if chunks <= 1 or not (dist.is_available() and dist.is_initialized()) \
                or dist.get_world_size(self.group) == 1:
            return None
        plan = getattr(self.local_op, "_compiled", None)
        ops = plan() if plan is not None else [self.local_op]
        if len(ops) != 1 or not hasattr(ops[0], "reduced_matvec"):
            return None
        group = self.group
        return ops[0].reduced_matvec(
            x, lambda view: dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True),
            chunks)

    def _mult(self, x):
        y = self._overlapped(x)
        if y is not None:
            return y
        y = self.local_op.matvec(x)
        if isinstance(y, np.ndarray):
            y = np.ascontiguousarray(y)
        elif not y.is_contiguous():
            y = y.contiguous()
        return allreduce_sum_(y, self.group)
