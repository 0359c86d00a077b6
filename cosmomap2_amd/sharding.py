"""
TOD sharding across the GPUs of one node (one process per GPU, RCCL over xGMI through
``torch.distributed``; backend "nccl" IS RCCL on ROCm, "gloo" in the CPU tests).

The reference is single-process; what makes sharding exact is the block
structure of N: ``P``, ``N^-1`` and ``P^T`` act on independent samples / independent
noise blocks (interfaces/blkop.py:195-206), so contiguous sample ranges cut AT BLOCK
BOUNDARIES can be processed by different ranks and only the pixel-domain result
couples them:

    A x = sum_ranks  P_k^T N_k^-1 P_k x          (one all-reduce of pol*npix doubles)

Two layouts of the map-domain vectors (x, r, p, z, Z):

* replicated (:class:`ShardedLO`): every rank holds whole vectors, so the two CG dot products
  and the preconditioners need no communication; after the all-reduce every rank holds
  bit-identical data and takes the same branches.  As a guard against a non-bitwise-identical
  collective, :func:`make_sync` max-reduces the 8-byte ||r||^2 the stop test reads.  Per PCG
  iteration the wire traffic is one all-reduce of the map (18.9 MB at nside 256, 75.5 MB at
  nside 512); every rank does the pixel-domain work of the whole map.
* row-sharded (:class:`RowShards`, :class:`RowShardedNormalLO`, :func:`row_sharded_bd`,
  :func:`row_sharded_two_level`; SURVEY 8e's preferred variant): rank k owns the contiguous,
  pixel-aligned rows [k n/N, (k+1) n/N) of every map vector and of Z / AZ.  One matvec =
  all-gather of p, local P^T N^-1 P, reduce-scatter of the result (the same bytes on the wire
  as the all-reduce); the dot products are local sums + an 8-byte all-reduce, Z^T r an r-vector
  all-reduce; M_BD, the vector updates and the passes over Z / AZ (3 n r 8 bytes per M2, 1.8 GB
  at nside 256, r = 32) shrink by the number of ranks.
"""
import os

import numpy as np

from . import device as D
from . import linop as lp

torch = D.torch

__all__ = ["shard_blocks", "allreduce_sum_", "ShardedLO", "make_sync", "world", "RowShards",
           "RowShardedNormalLO", "row_sharded_bd", "row_sharded_two_level"]


def _trivial(group=None):
    """True when a collective over `group` would be the identity and may be skipped: no process
    group, or one rank -- unless CM2_FORCE_COLLECTIVES is set (the one-rank RCCL test then sends
    every collective of this module through the transport)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not os.environ.get("CM2_FORCE_COLLECTIVES")


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
    if _trivial(group):
        return t
    if isinstance(t, np.ndarray):
        buf = torch.from_numpy(t)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _collective_max(value, group=None):
    """Max over ranks of a host integer (RCCL needs the buffer in HBM, gloo on the host)."""
    dist = torch.distributed
    if _trivial(group):
        return int(value)
    dev = D.dev() if dist.get_backend(group) == "nccl" else "cpu"
    buf = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
    return int(buf.item())


def make_sync(group=None, force=False):
    """Callable for ``cg(..., sync=)``: every rank continues with the max over ranks of
    its ||r||^2, so that all ranks stop at the same iteration.  ``None`` for a one-rank group
    unless ``force`` (the RCCL test drives the collective through a one-rank group).

    The callable has two forms.  ``sync(value)`` maps the host value (blocking all-reduce).
    ``sync.reduce_(t)`` max-reduces a one-element device tensor in place on the current stream
    WITHOUT waiting for it (RCCL only; ``None`` on gloo): the PCG driver folds it into its deferred
    read of ||r||^2, so the stop test costs no extra host synchronisation."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if _trivial(group) and not force:
        return None
    backend = dist.get_backend(group)

    def sync(value):
        dev = D.dev() if backend == "nccl" else "cpu"
        buf = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        return float(buf.item())

    def reduce_(t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t
    sync.reduce_ = reduce_ if backend == "nccl" else None
    return sync


class ShardedLO(lp.LinearOperator):
    """
    ``sum over ranks of A_local``: wraps this rank's ``P_k^T N_k^-1 P_k`` (any symmetric
    map-domain operator) and all-reduces its output.  Vectors are replicated.
    """

    def __init__(self, local_op, group=None, persistent_output=False):
        """``persistent_output``: device results are written into ONE buffer the operator keeps (no
        allocation per matvec, the same addresses for the all-reduce every time) and that buffer is what
        ``matvec`` returns -- valid until the next application.  The PCG and Arnoldi drivers of this
        package consume ``A p`` before they apply ``A`` again, so they may be given such an operator;
        a caller that keeps results across applications must not set it."""
        self.local_op = local_op
        self.group = group
        self.persistent_output = bool(persistent_output)
        self._out = None
        self._chunks = None              # collective choice, made at the first matvec
        self.collectives_issued = 0      # map all-reduces issued so far (tests compare ranks)
        n = local_op.shape[0]
        super(ShardedLO, self).__init__(n, n, self._mult, symmetric=True,
                                        device_ok=lp.supports_device(local_op))

    def _single(self, method):
        """The one compiled operator behind ``local_op`` if it has ``method`` (the tile-order chain's
        ``reduced_matvec`` / ``matvec_into``), else None."""
        plan = getattr(self.local_op, "_compiled", None)
        ops = plan() if plan is not None else [self.local_op]
        return ops[0] if len(ops) == 1 and hasattr(ops[0], method) else None

    def _out_buffer(self, x):
        if not (self.persistent_output and D.is_dev(x)):
            return None
        if self._out is None or self._out.device != x.device:
            self._out = torch.empty(self.shape[0], dtype=torch.float64, device=x.device)
        return self._out

    def allreduce_chunks(self, nloc=None):
        """Number of tile groups whose all-reduces are overlapped with the back-projection.  The
        choice is COLLECTIVE: every rank must issue the same number (and sizes) of collectives per
        matvec, so it is derived from the MAX over ranks of the per-rank sample count (one 8-byte
        all-reduce at the first matvec, cached), never from this rank's own shard -- uneven shards
        on either side of a threshold would otherwise disagree.  A chunk's all-reduce hides behind
        the back-projection of the next chunk: worth it while that takes longer than a small
        collective's latency (P^T of 1e8 samples: 0.4 ms) -> 4 groups from 4e7 samples per rank
        up, 2 from 1.5e7; a strongly scaled shard (1e7 samples: 0.05 ms) sends the map in one
        piece.  CM2_ALLREDUCE_CHUNKS overrides (it must be set identically on every rank)."""
        if os.environ.get("CM2_ALLREDUCE_CHUNKS"):
            return int(os.environ["CM2_ALLREDUCE_CHUNKS"])
        if self._chunks is None:
            if nloc is None:
                return None
            nmax = _collective_max(int(nloc), self.group)
            self._chunks = 4 if nmax >= 40_000_000 else (2 if nmax >= 15_000_000 else 1)
        return self._chunks

    def _overlapped(self, x):
        """Tile-order chain with a fused noise operator: reduce tile groups while the next
        ones are still being back-projected (:meth:`allreduce_chunks` groups; 0 or 1 = one
        all-reduce after the matvec).  With CM2_ALLREDUCE_CHUNKS set the chunked path is also
        taken by a one-rank group (the collectives are then trivial; used by the RCCL test)."""
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if _trivial(self.group) and not os.environ.get("CM2_ALLREDUCE_CHUNKS"):
            return None
        op = self._single("reduced_matvec")
        if op is None:
            return None
        chunks = self.allreduce_chunks(int(getattr(getattr(op, "P", None), "nrows", 0)))
        if chunks <= 1:
            return None
        group = self.group
        self.collectives_issued += chunks
        out = self._out_buffer(x)
        kw = {} if out is None else {"out": out}
        return op.reduced_matvec(
            x, lambda view: dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True),
            chunks, **kw)

    def _mult(self, x):
        y = self._overlapped(x)
        if y is not None:
            return y
        out = self._out_buffer(x)
        op = self._single("matvec_into") if out is not None else None
        if op is not None:
            y = op.matvec_into(x, out)
        else:
            y = self.local_op.matvec(x)
            if out is not None and D.is_dev(y):
                out.copy_(y)
                y = out
        if isinstance(y, np.ndarray):
            y = np.ascontiguousarray(y)
        elif not y.is_contiguous():
            y = y.contiguous()
        self.collectives_issued += 1
        return allreduce_sum_(y, self.group)


# ------------------------------------------------------------ row-sharded vectors -----
class RowShards(object):
    """
    Partition of the ``pol * npix`` map rows into one contiguous, pixel-aligned range per rank,
    all of the same length ``rows`` (the last ranks' ranges are padded with rows that stay 0), so
    that ``all_gather_into_tensor`` / ``reduce_scatter_tensor`` move equal pieces.  Works on
    HBM tensors (RCCL), CPU tensors and NumPy arrays (gloo).
    """

    def __init__(self, npix, pol, group=None):
        self.group = group
        self.rank, self.world = world(group)
        self.npix, self.pol = int(npix), int(pol)
        self.n = self.npix * self.pol
        self.pix_per_rank = -(-self.npix // self.world)
        self.rows = self.pix_per_rank * self.pol
        self.pix_lo = min(self.rank * self.pix_per_rank, self.npix)
        self.pix_hi = min(self.pix_lo + self.pix_per_rank, self.npix)
        self.lo, self.hi = self.pix_lo * self.pol, self.pix_hi * self.pol

    # -- helpers for the three vector kinds
    @staticmethod
    def _as_tensor(a):
        return torch.from_numpy(a) if isinstance(a, np.ndarray) else a

    def _like(self, a, length):
        if isinstance(a, np.ndarray):
            return np.zeros(length, dtype=np.float64)
        return torch.zeros(length, dtype=torch.float64, device=a.device)

    def local(self, full):
        """This rank's (padded) rows of a whole vector."""
        out = self._like(full, self.rows)
        out[:self.hi - self.lo] = full[self.lo:self.hi]
        return out

    def gather(self, loc):
        """Whole vector from every rank's rows (all-gather)."""
        if _trivial(self.group):
            return loc[:self.n] if len(loc) != self.n else loc
        full = self._like(loc, self.rows * self.world)
        torch.distributed.all_gather_into_tensor(self._as_tensor(full), self._as_tensor(
            loc if not isinstance(loc, np.ndarray) else np.ascontiguousarray(loc)), group=self.group)
        return full[:self.n]

    def reduce_scatter(self, full):
        """This rank's rows of the sum over ranks of whole vectors (reduce-scatter)."""
        if _trivial(self.group):
            return self.local(full)
        padded = self._like(full, self.rows * self.world)
        padded[:self.n] = full
        out = self._like(full, self.rows)
        torch.distributed.reduce_scatter_tensor(self._as_tensor(out), self._as_tensor(padded),
                                                op=torch.distributed.ReduceOp.SUM, group=self.group)
        return out

    # -- persistent exchange buffers of the matvec (device tensors only).  `gather` / `reduce_scatter`
    #    above return fresh memory and zero-fill / copy whole padded vectors: right for the odd call
    #    (collecting a solution), three map-sized passes plus allocator traffic when done per matvec.
    def exchange_buffers(self, device):
        """(gathered, to_sum): two vectors of ``rows * world`` doubles kept for the life of the object.
        ``gathered`` receives the all-gather (every rank's padded rows: the padding is 0 because every
        rank's padded rows are); the local operator writes its whole-map result into ``to_sum[:n]``,
        whose padding ``[n:]`` is zeroed once, here, and never written again."""
        bufs = getattr(self, "_xbuf", None)
        if bufs is None or bufs[0].device != device:
            g = torch.empty(self.rows * self.world, dtype=torch.float64, device=device)
            t = torch.zeros(self.rows * self.world, dtype=torch.float64, device=device)
            bufs = self._xbuf = (g, t)
        return bufs

    def gather_view(self, loc):
        """All-gather of the ranks' rows into the persistent buffer; returns its first ``n`` entries (a
        VIEW, valid until the next call).  One rank: ``loc`` itself."""
        if _trivial(self.group):
            return loc[:self.n] if len(loc) != self.n else loc
        g, _ = self.exchange_buffers(loc.device)
        torch.distributed.all_gather_into_tensor(g, loc if loc.is_contiguous() else loc.contiguous(),
                                                 group=self.group)
        return g[:self.n]

    def reduce_scatter_buffer(self, device, out=None):
        """This rank's rows of the sum over ranks of the ``to_sum`` buffers; ``out``: the (rows,) vector
        to receive them (default: a new one)."""
        _, t = self.exchange_buffers(device)
        if out is None:
            out = torch.empty(self.rows, dtype=torch.float64, device=device)
        if _trivial(self.group):
            out.zero_()
            out[:self.hi - self.lo] = t[self.lo:self.hi]
            return out
        torch.distributed.reduce_scatter_tensor(out, t, op=torch.distributed.ReduceOp.SUM, group=self.group)
        return out

    def allreduce_(self, t):
        """In-place sum over ranks (dot products, Z^T r)."""
        return allreduce_sum_(t, self.group)

    def bytes_per_matvec(self):
        """Bytes each rank sends per matvec (ring collectives): all-gather + reduce-scatter."""
        return 2 * (self.world - 1) * self.rows * 8


class RowShardedNormalLO(lp.LinearOperator):
    """``A`` on row-sharded vectors: all-gather, this rank's ``P_k^T N_k^-1 P_k``, reduce-scatter.
    Input and output are the rank's (padded) rows."""

    def __init__(self, local_op, shards, persistent_output=False):
        """Device vectors go through the shards' persistent exchange buffers: all-gather into one,
        the local operator writes its whole-map result straight into the other (``matvec_into`` of the
        tile-order operator; one copy for any other operator), reduce-scatter out of it -- no
        zero-fill, no copy and no map-sized allocation per matvec.  ``persistent_output``: the rank's
        result rows are also written into a kept vector, which is what ``matvec`` returns (valid
        until the next application; see ShardedLO)."""
        self.local_op, self.shards = local_op, shards
        self.persistent_output = bool(persistent_output)
        self._out = None
        if local_op.shape[0] != shards.n:
            raise lp.ShapeError("operator has %d rows, the shards cover %d" % (local_op.shape[0], shards.n))
        super(RowShardedNormalLO, self).__init__(shards.rows, shards.rows, self._mult, symmetric=True,
                                                 device_ok=lp.supports_device(local_op))

    def _mult(self, p_loc):
        sh = self.shards
        if not D.is_dev(p_loc):
            full = sh.gather(p_loc)
            if not isinstance(full, np.ndarray) and not full.is_contiguous():
                full = full.contiguous()
            return sh.reduce_scatter(self.local_op.matvec(full))
        full = sh.gather_view(p_loc)
        _, to_sum = sh.exchange_buffers(p_loc.device)
        plan = getattr(self.local_op, "_compiled", None)
        ops = plan() if plan is not None else [self.local_op]
        if len(ops) == 1 and hasattr(ops[0], "matvec_into"):
            ops[0].matvec_into(full, to_sum[:sh.n])
        else:
            to_sum[:sh.n].copy_(self.local_op.matvec(full))
        out = None
        if self.persistent_output:
            if self._out is None or self._out.device != p_loc.device:
                self._out = torch.empty(sh.rows, dtype=torch.float64, device=p_loc.device)
            out = self._out
        return sh.reduce_scatter_buffer(p_loc.device, out=out)


class _SlicedWeights(object):
    """The per-pixel weight arrays of a ProcessTimeSamples restricted to a pixel range (padded
    with empty pixels: hits 0, which M_BD maps to 0)."""

    def __init__(self, ces, shards):
        from .utilities.process_ces import _FIELDS
        npad = shards.pix_per_rank
        dev = getattr(ces, "_dev_weights", None) or {}
        self._dev_weights = {}
        for k in _FIELDS:
            src = dev.get(k)
            if src is None:
                self._dev_weights[k] = None
                continue
            out = torch.zeros(npad, dtype=torch.float64, device=src.device)
            out[:shards.pix_hi - shards.pix_lo] = src[shards.pix_lo:shards.pix_hi]
            self._dev_weights[k] = out

    def __getattr__(self, name):
        dw = self.__dict__.get("_dev_weights", {})
        if name in dw and dw[name] is not None:
            return D.to_host(dw[name])
        raise AttributeError(name)


def row_sharded_bd(ces, shards):
    """``M_BD`` acting on this rank's rows (its pixels' blocks only)."""
    from .interfaces.linearoperators import BlockDiagonalPreconditionerLO
    return BlockDiagonalPreconditionerLO(_SlicedWeights(ces, shards), shards.pix_per_rank, pol=shards.pol)


def row_sharded_two_level(Mbd_loc, Z_loc, AZ_loc, shards, apply="eig"):
    """Two-level preconditioner on row-sharded vectors: ``Z_loc`` / ``AZ_loc`` are this rank's
    rows of Z and A Z (``rows x r``, padded rows zero).  ``E = Z^T A Z`` is the all-reduced sum of
    the local contractions, ``Z^T r`` the all-reduced sum of the local products."""
    from .interfaces.linearoperators import CoarseLO, DeflationLO, TwoLevelPreconditionerLO
    r = int(Z_loc.shape[1])
    E = CoarseLO(Z_loc, AZ_loc, r, apply=apply, allreduce=shards.allreduce_)
    return TwoLevelPreconditionerLO(Mbd_loc, DeflationLO(Z_loc), DeflationLO(AZ_loc), E,
                                    allreduce=shards.allreduce_)
