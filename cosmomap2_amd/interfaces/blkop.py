"""
Block-diagonal operator container (reference: interfaces/blkop.py:140-242, itself a
copy of linop's block operators).  Only ``BlockDiagonalLinearOperator`` is provided:
it is the base of ``BlockLO`` and the only class of that file the reference ever
instantiates.

The generic form applies block ``b`` to the slice of ``x`` it owns (blk_matvec,
blkop.py:178-208).  ``BlockLO`` replaces that Python loop by a single device call and
only materialises the per-block operator list when someone asks for it.
"""
import numpy as np

from ..linop import BaseLinearOperator, LinearOperator, ShapeError, null_log  # noqa: F401
from .. import device as D

__all__ = ["BlockDiagonalLinearOperator"]


class BlockDiagonalLinearOperator(LinearOperator):
    """``BlockDiagonalLinearOperator([A, B, C])``: y = [A x_A, B x_B, C x_C]."""

    def __init__(self, blocks, **kwargs):
        lazy = kwargs.pop("_lazy", None)
        fused = kwargs.pop("matvec", None)
        if lazy is not None:
            self._factory, sizes = lazy
            self._blocks_cache = None
            nargins = nargouts = [int(s) for s in sizes]
            symmetric = True
            op_dtype = np.float64
        else:
            try:
                for b in blocks:
                    b.shape
            except (TypeError, AttributeError):
                raise ValueError("blocks should be a flattened list of operators")
            self._factory = None
            self._blocks_cache = list(blocks)
            nargins = [b.shape[-1] for b in blocks]
            nargouts = [b.shape[0] for b in blocks]
            symmetric = all(b.symmetric for b in blocks)
            op_dtype = np.result_type(*[b.dtype for b in blocks[:10]])
        self._nargins, self._nargouts = nargins, nargouts
        log = kwargs.get("logger", null_log)
        log.debug("Building new BlockDiagonalLinearOperator")
        mv = fused if fused is not None else (lambda x: self._blk_matvec(x, False))
        rmv = None if symmetric else (lambda x: self._blk_matvec(x, True))
        super(BlockDiagonalLinearOperator, self).__init__(
            sum(nargins), sum(nargouts), mv, rmatvec=rmv, symmetric=symmetric, dtype=op_dtype,
            **kwargs)

    def _blk_matvec(self, x, transposed):
        nin = self._nargouts if transposed else self._nargins
        nout = self._nargins if transposed else self._nargouts
        if x.shape[0] != sum(nin):
            raise ShapeError("Multiplying with vector of wrong shape.")
        pieces = []
        start = 0
        for blk, n in zip(self.blocks, nin):
            op = blk.T if transposed else blk
            pieces.append(op * x[start:start + n])
            start += n
        if isinstance(x, np.ndarray):
            return np.concatenate([np.asarray(p) for p in pieces])
        return D.torch.cat(pieces)

    @property
    def blocks(self):
        """The list of blocks defining the block diagonal operator."""
        if self._blocks_cache is None:
            self._blocks_cache = self._factory()
        return self._blocks_cache

    _blocks = blocks

    def __getitem__(self, idx):
        blks = self.blocks[idx]
        if isinstance(idx, slice):
            return BlockDiagonalLinearOperator(blks)
        return blks

    def __setitem__(self, idx, ops):
        seq = ops if isinstance(ops, (list, tuple)) else [ops]
        for op in seq:
            if not isinstance(op, BaseLinearOperator):
                raise ValueError("Block operators can only contain linear operators")
        self.blocks[idx] = ops
