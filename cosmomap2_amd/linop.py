"""
Minimal matrix-free operator algebra with the surface of the third-party ``linop``
package that COSMOMAP2's operators subclass (the reference imports it as ``lp``:
interfaces/linearoperators.py:14; it is not vendored and not installed here).

Only what the reference's own code uses is provided: the constructor
``LinearOperator(nargin, nargout, matvec, rmatvec=None, symmetric=False, dtype=...)``
(linearoperators.py:538-539, 600-601, 858-859), ``op*vector``, ``op*op``, ``op+op``,
``op-op``, scalar scaling, ``.T``/``.H``, ``.matvec``/``.rmatvec``, ``.shape``,
``.dtype``, ``.symmetric``, ``.to_array()``, ``IdentityOperator``,
``DiagonalOperator`` (linearoperators.py:680), ``ShapeError``, ``null_log``
(interfaces/blkop.py:1-2).  Operators are accepted by
``scipy.sparse.linalg.aslinearoperator`` / ``cg`` / ``eigsh``.

Vectors may be NumPy arrays (host round trip at the API edge, like the reference)
or torch tensors resident in HBM (no transfer): every operator just forwards what
it is given, and the algebra (sum, scaling) uses arithmetic both types support.

Products keep the flat list of their primitive factors (``_chain``) so that the
hot-path pattern  P.T * N * P  can be replaced by ONE fused kernel when it is
applied (see interfaces/linearoperators.py: ``_fuse_chain``).
"""
import logging
import numbers

import numpy as np

try:                      # torch is the device-memory plumbing; optional for host-only use
    import torch
except Exception:         # pragma: no cover
    torch = None

from . import device as _vec

__all__ = ["BaseLinearOperator", "LinearOperator", "IdentityOperator", "DiagonalOperator",
           "ZeroOperator", "ShapeError", "null_log", "is_vector", "register_chain_fusion",
           "supports_device"]


class ShapeError(Exception):
    """Operator / vector shapes do not match (linop.ShapeError, used at blkop.py:80-81)."""


null_log = logging.getLogger("cosmomap2_amd.nulllog")
null_log.setLevel(logging.INFO)
null_log.addHandler(logging.NullHandler())

_chain_fusers = []


def register_chain_fusion(fn):
    """fn(list_of_primitive_ops) -> new list (possibly shorter) or None."""
    _chain_fusers.append(fn)
    return fn


def is_vector(x):
    if isinstance(x, np.ndarray):
        return True
    return torch is not None and isinstance(x, torch.Tensor)


class BaseLinearOperator(object):
    """Shape / dtype / symmetry bookkeeping (linop.BaseLinearOperator)."""

    def __init__(self, nargin, nargout, symmetric=False, hermitian=False, dtype=np.float64,
                 **kwargs):
        self.__nargin = int(nargin)
        self.__nargout = int(nargout)
        self.__symmetric = bool(symmetric) or bool(hermitian)
        self.__shape = (self.__nargout, self.__nargin)
        self.__dtype = np.dtype(dtype)
        self._nMatvec = 0
        self.logger = kwargs.get("logger", null_log)

    nargin = property(lambda self: self.__nargin)
    nargout = property(lambda self: self.__nargout)
    symmetric = property(lambda self: self.__symmetric)
    hermitian = property(lambda self: self.__symmetric)
    shape = property(lambda self: self.__shape)
    dtype = property(lambda self: self.__dtype)
    nMatvec = property(lambda self: self._nMatvec)

    def __repr__(self):
        return "<%s %dx%d>" % (self.__class__.__name__, self.nargout, self.nargin)


def supports_device(op):
    """True when ``op.matvec`` accepts float64 tensors resident in HBM."""
    return bool(getattr(op, "_device_ok", False))


class LinearOperator(BaseLinearOperator):
    """Operator defined by a ``matvec`` callable and optionally its transpose.
    ``device_ok=True`` declares that the callables also accept HBM-resident tensors."""
    _device_ok = False

    def __init__(self, nargin, nargout, matvec, rmatvec=None, matvec_transp=None,
                 matvec_adj=None, symmetric=False, **kwargs):
        super(LinearOperator, self).__init__(nargin, nargout, symmetric=symmetric, **kwargs)
        self.__matvec = matvec
        transp = rmatvec or matvec_transp or matvec_adj
        if "device_ok" in kwargs:
            self._device_ok = bool(kwargs["device_ok"])
        self._is_transpose = "_transpose_of" in kwargs
        if self._is_transpose:
            self.__T = kwargs["_transpose_of"]
        elif self.symmetric:
            self.__T = self
        elif transp is not None:
            self.__T = LinearOperator(nargout, nargin, transp, symmetric=False,
                                      dtype=kwargs.get("dtype", np.float64),
                                      logger=self.logger, _transpose_of=self,
                                      device_ok=supports_device(self))
        else:
            self.__T = None
        self._chain = [self]

    # -- transposes ---------------------------------------------------------------
    @property
    def T(self):
        if self.__T is None:
            raise NotImplementedError("transpose of this operator is not defined")
        return self.__T

    H = T           # real operators only on this path

    # -- application --------------------------------------------------------------
    def _check(self, x):
        n = x.shape[0] if is_vector(x) else len(x)
        if n != self.nargin:
            raise ShapeError("Multiplying with vector of wrong shape: operator is %dx%d, "
                             "vector has %d entries" % (self.nargout, self.nargin, n))

    def matvec(self, x):
        if not is_vector(x):
            x = np.asarray(x, dtype=np.float64)
        if x.ndim == 2 and x.shape[1] == 1:          # scipy passes columns sometimes
            return self.matvec(x[:, 0]).reshape(-1, 1)
        self._check(x)
        self._nMatvec += 1
        return self.__matvec(x)

    def rmatvec(self, x):
        return self.T.matvec(x)

    __call__ = matvec

    def matmat(self, X):
        X = np.asarray(X)
        return np.column_stack([self.matvec(np.ascontiguousarray(X[:, j]))
                                for j in range(X.shape[1])])

    def to_array(self):
        """Dense matrix, one matvec per column (used at tests/test_coarse_wclass.py:56)."""
        n, m = self.shape
        out = np.empty((n, m))
        e = np.zeros(m)
        for j in range(m):
            e[j] = 1.0
            out[:, j] = np.asarray(self.matvec(e.copy()))
            e[j] = 0.0
        return out

    full = to_array

    # -- algebra ------------------------------------------------------------------
    def __mul__(self, other):
        if isinstance(other, numbers.Number):
            return _scaled(self, other)
        if isinstance(other, BaseLinearOperator):
            return _product(self, other)
        return self.matvec(other)

    def __rmul__(self, other):
        if isinstance(other, numbers.Number):
            return _scaled(self, other)
        raise ValueError("cannot multiply %r by an operator" % type(other))

    def __truediv__(self, other):
        if not isinstance(other, numbers.Number):
            raise ValueError("operators can only be divided by scalars")
        return _scaled(self, 1.0 / other)

    def __neg__(self):
        return _scaled(self, -1.0)

    def __add__(self, other):
        return _sum(self, other, +1.0)

    def __sub__(self, other):
        return _sum(self, other, -1.0)


def _maybe_T(op):
    try:
        return op.T
    except NotImplementedError:
        return None


def _scaled(op, alpha):
    alpha = float(alpha)
    t = _maybe_T(op)
    return LinearOperator(op.nargin, op.nargout, lambda x: _vec.scaled(alpha, op.matvec(x)),
                          rmatvec=(None if t is None
                                   else (lambda x: _vec.scaled(alpha, t.matvec(x)))),
                          symmetric=op.symmetric, dtype=op.dtype,
                          device_ok=supports_device(op))


def _sum(a, b, sign):
    if not isinstance(b, BaseLinearOperator):
        raise ValueError("cannot add an operator and %r" % type(b))
    if a.shape != b.shape:
        raise ShapeError("Cannot add operators of shapes %r and %r" % (a.shape, b.shape))
    ta, tb = _maybe_T(a), _maybe_T(b)
    mv = lambda x: _vec.add_scaled(a.matvec(x), sign, b.matvec(x))
    rmv = None
    if ta is not None and tb is not None:
        rmv = lambda x: _vec.add_scaled(ta.matvec(x), sign, tb.matvec(x))
    return LinearOperator(a.nargin, a.nargout, mv, rmatvec=rmv,
                          symmetric=a.symmetric and b.symmetric,
                          dtype=np.result_type(a.dtype, b.dtype),
                          device_ok=supports_device(a) and supports_device(b))


class _Product(LinearOperator):
    """a*b with the flattened factor list kept for kernel fusion."""

    def __init__(self, factors):
        self._factors = list(factors)
        self._plan = None
        first, last = self._factors[0], self._factors[-1]
        for l, r in zip(self._factors[:-1], self._factors[1:]):
            if l.nargin != r.nargout:
                raise ShapeError("Cannot multiply operators of shapes %r and %r"
                                 % (l.shape, r.shape))
        transposable = all(_maybe_T(f) is not None for f in self._factors)
        super(_Product, self).__init__(
            last.nargin, first.nargout, self._apply,
            rmatvec=(self._apply_T if transposable else None), symmetric=False,
            dtype=np.result_type(*[f.dtype for f in self._factors]),
            device_ok=all(supports_device(f) for f in self._factors))
        self._chain = self._factors

    def _compiled(self):
        if self._plan is None:
            chain = list(self._factors)
            for fuse in _chain_fusers:
                out = fuse(chain)
                if out is not None:
                    chain = out
            self._plan = chain
        return self._plan

    def _apply(self, x):
        # host vector through an all-device chain: upload once, download once
        stay = isinstance(x, np.ndarray) and self._device_ok and _vec.gpu_available()
        y = _vec.f64(x) if stay else x
        for op in reversed(self._compiled()):
            y = op.matvec(y)
        return _vec.to_host(y) if stay else y

    def _apply_T(self, x):
        stay = isinstance(x, np.ndarray) and self._device_ok and _vec.gpu_available()
        y = _vec.f64(x) if stay else x
        for op in self._factors:
            y = op.T.matvec(y)
        return _vec.to_host(y) if stay else y


def _product(a, b):
    if a.nargin != b.nargout:
        raise ShapeError("Cannot multiply operators of shapes %r and %r" % (a.shape, b.shape))
    ca = getattr(a, "_chain", [a])
    cb = getattr(b, "_chain", [b])
    return _Product(list(ca) + list(cb))


class IdentityOperator(LinearOperator):
    def __init__(self, nargin, **kwargs):
        kwargs.setdefault("device_ok", True)
        super(IdentityOperator, self).__init__(nargin, nargin, lambda x: x, symmetric=True,
                                               **kwargs)


class ZeroOperator(LinearOperator):
    def __init__(self, nargin, nargout, **kwargs):
        def mv(x):
            if isinstance(x, np.ndarray):
                return np.zeros(nargout, dtype=x.dtype)
            return x.new_zeros(nargout)

        def rmv(x):
            if isinstance(x, np.ndarray):
                return np.zeros(nargin, dtype=x.dtype)
            return x.new_zeros(nargin)
        kwargs.setdefault("device_ok", True)
        super(ZeroOperator, self).__init__(nargin, nargout, mv, rmatvec=rmv, **kwargs)


class DiagonalOperator(LinearOperator):
    """diag(d): matvec is ``d * x`` (linop.DiagonalOperator, linearoperators.py:680)."""

    def __init__(self, diag, **kwargs):
        self._diag_host = np.ascontiguousarray(diag, dtype=np.float64).copy()
        if self._diag_host.ndim != 1:
            raise ValueError("Input must be 1-d array")
        self._diag_dev = None
        n = self._diag_host.shape[0]
        kwargs.setdefault("device_ok", True)
        super(DiagonalOperator, self).__init__(n, n, self._mult, symmetric=True, **kwargs)

    @property
    def diag(self):
        return self._diag_host

    def _mult(self, x):
        if isinstance(x, np.ndarray):
            return self._diag_host * x
        if self._diag_dev is None or self._diag_dev.device != x.device:
            self._diag_dev = _vec.f64(self._diag_host) if x.is_cuda else torch.from_numpy(self._diag_host)
        return _vec.multiply(self._diag_dev, x)
