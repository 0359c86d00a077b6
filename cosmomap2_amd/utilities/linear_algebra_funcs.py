"""
``dgemm``, ``norm2``, ``scalprod`` with the reference's semantics
(utilities/linear_algebra_funcs.py:16-44), evaluated on the GPU:

* ``dgemm(A, B)``    = ``A.T @ B.T``   (BLAS gemm with a=A.T, trans_b -- :27-29)
* ``norm2(q)``       = ``||q||_2``     (BLAS nrm2 -- :31-37)
* ``scalprod(a, b)`` = ``a . b``       (BLAS dot -- :39-44)
* ``get_legendre_polynomials(polyorder, size)``: set-up table of FilterLO (:47-59),
  a few KB built once on the host (the filter itself runs on the GPU)

NumPy inputs are uploaded and the result comes back as NumPy / float; tensors already
in HBM are used in place (``dgemm`` then returns a tensor).
"""
import math

import numpy as np

from .. import _hip
from .. import device as D

torch = D.torch

__all__ = ["dgemm", "norm2", "scalprod", "get_legendre_polynomials"]


def dgemm(A, B):
    """``A.T @ B.T`` for A (k x m) and B (n x k); lists are taken as matrices."""
    want_host = not (D.is_tensor(A) or D.is_tensor(B))
    if isinstance(A, list):
        A = np.asarray(A)
    if isinstance(B, list):
        B = np.asarray(B)
    dA = D.f64(np.ascontiguousarray(A, dtype=np.float64) if not D.is_tensor(A) else A)
    dB = D.f64(np.ascontiguousarray(B, dtype=np.float64) if not D.is_tensor(B) else B)
    if dA.dim() == 1:
        dA = dA.reshape(1, -1)
    if dB.dim() == 1:
        dB = dB.reshape(-1, 1)
    k, m = int(dA.shape[0]), int(dA.shape[1])
    n, k2 = int(dB.shape[0]), int(dB.shape[1])
    if k != k2:
        raise ValueError("dgemm: A is %dx%d, B is %dx%d; need A.shape[0] == B.shape[1]"
                         % (k, m, n, k2))
    out = D.empty(m * n)
    _hip.call("cm2_gemm_atbt", m, n, k, D.ptr(dA), D.ptr(dB), D.ptr(out), D.stream())
    out = out.reshape(m, n)
    return D.to_host(out) if want_host else out


def scalprod(a, b):
    """Scalar product of two vectors."""
    da, db = D.f64(a), D.f64(b)
    if da.numel() != db.numel():
        raise ValueError("scalprod: vectors of different length (%d, %d)" % (da.numel(), db.numel()))
    return D.dot(da.reshape(-1), db.reshape(-1))


def norm2(q):
    """Euclidean norm."""
    dq = D.f64(q).reshape(-1)
    return math.sqrt(D.dot(dq, dq))


def get_legendre_polynomials(polyorder, size):
    """``size x (polyorder+1)`` matrix whose columns are the Legendre polynomials on
    ``linspace(-1, 1, size)``, each scaled to unit 2-norm (reference :47-59)."""
    from scipy.linalg import get_blas_funcs
    from scipy.special import legendre
    legendres = np.empty([size, polyorder + 1])
    x = np.linspace(-1, 1, size)
    for i in range(polyorder + 1):
        col = legendre(i)(x)
        legendres[:, i] = col / get_blas_funcs('nrm2', (col,))(col)
    return legendres
