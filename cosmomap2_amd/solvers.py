"""
Preconditioned conjugate gradient with map vectors resident in HBM.

The reference has no solver of its own: it calls ``scipy.sparse.linalg.cg``
(tests/test_2level_preconditioner.py:52, tests/test_arnoldi_algorithm.py:46,91,
src/test_BD_precond_onto_real_data.py:47, src/test_M2_precond_onto_real_data.py:117,
and InverseLO.mult interfaces/linearoperators.py:882).  :func:`cg` keeps that calling
convention and reproduces scipy's recurrence step for step (scipy 1.15
``_isolve/iterative.py::cg``):

    atol = max(atol, rtol*||b||);  r = b - A x0 (b if x0 == 0)
    loop:  if ||r|| < atol: return x, 0
           z = M r; rho = r.z; p = z (first) | p = beta p + z, beta = rho/rho_prev
           q = A p; alpha = rho / (p.q); x += alpha p; r -= alpha q; callback(x)
    return x, maxiter

"Iterations" = number of callback invocations, as the reference counts them
(src/test_BD_precond_onto_real_data.py:41-47).  The scalars alpha and beta never
visit the host: the fused update kernels read rho, rho_prev and p.q from device
memory; the only per-iteration synchronisation is the 8-byte read of ||r||^2 for
the convergence test, and that read is deferred behind the launches of the next iteration
(see the comment in :func:`cg`), so the GPU does not wait for the host between iterations.
"""
import math

import numpy as np

from . import _hip
from . import device as D
from . import linop as lp

torch = D.torch

__all__ = ["cg"]


def _apply(op, x):
    """Apply an operator / callable / matrix to a device vector, staying in HBM when
    the operator can."""
    if op is None:
        return x
    if isinstance(op, lp.BaseLinearOperator):
        if lp.supports_device(op):
            return D.f64(op.matvec(x))
        return D.f64(np.asarray(op.matvec(D.to_host(x))))
    if callable(op) and not hasattr(op, "matvec"):
        return D.f64(op(x))
    if hasattr(op, "matvec"):                      # scipy LinearOperator and the like
        return D.f64(np.asarray(op.matvec(D.to_host(x))))
    return D.f64(np.asarray(op).dot(D.to_host(x)))


def cg(A, b, x0=None, tol=None, maxiter=None, M=None, callback=None, atol=0., rtol=1e-5,
       sync=None, dot_reduce=None):
    """
    Solve ``A x = b`` by PCG on the GPU.  Arguments and return value ``(x, info)`` follow
    ``scipy.sparse.linalg.cg``; ``tol`` is accepted as the old name of ``rtol`` (the
    reference still passes ``tol=``).  ``b`` / ``x0`` may be NumPy arrays (result is NumPy)
    or float64 tensors in HBM (result is a tensor, ``callback`` receives a tensor).

    ``sync``: optional callable mapping the host value of ||r||^2 to the value every
    rank must use (sharding.make_sync: a max over ranks so that all ranks take the same
    branch); when it has a ``reduce_`` attribute (RCCL) the reduction is queued on the device
    in front of the deferred read instead, and the host form is not called.  ``dot_reduce``: optional callable summing a device scalar over ranks in place --
    for row-sharded vectors (sharding.RowShards.allreduce_), where every rank holds only its
    rows and each scalar product is a local sum plus an 8-byte all-reduce; alpha and beta still
    never visit the host.
    """
    D.require_gpu()
    if tol is not None:
        rtol = tol
    host_io = not D.is_tensor(b)
    bd = D.f64(b).reshape(-1)
    n = bd.numel()
    lib = _hip.load()
    work = D.reduce_work()
    st = D.stream

    def dot_into(out, u, v):
        _hip.check(lib.cm2_dot(n, D.ptr(u), D.ptr(v), D.ptr(out), D.ptr(work), st()))
        if dot_reduce is not None:
            dot_reduce(out)

    scal = D.zeros(5)                   # rho_a, rho_b, pq, rr, tmp  (device scalars)
    rho = [scal[0:1], scal[1:2]]
    pq, rr, tmp = scal[2:3], scal[3:4], scal[4:5]

    dot_into(tmp, bd, bd)
    bnrm2 = math.sqrt(float(tmp.item()))
    atol = max(float(atol), float(rtol) * bnrm2)
    if bnrm2 == 0:
        return (D.to_host(bd) if host_io else bd.clone()), 0
    if maxiter is None:
        maxiter = n * 10

    if x0 is None:
        x = D.zeros(n)
        r = bd.clone()
    else:
        x = D.f64(x0).reshape(-1).clone()
        dot_into(tmp, x, x)
        xx = float(tmp.item())
        if xx != 0.0 or xx != xx:
            r = D.add_scaled(bd, -1.0, _apply(A, x))          # r = b - A x0
        else:
            r = bd.clone()
    # ||r||^2 reaches the host through a deferred read: once it is final on the stream (summed
    # over ranks for row-sharded vectors, max-reduced on the device when `sync` can), an
    # asynchronous copy to pinned memory and an event are queued behind it.  The host does not
    # wait there: it first queues everything of the NEXT iteration that touches neither x nor r
    # (z = M r, rho, the update of p, q = A p, p.q) and only then waits for the event -- by which
    # time the GPU has that iteration's kernels in its queue and never idles while Python issues
    # launches.  When the value says "converged", x is returned as it is and the speculative
    # work is simply dropped (it wrote z, p, q and two scalars only), so callback invocations,
    # return values and iteration counts are those of scipy's recurrence.
    rr_pinned, rr_event = _deferred_acquire()
    reduce_dev = getattr(sync, "reduce_", None) if sync is not None else None
    try:
        return _cg_loop(A, M, x, r, n, lib, work, st, dot_into, scal, maxiter, atol, callback, host_io,
                        sync, dot_reduce, reduce_dev, rr_pinned, rr_event)
    finally:
        _deferred_release(rr_pinned, rr_event)


# Pinned 8-byte buffers and events of the deferred reads: allocating page-locked memory costs from
# 0.1 ms to tens of milliseconds, so they are kept and reused; a pool rather than one buffer because
# a solve may run inside another one's operator (InverseLO inside a product).
# Keyed by device: an event first recorded on one device cannot be recorded on another.
_deferred_pool = {}


def _deferred_acquire():
    pool = _deferred_pool.setdefault(torch.cuda.current_device(), [])
    if pool:
        return pool.pop()
    return torch.empty(1, dtype=torch.float64).pin_memory(), torch.cuda.Event()


def _deferred_release(buf, ev):
    # the last queued copy into `buf` may still be in flight (a solve that stopped at maxiter queues
    # one it never reads): the pair only goes back once that copy is done
    ev.synchronize()
    pool = _deferred_pool.setdefault(torch.cuda.current_device(), [])
    if len(pool) < 8:
        pool.append((buf, ev))


def _cg_loop(A, M, x, r, n, lib, work, st, dot_into, scal, maxiter, atol, callback, host_io, sync,
             dot_reduce, reduce_dev, rr_pinned, rr_event):
    """The iteration of :func:`cg` (see there)."""
    rho = [scal[0:1], scal[1:2]]
    pq, rr = scal[2:3], scal[3:4]

    def post_rr():
        if reduce_dev is not None:
            reduce_dev(rr)
        rr_pinned.copy_(rr, non_blocking=True)
        rr_event.record()

    def wait_rr():
        rr_event.synchronize()
        value = float(rr_pinned[0])
        if sync is not None and reduce_dev is None:
            value = sync(value)
        return value

    dot_into(rr, r, r)
    post_rr()

    def ahead_cheap():
        """z = M r, rho, p of iteration `iteration`: per-pixel and vector kernels (a few per cent of
        an iteration), always queued before the stop test's value is waited for."""
        nonlocal p
        z = _apply(M, r) if M is not None else r
        dot_into(rho[cur], r, z)
        if iteration > 0:
            _hip.check(lib.cm2_pcg_update_p(n, D.ptr(rho[cur]), D.ptr(rho[1 - cur]), D.ptr(z),
                                            D.ptr(p), st()))
        else:
            p = z.clone()

    def ahead_matvec():
        """q = A p, p.q of iteration `iteration` (nothing here or above writes x or r)."""
        q = _apply(A, p)
        dot_into(pq, p, q)
        return q

    # Running ahead with the matvec costs one iteration's worth of GPU time when the stop test then
    # says "converged" (the queued kernels cannot be recalled), so it is skipped when the two last
    # residuals known to the host predict convergence at this test (geometric extrapolation, with a
    # factor 10 in the norm to spare): that iteration queues only the cheap kernels ahead and the
    # matvec after the wait -- what the GPU then waits for is one launch, not the whole host side
    # of an iteration.  The prediction only chooses between two orders of the same operations;
    # results never depend on it.
    p = None
    cur = 0
    known = []                                       # ||r||^2 values the host has seen
    for iteration in range(int(maxiter)):
        predicted = known[-1] * (known[-1] / known[-2]) if len(known) >= 2 and known[-2] > 0 \
            else (known[-1] if known else float("inf"))
        run_ahead = not (predicted <= 100.0 * atol * atol)
        ahead_cheap()
        q = ahead_matvec() if run_ahead else None
        rr_host = wait_rr()                          # ||r||^2 of the state before this iteration
        known.append(rr_host)
        if math.sqrt(rr_host) < atol:
            return (D.to_host(x) if host_io else x), 0
        if q is None:
            q = ahead_matvec()
        _hip.check(lib.cm2_pcg_update_xr(n, D.ptr(rho[cur]), D.ptr(pq), D.ptr(p), D.ptr(q),
                                         D.ptr(x), D.ptr(r), D.ptr(rr), D.ptr(work), st()))
        cur = 1 - cur
        if dot_reduce is not None:
            dot_reduce(rr)
        post_rr()
        if callback is not None:
            callback(D.to_host(x) if host_io else x)
    return (D.to_host(x) if host_io else x), int(maxiter)
