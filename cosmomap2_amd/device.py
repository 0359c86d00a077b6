"""
Device-memory plumbing: torch tensors are used ONLY as HBM allocations, streams
and (in sharding.py) the torch.distributed transport.  All arithmetic on map- or
TOD-sized data goes through the HIP kernels of libcosmomap2_hip.so.

Vector kinds handled by the host classes:
  * numpy.ndarray            -- API edge, like the reference (host round trip)
  * torch cuda tensor (f64)  -- resident in HBM, no transfer
  * torch cpu tensor         -- only met in the gloo multi-process tests
"""
import threading

import numpy as np

try:
    import torch
except Exception:         # pragma: no cover
    torch = None

from . import _hip


def gpu_available():
    return torch is not None and torch.cuda.is_available()


def require_gpu():
    if not gpu_available():
        raise _hip.HipError("cosmomap2_amd needs an AMD GPU visible to PyTorch-ROCm; "
                            "there is no CPU fallback for the map-making kernels")
    _hip.load()


def dev():
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    """hipStream_t of torch's current stream, as an integer for ctypes."""
    return torch.cuda.current_stream().cuda_stream


def is_dev(x):
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


def is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


# Host <-> device copies of NumPy arrays go through page-locked staging buffers, in pieces: the HIP
# runtime pins pageable host memory for copies of 1 MB and more, and when that memory is freed later the
# kernel driver evicts the process's GPU queues for 10-30 ms (cm2_core.hip, profiles/r04_stall_probe.md).
_STAGE_BYTES = 8 << 20
_stage = threading.local()         # a pair of buffers per (host thread, device): copy_ releases the GIL, two
                                   # threads in to_dev / to_host at once must not share staging memory


def _stage_pair():
    d = torch.cuda.current_device()
    pairs = getattr(_stage, "pairs", None)
    if pairs is None:
        pairs = _stage.pairs = {}
    if d not in pairs:
        pairs[d] = [(torch.empty(_STAGE_BYTES, dtype=torch.uint8).pin_memory(), torch.cuda.Event())
                    for _ in range(2)]
    return pairs[d]


def _upload(arr):
    """Contiguous NumPy array -> new tensor on the current GPU."""
    t = _alloc(lambda: torch.empty(arr.shape, dtype=torch.from_numpy(arr.reshape(-1)[:0]).dtype, device=dev()))
    nbytes = arr.nbytes
    if nbytes == 0:
        return t
    src = torch.from_numpy(arr.reshape(-1).view(np.uint8))
    dst = t.reshape(-1).view(torch.uint8)
    pair = _stage_pair()
    used = [False, False]
    for k, o in enumerate(range(0, nbytes, _STAGE_BYTES)):
        buf, ev = pair[k & 1]
        if used[k & 1]:
            ev.synchronize()                         # the copy that last read this buffer is done
        m = min(_STAGE_BYTES, nbytes - o)
        buf[:m].copy_(src[o:o + m])
        dst[o:o + m].copy_(buf[:m], non_blocking=True)
        ev.record()
        used[k & 1] = True
    for (buf, ev), u in zip(pair, used):
        if u:
            ev.synchronize()
    return t


def _download(t):
    """Device tensor -> new NumPy array."""
    t = t.detach().contiguous()
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    nbytes = out.nbytes
    if nbytes == 0:
        return out
    dst = torch.from_numpy(out.reshape(-1).view(np.uint8))
    src = t.reshape(-1).view(torch.uint8)
    pair = _stage_pair()
    pending = [None, None]
    for k, o in enumerate(range(0, nbytes, _STAGE_BYTES)):
        buf, ev = pair[k & 1]
        if pending[k & 1] is not None:
            ev.synchronize()
            po, pm = pending[k & 1]
            dst[po:po + pm].copy_(buf[:pm])
        m = min(_STAGE_BYTES, nbytes - o)
        buf[:m].copy_(src[o:o + m], non_blocking=True)
        ev.record()
        pending[k & 1] = (o, m)
    for (buf, ev), pend in zip(pair, pending):
        if pend is not None:
            ev.synchronize()
            dst[pend[0]:pend[0] + pend[1]].copy_(buf[:pend[1]])
    return out


def to_dev(a, dtype=None):
    """Upload (or pass through) as a contiguous tensor on the current GPU."""
    require_gpu()
    if is_tensor(a):
        t = a
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if not t.is_cuda:
            t = t.to(dev())
        return t.contiguous()
    arr = np.ascontiguousarray(a)
    if dtype is not None:
        want = torch.empty(0, dtype=dtype).numpy().dtype
        if arr.dtype != want:
            arr = arr.astype(want)
    return _upload(arr)


def f64(a):
    return to_dev(a, torch.float64)


def i32(a):
    return to_dev(a, torch.int32)


def _alloc(make):
    """torch's allocator and the library's own block cache (cm2_core.hip) share the device and do not
    see each other's idle memory: when torch runs out, the library's cached blocks (and torch's own)
    go back to the driver and the allocation is tried once more."""
    try:
        return make()
    except torch.cuda.OutOfMemoryError:
        from . import _hip
        _hip.call("cm2_release_cached_memory")
        torch.cuda.empty_cache()
        return make()


def empty(n, dtype=None):
    require_gpu()
    return _alloc(lambda: torch.empty(int(n), dtype=dtype or torch.float64, device=dev()))


def zeros(n, dtype=None):
    require_gpu()
    return _alloc(lambda: torch.zeros(int(n), dtype=dtype or torch.float64, device=dev()))


def to_host(t):
    if isinstance(t, np.ndarray):
        return t
    if t.is_cuda:
        return _download(t)
    return t.detach().numpy()


def like_input(result_dev, x):
    """Return `result_dev` in the same kind as the caller's vector `x`."""
    if isinstance(x, np.ndarray):
        return to_host(result_dev)
    if is_tensor(x) and not x.is_cuda:
        return result_dev.cpu()
    return result_dev


# ----------------------------------------------------------------- algebra ------
_work = {}


def reduce_work():
    """Scratch for the two-stage reductions (one per device)."""
    d = torch.cuda.current_device()
    if d not in _work:
        _work[d] = torch.empty(int(_hip.load().cm2_reduce_work_doubles()), dtype=torch.float64,
                               device=dev())
    return _work[d]


def add_scaled(y, alpha, x):
    """New vector y + alpha*x (operator sums and differences)."""
    if is_dev(y) or is_dev(x):
        y, x = f64(y), f64(x)
        out = y.clone()
        _hip.call("cm2_axpy", out.numel(), float(alpha), ptr(x), ptr(out), stream())
        return out
    return y + alpha * x


def scaled(alpha, x):
    if is_dev(x):
        out = x.clone()
        _hip.call("cm2_scal", out.numel(), float(alpha), ptr(out), stream())
        return out
    return x * alpha


def multiply(d, x):
    if is_dev(x):
        out = torch.empty_like(x)
        _hip.call("cm2_xmy", x.numel(), ptr(d), ptr(x), ptr(out), stream())
        return out
    return d * x


def dot_dev(x, y):
    """Device dot product left in HBM (a 1-element tensor; no synchronisation)."""
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    _hip.call("cm2_dot", x.numel(), ptr(x), ptr(y), ptr(out), ptr(reduce_work()), stream())
    return out


def dot(x, y):
    """Device dot product -> python float (synchronises)."""
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    _hip.call("cm2_dot", x.numel(), ptr(x), ptr(y), ptr(out), ptr(reduce_work()), stream())
    return float(out.item())


def memory_info():
    """Device memory of the HIP library (not torch's): bytes held by live plans and operators,
    bytes cached for reuse, requests served from the cache / by the driver."""
    import ctypes
    from . import _hip
    info = (ctypes.c_int64 * 4)()
    _hip.call("cm2_device_memory_info", info)
    return dict(zip(("live_bytes", "cached_bytes", "cache_hits", "driver_allocations"), [int(v) for v in info]))


def release_cached_memory():
    """Returns the library's cached device blocks to the driver (the counterpart of
    torch.cuda.empty_cache() for libcosmomap2_hip.so)."""
    from . import _hip
    _hip.call("cm2_release_cached_memory")

