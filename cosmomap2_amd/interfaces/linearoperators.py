"""
The map-making operators of COSMOMAP2 with the reference's class names,
constructor signatures and attributes (interfaces/linearoperators.py of the
reference, cited per class), running on MI355X through libcosmomap2_hip.so.

Every ``mult`` accepts a NumPy array (uploaded, result downloaded -- the
reference's calling convention) or a float64 torch tensor already in HBM
(returned as a tensor, nothing crosses PCIe).  There is no CPU code path: without
the HIP library and a GPU the constructors raise.

Ownership: the reference keeps *references* to the caller's arrays
(linearoperators.py:531-534, 848-856).  Here the arrays are snapshotted into HBM
at construction (NumPy arrays by their upload, HBM tensors by a copy); later mutation
is not seen.  The one shared buffer is the flagged pixel stream of a ProcessTimeSamples
handed over as ``pix_samples is CES.pixs``: it is final once ProcessTimeSamples returns.
"""
import ctypes
import os
import weakref

import numpy as np
import scipy.linalg as sla

from .. import _hip
from .. import device as D
from .. import linop as lp
from . import blkop as blk

torch = D.torch

__all__ = ["SparseLO", "ToeplitzLO", "BlockLO", "BlockDiagonalLO",
           "BlockDiagonalPreconditionerLO", "InverseLO", "CoarseLO", "DeflationLO",
           "TwoLevelPreconditionerLO", "FilterLO", "GroundFilterLO", "set_pointing_mode", "lp"]

_I64P = ctypes.POINTER(ctypes.c_int64)
_DBLP = ctypes.POINTER(ctypes.c_double)


def _as_i64(a):
    arr = np.ascontiguousarray(a, dtype=np.int64)
    return arr, arr.ctypes.data_as(_I64P)


def _as_f64(a):
    arr = np.ascontiguousarray(a, dtype=np.float64)
    return arr, arr.ctypes.data_as(_DBLP)


class _DeviceOp(lp.LinearOperator):
    """Marker base: matvec understands device tensors (lets operator products stay
    in HBM between factors)."""
    _device_ok = True


# ==================================================================== SparseLO ===
class SparseLO(_DeviceOp):
    """
    Pointing matrix P (reference: interfaces/linearoperators.py:326-557).

    ``SparseLO(n, m, pix_samples, pol=1, angle_processed=None)`` with ``n`` pixels,
    ``m`` samples, ``pix_samples`` the pixel of each sample (-1 = flagged) and
    ``angle_processed`` a :class:`ProcessTimeSamples` carrying ``cos``/``sin``.
    ``P*x`` gathers (mult/mult_qu/mult_iqu, :356-497); ``P.T*v`` is the scatter-add
    (rmult*, :385-526) computed as a fixed-order per-pixel reduction.
    """

    def __init__(self, n, m, pix_samples, pol=1, angle_processed=None):
        self.ncols = int(n)
        self.nrows = int(m)
        self.pol = pol
        self.pairs = pix_samples
        if pol not in (1, 2, 3):
            # same exception type and text class as linearoperators.py:549-550
            raise RuntimeError("No valid polarization key set!\t=>\tpol=%r \n "
                               "Possible values are pol=%d(I),%d(QU), %d(IQU)." % (pol, 1, 2, 3))
        D.require_gpu()
        self._d_pix = None
        shared = getattr(angle_processed, "_d_pix", None)
        if shared is not None and pix_samples is getattr(angle_processed, "pixs", None):
            self._d_pix = shared                     # already flagged and resident
        if self._d_pix is None:
            self._d_pix = D.i32(pix_samples)
            if D.is_dev(pix_samples) and self._d_pix.data_ptr() == pix_samples.data_ptr():
                # a caller-owned HBM tensor: snapshot it, like a NumPy array is by its upload --
                # P_time reads the live buffer while the pixel-major and tile plans are built
                # once, so flagging samples after construction would make P and P^T disagree
                self._d_pix = self._d_pix.clone()
        if self._d_pix.numel() != self.nrows:
            raise lp.ShapeError("pix_samples has %d entries, expected m=%d"
                                % (self._d_pix.numel(), self.nrows))
        self._d_cos = self._d_sin = None
        if pol > 1:
            if angle_processed is None:
                raise RuntimeError("pol=%d needs angle_processed (cos/sin of 2*phi)" % pol)
            self._angles = angle_processed         # .cos / .sin are linked, not copied (:533-534)
            dc = getattr(angle_processed, "_d_cos", None)
            ds = getattr(angle_processed, "_d_sin", None)
            self._d_cos = dc if dc is not None else D.f64(angle_processed.cos)
            self._d_sin = ds if ds is not None else D.f64(angle_processed.sin)
        handle = ctypes.c_void_p()
        _hip.call("cm2_pointing_create", ctypes.byref(handle), D.ptr(self._d_pix),
                  D.ptr(self._d_cos), D.ptr(self._d_sin), self.nrows, self.ncols, int(pol),
                  D.stream())
        self._plan = handle
        self._weights_of = None          # BlockLO whose diagonal is attached for the fused matvec
        self.__runcase = {1: "I", 2: "QU", 3: "IQU"}[pol]
        mv = {1: self.mult, 2: self.mult_qu, 3: self.mult_iqu}[pol]
        rmv = {1: self.rmult, 2: self.rmult_qu, 3: self.rmult_iqu}[pol]
        super(SparseLO, self).__init__(nargin=self.pol * self.ncols, nargout=self.nrows,
                                       matvec=mv, symmetric=False, rmatvec=rmv)

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan:
            try:
                _hip.load().cm2_pointing_destroy(plan)
            except Exception:
                pass
            self._plan = None

    @property
    def maptype(self):
        """'I', 'QU' or 'IQU' (linearoperators.py:552-557)."""
        return self.__runcase

    @property
    def cos(self):
        return self._angles.cos

    @property
    def sin(self):
        return self._angles.sin

    def plan_info(self):
        """Sizes of the pixel-major plan (built here if no P^T has needed it yet)."""
        _hip.call("cm2_pointing_build_sell", self._plan, D.stream())
        info = (ctypes.c_int64 * 6)()
        _hip.call("cm2_pointing_info", self._plan, info)
        return dict(nt=info[0], npix=info[1], pol=info[2], nvalid=info[3],
                    padded_len=info[4], nslices=info[5])

    # -- P x -----------------------------------------------------------------------
    def _gather(self, v):
        x = D.f64(v)
        if x.numel() != self.pol * self.ncols:
            raise lp.ShapeError("map vector has %d entries, expected %d"
                                % (x.numel(), self.pol * self.ncols))
        out = D.empty(self.nrows)
        _hip.call("cm2_P_apply", self._plan, D.ptr(x), D.ptr(out), D.stream())
        return D.like_input(out, v)

    # -- P^T v ---------------------------------------------------------------------
    def _scatter(self, v):
        x = D.f64(v)
        if x.numel() != self.nrows:
            raise lp.ShapeError("time-domain vector has %d entries, expected %d"
                                % (x.numel(), self.nrows))
        out = D.empty(self.pol * self.ncols)
        _hip.call("cm2_Pt_apply", self._plan, D.ptr(x), D.ptr(out), D.stream())
        return D.like_input(out, v)

    # the reference's six method names, one gather/scatter pair per map type
    mult = mult_qu = mult_iqu = _gather
    rmult = rmult_qu = rmult_iqu = _scatter

    # -- fused P^T diag(w) P ---------------------------------------------------------
    def _attach_weights(self, noise):
        """Make `noise` (a constant-diagonal BlockLO, or None for N = I) the weight
        of the fused matvec."""
        key = id(noise) if noise is not None else 0
        if self._weights_of != key:
            w = None if noise is None else noise._device_diag()
            if w is not None and w.numel() != self.nrows:
                raise lp.ShapeError("noise operator has %d samples, pointing has %d"
                                    % (w.numel(), self.nrows))
            _hip.call("cm2_pointing_set_weights", self._plan, D.ptr(w), D.stream())
            self._weights_of = key
            self._weights_keepalive = noise

    def fused_normal_matvec(self, v, noise=None):
        """(P.T * N * P) * v in one pass over the samples, N constant-diagonal."""
        self._attach_weights(noise)
        x = D.f64(v)
        if x.numel() != self.pol * self.ncols:
            raise lp.ShapeError("map vector has %d entries, expected %d"
                                % (x.numel(), self.pol * self.ncols))
        out = D.empty(self.pol * self.ncols)
        _hip.call("cm2_PtNP_diag_apply", self._plan, D.ptr(x), D.ptr(out), D.stream())
        return D.like_input(out, v)


class _TileHandle(object):
    """Owns a cm2_tiles object (tile-bucketed copy of the pointing)."""

    def __init__(self, handle):
        self.h = handle
        info = self._info()
        self.nt, self.nvalid, self.tile_pixels, self.ntiles, self.nitems = [int(v) for v in info[:5]]
        self.half_angle = bool(info[5])
        self.pt_fixed = bool(info[6])
        self.pt_mode = int(info[6])              # 0 atomic, 1 fixed (hot runs chunked), 2 exact
        self.plan_id = int(info[7])
        self.nspans, self.span_samples = int(info[10]), int(info[11])

    def _info(self):
        info = (ctypes.c_int64 * 12)()
        _hip.call("cm2_tiles_info", self.h, info)
        return info

    def fixed_order_info(self):
        """(slice length, designed bytes per application) of the fixed-order P^T lists; zeros
        until the first P^T has built them."""
        info = self._info()
        return int(info[8]), int(info[9])

    def pt_parts(self):
        """Work items of the fixed-order P^T (after the lists exist): workgroups per application,
        tiles whose slices are shared out to several workgroups (uneven hit maps), bytes of the tile
        copies they write, simulated finish time over the ideal."""
        info = (ctypes.c_int64 * 4)()
        _hip.call("cm2_tiles_pt_parts", self.h, info)
        return {"workgroups": int(info[0]), "tiles_split": int(info[1]), "copy_bytes": int(info[2]),
                "simulated_finish_over_ideal": int(info[3]) / 1000.0}

    def pixel_range(self, tile_lo, tile_hi):
        """Pixels [p0, p1) covered by the tiles [tile_lo, tile_hi) (tiles are not of equal width when
        the plan balanced them for an uneven hit map)."""
        r = (ctypes.c_int64 * 2)()
        _hip.call("cm2_tiles_pixel_range", self.h, int(tile_lo), int(tile_hi), r)
        return int(r[0]), int(r[1])

    def set_pt_order(self, fixed):
        """True / 1: P^T sums every pixel in time order, hot pixels (more than 256 hits inside one
        slice of a tile) in fixed chunks of 32 terms (default: reproducible and independent of the
        hit map); "exact" / 2: pure time order for every pixel; False / 0: LDS atomics."""
        mode = 2 if fixed in ("exact", 2) else (1 if fixed else 0)
        _hip.call("cm2_tiles_set_pt_order", self.h, mode)
        self.pt_fixed = bool(mode)
        self.pt_mode = mode

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _hip.load().cm2_tiles_destroy(self.h)
            except Exception:
                pass
            self.h = None


def _sparse_tiles(P, tile_pixels=None, slice_samples=None):
    """Tile-bucketed plan of a SparseLO, built on first use (sort by pixel tile)."""
    if getattr(P, "_tiles", None) is None:
        if tile_pixels is None:
            if os.environ.get("CM2_PT_ORDER", "fixed") == "atomic":
                tile_pixels = {1: 4096, 2: 2048, 3: 2048}[P.pol]   # 32 / 32 / 48 KB of LDS per tile
            else:
                # fixed-order P^T: one workgroup (512 threads) owns a tile from its first sample to
                # its last and keeps the tile plus a 12 KB slice of the bucket in LDS, two workgroups
                # per CU: the chip takes 512 tiles at a time, so the number of tiles should be just
                # under a multiple of 512 (nside 256, IQU: 615 tiles of 1280 pixels cost P^T 0.53 ms
                # where 512 tiles of 1536 pixels cost 0.40; 384 tiles of 2048 pixels 0.52).  Largest
                # tile (longest address runs for the overlap-save kernel) that leaves 512 k tiles and
                # stays within 32 / 32 / 48 KB of LDS for I / QU / IQU.
                tp_max = {1: 4096, 2: 2048, 3: 2048}[P.pol]
                k = 1
                while True:
                    tile_pixels = -(-P.ncols // (512 * k))
                    tile_pixels = max(64, -(-tile_pixels // 64) * 64)
                    if tile_pixels <= tp_max:
                        break
                    k += 1
            if os.environ.get("CM2_TILE_PIXELS"):
                tile_pixels = int(os.environ["CM2_TILE_PIXELS"])
            while tile_pixels > 64 and tile_pixels // 2 >= P.ncols:
                tile_pixels //= 2
        if slice_samples is None:
            # work items of k_P_tiles: ~8192 items at any size (1e8 samples: 12 207-sample slices
            # 0.339-0.353 ms, 24 414 0.344-0.364, 6 144 0.386, 98 304 0.390)
            slice_samples = max(4096, min(65536, P.nrows // 8192))
            if os.environ.get("CM2_TILE_SLICE"):
                slice_samples = int(os.environ["CM2_TILE_SLICE"])
        h = ctypes.c_void_p()
        _hip.call("cm2_tiles_create", ctypes.byref(h), D.ptr(P._d_pix), D.ptr(P._d_cos),
                  D.ptr(P._d_sin), P.nrows, P.ncols, int(P.pol), int(tile_pixels),
                  int(slice_samples), D.stream())
        # the fixed-order P^T lists are built with the plan: applications then only launch kernels
        _hip.call("cm2_tiles_prepare_pt", h, D.stream())
        P._tiles = _TileHandle(h)
    return P._tiles


class _TiledNormalLO(_DeviceOp):
    """
    ``P^T N^-1 P`` for a noise operator with off-diagonal terms, on the tile-bucketed TOD
    order: LDS-staged gather -> time order -> N^-1 -> tile order -> LDS-staged scatter-add.
    Every HBM access is a coalesced stream.  Equal to the three separate stages up to
    the summation order inside P^T (LDS atomics), i.e. to rounding.
    """

    def __init__(self, P, noise):
        self.P, self.noise = P, noise
        self._fused_noise = None
        self._prepared_plan = None
        self._work = None
        n = P.pol * P.ncols
        super(_TiledNormalLO, self).__init__(n, n, self._mult, symmetric=True)

    def reduced_matvec(self, v, reducer, ngroups=4, out=None):
        """``sum over ranks of (P^T N^-1 P) v`` with the cross-rank reduction of a finished
        part of the map overlapped with the back-projection of the next part: P^T runs tile
        group by tile group and ``reducer(view)`` (an asynchronous in-place sum, returning an
        object with ``wait()``) is started on each group's slice of the output as soon as its
        kernel is queued.  Returns None for a host vector (the caller then reduces the whole
        result itself).  ``out``: write into this device vector instead of a new one."""
        if not D.is_dev(v):
            return None
        return self._mult(v, reducer=reducer, ngroups=int(ngroups), out=out)

    def matvec_into(self, v, out):
        """``out[:] = (P^T N^-1 P) v`` for device vectors, into memory the CALLER owns (a persistent
        exchange buffer of the sharded layouts, sharding.py: no allocation and no copy per matvec, the
        same addresses for the collective every time).  ``out``: contiguous float64 tensor of pol*npix."""
        if not (D.is_dev(v) and D.is_dev(out)) or out.numel() != self.shape[0] or not out.is_contiguous() \
                or out.dtype != D.torch.float64:
            raise lp.ShapeError("matvec_into needs device vectors of %d doubles" % self.shape[0])
        self._mult(v, out=out)
        return out

    def _mult(self, v, reducer=None, ngroups=1, out=None):
        P = self.P
        T = _sparse_tiles(P)
        x = D.f64(v)
        if x.numel() != P.pol * P.ncols:
            raise lp.ShapeError("map vector has %d entries, expected %d"
                                % (x.numel(), P.pol * P.ncols))
        st = D.stream()
        if self._work is None or self._work[0].numel() != max(T.nvalid, 1):
            # TOD-sized scratch is kept for the life of the operator (two buffers of nvalid
            # doubles): a PCG calls this every iteration and large hipMallocs are slow
            self._work = (D.empty(max(T.nvalid, 1)), D.empty(max(T.nvalid, 1)))
        d_tb, v_tb = self._work
        _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st)
        if self._fused_noise is None:
            info = getattr(self.noise, "noise_info", None)
            self._fused_noise = info is not None and info()["tiles_ok"]
        tiled_apply = getattr(self.noise, "_apply_tiles", None)
        if self._fused_noise:
            # overlap-save kernel reads and writes the tile order directly; its address lists for this
            # plan are built by the first application (later ones only launch the kernel)
            if self._prepared_plan != T.plan_id:
                _hip.call("cm2_noise_prepare_tiles", self.noise._noise.h, T.h, st)
                self._prepared_plan = T.plan_id
            _hip.call("cm2_noise_apply_tiles", self.noise._noise.h, T.h, D.ptr(d_tb),
                      D.ptr(v_tb), st)
            src = v_tb
        elif tiled_apply is not None and tiled_apply(T, d_tb, v_tb):
            src = v_tb              # the time-domain operator ran on the tile order itself
        else:
            tod = D.empty(P.nrows)
            _hip.call("cm2_tod_tiles_to_time", T.h, D.ptr(d_tb), D.ptr(tod), st)
            tod2 = self.noise._apply_all(tod)
            _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(tod2), D.ptr(d_tb), st)
            src = d_tb
        if out is None:
            out = D.empty(P.pol * P.ncols)
        if reducer is None:
            _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(src), D.ptr(out), st)
            return D.like_input(out, v)
        # groups of tiles with the same pixel boundaries on every rank (ranks whose hit maps differ
        # may have cut their tiles differently: the boundaries depend on npix and the tile size only)
        ngroups = max(1, min(int(ngroups), 8))
        cuts = (ctypes.c_int64 * (ngroups + 1))()
        _hip.call("cm2_tiles_group_tiles", T.h, ngroups, cuts)
        works = []
        for g in range(ngroups):
            lo, hi = int(cuts[g]), int(cuts[g + 1])
            if hi == lo:
                continue
            _hip.call("cm2_Pt_tiles_apply_range", T.h, D.ptr(src), D.ptr(out), lo, hi, st)
            p0, p1 = T.pixel_range(lo, hi)
            a, b = P.pol * p0, P.pol * p1
            works.append(reducer(out[a:b]))
        for w in works:
            if w is not None:
                w.wait()
        return out


#: "exact": P^T is always the fixed-order pixel-major reduction (bit-reproducible, equal to
#: the reference's serial loop).  "tiled": products P.T*N*P with a Toeplitz N run on the
#: tile-bucketed order (faster, P^T equal to rounding).  "auto": tiled from 2^20 samples up.
POINTING_MODE = os.environ.get("CM2_POINTING_MODE", "auto")


def set_pointing_mode(mode):
    global POINTING_MODE
    if mode not in ("exact", "tiled", "auto"):
        raise ValueError("pointing mode must be 'exact', 'tiled' or 'auto'")
    POINTING_MODE = mode


def _use_tiles(P):
    return POINTING_MODE == "tiled" or (POINTING_MODE == "auto" and P.nrows >= (1 << 20))


class _FusedNormalLO(_DeviceOp):
    """P^T diag(w) P as one operator (produced by the product-chain fusion)."""

    def __init__(self, P, noise):
        self.P, self.noise = P, noise
        n = P.pol * P.ncols
        super(_FusedNormalLO, self).__init__(n, n, lambda v: P.fused_normal_matvec(v, noise),
                                             symmetric=True)


@lp.register_chain_fusion
def _fuse_chain(chain):
    """[..., P.T, N(diag), P, ...] -> [..., fused, ...]  and  [P.T, P] likewise.
    The fused kernel performs the same operations in the same order as the three
    stages, so results are identical; only the HBM traffic changes."""
    out, i, changed = [], 0, False
    n = len(chain)
    while i < n:
        a = chain[i]
        P = getattr(a, "T", None) if getattr(a, "_is_transpose", False) else None
        if isinstance(P, SparseLO):
            if i + 2 < n and chain[i + 2] is P and isinstance(chain[i + 1], BlockLO) \
                    and not chain[i + 1].isoffdiag:
                out.append(_FusedNormalLO(P, chain[i + 1]))
                i += 3
                changed = True
                continue
            if i + 2 < n and chain[i + 2] is P and isinstance(chain[i + 1], BlockLO) \
                    and chain[i + 1].isoffdiag and _use_tiles(P):
                out.append(_TiledNormalLO(P, chain[i + 1]))
                i += 3
                changed = True
                continue
            if i + 2 < n and chain[i + 2] is P and isinstance(chain[i + 1], (FilterLO, GroundFilterLO)) \
                    and _use_tiles(P) and chain[i + 1]._tile_compatible(P):
                # A = P^T F P, the production operator: tile-order P / P^T around the
                # time-order filter (two streaming permutations instead of the exact-order
                # gather and fixed-order scatter)
                out.append(_TiledNormalLO(P, chain[i + 1]))
                i += 3
                changed = True
                continue
            if i + 1 < n and chain[i + 1] is P:
                out.append(_FusedNormalLO(P, None))
                i += 2
                changed = True
                continue
        out.append(a)
        i += 1
    return out if changed else None


# ================================================================== noise N^-1 ===
class _NoiseHandle(object):
    """Owns a cm2_noise object."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _hip.load().cm2_noise_destroy(self.h)
            except Exception:
                pass
            self.h = None


def _make_toeplitz(bands, sizes, method=0):
    bands = np.ascontiguousarray(bands, dtype=np.float64)
    nb, lam = bands.shape
    _, pb = _as_f64(bands.ravel())
    sz, ps = _as_i64(sizes)
    h = ctypes.c_void_p()
    _hip.call("cm2_noise_create_toeplitz", ctypes.byref(h), pb, int(lam), ps, int(nb),
              int(method), D.stream())
    return _NoiseHandle(h)


def _noise_apply(handle, nt, v):
    x = D.f64(v)
    if x.numel() != nt:
        raise lp.ShapeError("Multiplying with vector of wrong shape: %d samples expected, got %d"
                            % (nt, x.numel()))
    out = D.empty(nt)
    _hip.call("cm2_noise_apply", handle.h, D.ptr(x), D.ptr(out), D.stream())
    return D.like_input(out, v)


class ToeplitzLO(_DeviceOp):
    """
    Symmetric banded Toeplitz block (reference: linearoperators.py:560-602):
    ``ToeplitzLO(a, size)`` with ``a`` the first ``len(a)`` entries of the first row.
    ``mult`` reproduces the zero-boundary product of :582-595.  ``method`` selects
    the direct kernel (reference summation order) or rocFFT overlap-save.
    """

    def __init__(self, a, size, method=0):
        D.require_gpu()
        self.array = a
        self._size = int(size)
        band = np.atleast_1d(np.asarray(a, dtype=np.float64))
        self._noise = _make_toeplitz(band.reshape(1, -1), [self._size], method)
        super(ToeplitzLO, self).__init__(nargin=size, nargout=size, matvec=self.mult,
                                         symmetric=True)

    def mult(self, v):
        return _noise_apply(self._noise, self._size, v)


class BlockLO(blk.BlockDiagonalLinearOperator):
    """
    Inverse noise N^-1, block diagonal over stationary intervals (reference:
    linearoperators.py:627-697).  ``BlockLO(blocksize, t, offdiag=False)``:

    * ``offdiag=False``: ``t[b]`` is the constant diagonal of block ``b``;
      ``.diag`` is the per-sample weight array (:677-683);
    * ``offdiag=True``: ``t[b]`` is the band of a :class:`ToeplitzLO` block; ``.diag``
      is only ``t[0]``, as in the reference (:673, :683).

    ``blocksize`` may be an int (equal blocks) or a list of per-block sizes (the
    docstring's promise at :638-639, which the reference itself does not honour).
    All blocks are applied by one device call (blkop.py:195-206 loops in Python).
    """
    _device_ok = True

    def __init__(self, blocksize, t, offdiag=False, method=0):
        D.require_gpu()
        self.__isoffdiag = bool(offdiag)
        self.blocksize = blocksize
        self.covnoise = t
        nb = len(t)
        if np.ndim(blocksize) == 0:
            self._sizes = [int(blocksize)] * nb
        else:
            self._sizes = [int(b) for b in blocksize]
            if len(self._sizes) != nb:
                raise lp.ShapeError("blocksize lists %d blocks, t has %d" % (len(self._sizes), nb))
        self._nt = int(sum(self._sizes))
        self._method = method
        self._diag_host = None
        self._diag_dev = None
        self.build_blocks()
        super(BlockLO, self).__init__(None, _lazy=(self._make_blocklist, self._sizes),
                                      matvec=self._apply_all)

    def build_blocks(self):
        """Create the device-side operator (the reference builds one Python
        operator per block here, :655-683; those are made lazily by ``blocklist``)."""
        if self.isoffdiag:
            bands = [np.atleast_1d(np.asarray(b, dtype=np.float64)) for b in self.covnoise]
            lam = len(bands[0])
            if any(len(b) != lam for b in bands):
                raise lp.ShapeError("all Toeplitz bands must have the same length")
            self._noise = _make_toeplitz(np.vstack(bands), self._sizes, self._method)
        else:
            vals, pv = _as_f64([float(v) for v in self.covnoise])
            sz, ps = _as_i64(self._sizes)
            h = ctypes.c_void_p()
            _hip.call("cm2_noise_create_diag", ctypes.byref(h), pv, ps, len(self._sizes))
            self._noise = _NoiseHandle(h)

    def _make_blocklist(self):
        if self.isoffdiag:
            return [ToeplitzLO(b, s, self._method) for b, s in zip(self.covnoise, self._sizes)]
        return [lp.DiagonalOperator(np.full(s, float(v))) for v, s in zip(self.covnoise, self._sizes)]

    @property
    def blocklist(self):
        return self.blocks

    @property
    def isoffdiag(self):
        """Whether the operator has off-diagonal terms (:691-697)."""
        return self.__isoffdiag

    @property
    def diag(self):
        """diag(N^-1) as a NumPy array (built on first use)."""
        if self._diag_host is None:
            if self.isoffdiag:
                self._diag_host = np.asarray(self.covnoise[0], dtype=np.float64)
            else:
                self._diag_host = D.to_host(self._device_diag())
        return self._diag_host

    def _device_diag(self):
        if self.isoffdiag:
            raise RuntimeError("per-sample diagonal is only defined for offdiag=False")
        if self._diag_dev is None:
            self._diag_dev = D.empty(self._nt)
            _hip.call("cm2_noise_expand_diag", self._noise.h, D.ptr(self._diag_dev), D.stream())
        return self._diag_dev

    def noise_info(self):
        info = (ctypes.c_int64 * 6)()
        _hip.call("cm2_noise_info", self._noise.h, info)
        return dict(nt=info[0], nblocks=info[1], lam=info[2], method=info[3], fft_len=info[4],
                    tiles_ok=bool(info[5]))

    def tile_kernel_info(self):
        """Overlap-save kernel of the tile-order application: complex points per thread, list format of
        the most recently used tile plan, window length, designed HBM bytes per sample."""
        info = (ctypes.c_int64 * 4)()
        bps = ctypes.c_double(0.0)
        _hip.call("cm2_noise_tile_kernel_info", self._noise.h, info, ctypes.byref(bps))
        return dict(os_kernel="real%d" % info[0],
                    os_lists={0: "not built", 1: "plain", 2: "run-coded", 3: "inverse run-coded"}[int(info[1])],
                    os_window=int(info[2]), os_windows_across_spans=int(info[3]),
                    tile_bytes_per_sample=round(float(bps.value), 2))

    def _apply_all(self, v):
        return _noise_apply(self._noise, self._nt, v)


# ====================================================== per-pixel Stokes blocks ===
class _PixelWeights(object):
    """Device copies of the per-pixel sums held by a ProcessTimeSamples."""

    FIELDS = ("counts", "cosine", "sine", "cos2", "sin2", "sincos")

    def __init__(self, CES, n, pol):
        D.require_gpu()
        self.npix, self.pol = int(n), int(pol)
        need = {1: ("counts",), 2: ("cos2", "sin2", "sincos"), 3: self.FIELDS}[pol]
        dev_side = getattr(CES, "_dev_weights", None) or {}
        self.d = {}
        for k in self.FIELDS:
            if k in need:
                t = dev_side.get(k)
                self.d[k] = t if t is not None else D.f64(getattr(CES, k))
                if self.d[k].numel() != self.npix:
                    raise lp.ShapeError("CES.%s has %d pixels, expected n=%d"
                                        % (k, self.d[k].numel(), self.npix))
            else:
                self.d[k] = None

    def ptrs(self):
        return [D.ptr(self.d[k]) for k in self.FIELDS]


class _CESFields(object):
    """The reference's per-pixel attributes of the block-diagonal operators (``counts, cos, sin,
    cos2, sin2, sincos``: linearoperators.py:713-722, :762-771) as host arrays read from the CES
    object when they are first asked for: the operators themselves work on the device copies
    (``_PixelWeights``) and never need them."""

    _FIELD_OF = {1: {"counts": "counts"},
                 2: {"sin2": "sin2", "cos2": "cos2", "sincos": "sincos"},
                 3: {"sin2": "sin2", "cos2": "cos2", "sincos": "sincos", "counts": "counts",
                     "cos": "cosine", "sin": "sine"}}

    def __getattr__(self, name):
        d = self.__dict__
        src = self._FIELD_OF.get(d.get("pol"), {}).get(name)
        if src is None or "_ces" not in d:
            raise AttributeError(name)
        d[name] = getattr(d["_ces"], src)
        return d[name]


class BlockDiagonalLO(_CESFields, _DeviceOp):
    """
    Explicit per-pixel apply of ``P^T diag(N^-1) P`` (reference:
    linearoperators.py:700-746): pol=1 ``x*counts``; pol=2 / pol=3 the 2x2 / 3x3
    blocks ``[[hits,c,s],[c,c2,cs],[s,cs,s2]]``.
    """

    def __init__(self, CES, n, pol=1):
        self.size = pol * n
        self.pol = pol
        self.pixels = np.arange(n)
        self._w = _PixelWeights(CES, n, pol)
        self._ces = CES
        super(BlockDiagonalLO, self).__init__(nargin=self.size, nargout=self.size,
                                              matvec=self.mult, symmetric=True)

    def mult(self, x):
        xd = D.f64(x)
        out = D.empty(self.size)
        _hip.call("cm2_bd_apply", int(self.pol), self._w.npix, *(self._w.ptrs() + [
            D.ptr(xd), D.ptr(out), D.stream()]))
        return D.like_input(out, x)


class BlockDiagonalPreconditionerLO(_CESFields, _DeviceOp):
    """
    ``M_BD = (P^T diag(N^-1) P)^-1`` per pixel (reference: linearoperators.py:749-859):
    closed-form 1x1 / 2x2 / 3x3 inverse; pixels with ``|det| <= 1e-5`` (pol>1) or
    ``counts <= 0`` (pol=1) give 0 (:780, :789, :795, :821).  The determinant and
    mask are computed once at construction (the reference recomputes them per call).
    """

    def __init__(self, CES, n, pol=1):
        self.size = pol * n
        self.pixels = np.arange(n)
        self.pol = pol
        self._w = _PixelWeights(CES, n, pol)
        self._ces = CES
        self._d_det = D.empty(n)
        self._d_mask = D.empty(n, torch.uint8)
        _hip.call("cm2_bd_det_mask", int(pol), int(n), *(self._w.ptrs() + [
            D.ptr(self._d_det), D.ptr(self._d_mask), D.stream()]))
        super(BlockDiagonalPreconditionerLO, self).__init__(nargin=self.size, nargout=self.size,
                                                            matvec=self.mult, symmetric=True)

    def mult(self, x):
        xd = D.f64(x)
        out = D.empty(self.size)
        _hip.call("cm2_bdprecond_apply", int(self.pol), self._w.npix, *(self._w.ptrs() + [
            D.ptr(self._d_det), D.ptr(self._d_mask), D.ptr(xd), D.ptr(out), D.stream()]))
        return D.like_input(out, x)


# =================================================================== InverseLO ===
class InverseLO(lp.LinearOperator):
    """
    ``A^-1`` as an operator: ``mult(x)`` runs ``method(A, x, M=preconditioner)`` and
    returns the solution (reference: linearoperators.py:861-941).  ``method`` is any
    solver with the ``scipy.sparse.linalg.cg`` calling convention, e.g.
    :func:`cosmomap2_amd.cg` (device PCG).
    """

    def __init__(self, A, method=None, preconditioner=None):
        super(InverseLO, self).__init__(nargin=A.shape[0], nargout=A.shape[1], matvec=self.mult,
                                        symmetric=True)
        self.A = A
        self.__method = method
        self.__preconditioner = preconditioner
        self.__converged = None

    def mult(self, x):
        y, info = self.method(self.A, x, M=self.preconditioner)
        self.isconverged(info)
        return y

    def isconverged(self, info):
        self.__converged = info
        return info == 0

    method = property(lambda self: self.__method)
    converged = property(lambda self: self.__converged)
    preconditioner = property(lambda self: self.__preconditioner)


# ============================================================ deflation / coarse ===
def _dev_matrix(z):
    """n x r matrix as a row-major float64 tensor in HBM."""
    if D.is_tensor(z):
        t = D.f64(z)
    else:
        t = D.f64(np.ascontiguousarray(np.asarray(z, dtype=np.float64)))
    if t.dim() != 2:
        raise lp.ShapeError("deflation matrix must be 2-d (n x r)")
    return t


class DeflationLO(_DeviceOp):
    """
    Deflation operator Z (reference: linearoperators.py:1029-1065): ``Z*y`` is the
    combination of the r columns (:1041-1050), ``Z.T*x`` the r scalar products
    (:1051-1056).  Z lives row-major in HBM, read once per application.
    """

    def __init__(self, z):
        D.require_gpu()
        self._d_Z = _dev_matrix(z)
        self.nrows, self.ncols = int(self._d_Z.shape[0]), int(self._d_Z.shape[1])
        self._z_host = None if D.is_tensor(z) else np.asarray(z)
        super(DeflationLO, self).__init__(nargin=self.ncols, nargout=self.nrows,
                                          matvec=self.mult, symmetric=False, rmatvec=self.rmult)

    @property
    def z(self):
        """The columns of Z as a list of arrays (:1059-1062)."""
        if self._z_host is None:
            self._z_host = D.to_host(self._d_Z)
        return [self._z_host[:, j] for j in range(self.ncols)]

    def mult(self, x):
        y = D.f64(x)
        if y.numel() != self.ncols:
            raise lp.ShapeError("coefficient vector has %d entries, expected r=%d"
                                % (y.numel(), self.ncols))
        out = D.empty(self.nrows)
        _hip.call("cm2_Z_apply", self.nrows, self.ncols, D.ptr(self._d_Z), D.ptr(y), D.ptr(out),
                  D.stream())
        return D.like_input(out, x)

    def rmult(self, x):
        xd = D.f64(x)
        if xd.numel() != self.nrows:
            raise lp.ShapeError("map vector has %d entries, expected %d" % (xd.numel(), self.nrows))
        out = D.empty(self.ncols)
        _hip.call("cm2_Zt_apply", self.nrows, self.ncols, D.ptr(self._d_Z), D.ptr(xd), D.ptr(out),
                  D.ptr(D.reduce_work()), D.stream())
        return D.like_input(out, x)


class CoarseLO(_DeviceOp):
    """
    Coarse operator, always applied as ``E^-1`` (reference: linearoperators.py:946-1027).
    ``E = Z^T (A Z)`` is contracted on the GPU (fp64 MFMA panels when r is a multiple of
    16), brought to the host (r x r) and factorised there exactly like the reference:
    ``apply='LU'`` -> ``lu(E, permute_l=True)`` and two solves (:975-976, :1025);
    ``apply='eig'`` -> ``eigh``, eigenvalues with ``|lambda/lambda_max| < 1e-6`` dropped
    (:994-1015).  Vectors already in HBM are multiplied by the explicit r x r inverse.
    """

    def __init__(self, Z, Az, r, apply='LU', allreduce=None):
        D.require_gpu()
        dZ, dAZ = _dev_matrix(Z), _dev_matrix(Az)
        n, rz = int(dZ.shape[0]), int(dZ.shape[1])
        if tuple(dAZ.shape) != (n, rz):
            raise lp.ShapeError("Z is %r, Az is %r" % (tuple(dZ.shape), tuple(dAZ.shape)))
        dE = D.empty(rz * rz)
        work = D.empty(int(_hip.load().cm2_gemm_tn_work_doubles(rz, rz)))
        _hip.call("cm2_gemm_tn", n, rz, rz, D.ptr(dZ), D.ptr(dAZ), D.ptr(dE), D.ptr(work),
                  D.stream())
        if allreduce is not None:             # Z, Az hold this rank's rows only (sharding.py)
            allreduce(dE)
        M = D.to_host(dE).reshape(rz, rz)                      # :1019  dgemm(Z, Az.T)
        self.E = M.copy()
        self.r = rz
        self._apply_kind = apply
        if apply == 'eig':
            self.setting_inverse_w_eigenvalues(M)
            inv = self.invE
            mv = self.mult_eig
        elif apply == 'LU':
            self.L, self.U = sla.lu(M, permute_l=True, check_finite=False)
            inv = sla.solve(self.U, sla.solve(self.L, np.eye(rz)))
            mv = self.mult
        else:
            raise ValueError("apply must be 'LU' or 'eig', got %r" % (apply,))
        self._d_inv = D.f64(np.ascontiguousarray(inv))
        super(CoarseLO, self).__init__(nargin=r, nargout=r, matvec=mv, symmetric=True)

    def _device_mult(self, v):
        out = D.empty(self.r)
        _hip.call("cm2_small_matvec", self.r, D.ptr(self._d_inv), D.ptr(D.f64(v)), D.ptr(out),
                  D.stream())
        return D.like_input(out, v)

    def mult(self, v):
        """x = E^-1 v through L and U (:969-977)."""
        if D.is_tensor(v):
            return self._device_mult(v)
        y = sla.solve(self.L, v)
        return sla.solve(self.U, y)

    def mult_eig(self, v):
        """x = E^+ v with the eigen-decomposed pseudo-inverse (:979-984)."""
        if D.is_tensor(v):
            return self._device_mult(v)
        return self.invE.dot(v)

    def setting_inverse_w_eigenvalues(self, E):
        """Pseudo-inverse of E keeping eigenvalues with |lambda/lambda_max| > 1e-6
        (:986-1015)."""
        eigenvals, W = sla.eigh(E)
        lambda_max = max(eigenvals)
        diags = eigenvals * 0.
        keep = np.where(abs(eigenvals / lambda_max) > 1.e-6)[0]
        self.n_discarded = len(eigenvals) - len(keep)
        diags[keep] = 1. / eigenvals[keep]
        self.invE = (W * diags).dot(W.T)


class TwoLevelPreconditionerLO(_DeviceOp):
    """
    ``M2 = Mbd (I - AZ E^-1 Z^T) + Z E^-1 Z^T`` as one operator (the composition of
    src/test_M2_precond_onto_real_data.py:98-112 and
    tests/test_2level_preconditioner.py:45-48), with ``y = E^-1 Z^T r`` evaluated
    once: Z^T r (one pass over Z), the r x r solve, then one fused pass over Z and AZ
    that also applies the per-pixel M_BD block.
    """

    def __init__(self, Mbd, Zd, AZd, E, allreduce=None):
        if not isinstance(Mbd, BlockDiagonalPreconditionerLO):
            raise TypeError("Mbd must be a BlockDiagonalPreconditionerLO")
        self.Mbd, self.Zd, self.AZd, self.E = Mbd, Zd, AZd, E
        self._allreduce = allreduce           # row-sharded vectors: Z^T r summed over ranks
        n = Mbd.size
        if Zd.nrows != n or AZd.nrows != n or Zd.ncols != AZd.ncols:
            raise lp.ShapeError("Z / AZ shapes do not match the map size %d" % n)
        super(TwoLevelPreconditionerLO, self).__init__(nargin=n, nargout=n, matvec=self.mult,
                                                       symmetric=True)

    def mult(self, x):
        res = D.f64(x)
        y0 = self.Zd.rmult(res)
        if self._allreduce is not None:
            self._allreduce(y0)
        y = self.E._device_mult(y0)
        out = D.empty(self.Mbd.size)
        w = self.Mbd._w
        _hip.call("cm2_m2_finish", int(self.Mbd.pol), w.npix, self.Zd.ncols,
                  D.ptr(self.Zd._d_Z), D.ptr(self.AZd._d_Z), D.ptr(y), D.ptr(res),
                  *(w.ptrs() + [D.ptr(self.Mbd._d_det), D.ptr(self.Mbd._d_mask), D.ptr(out),
                                D.stream()]))
        return D.like_input(out, x)


# ==================================================================== FilterLO ===
def filter_plan(subscans, tstart, nsamples, nbolos, poly_order, legendres):
    """Host-side set-up of :class:`FilterLO` (index arithmetic only, no TOD data).

    Returns ``(starts, lens, table_off, table)`` as cm2_filter_create wants them.
    Chunks follow the loops of linearoperators.py:134-140 (CES -> detector pair ->
    sub-scan) and are then put in ascending order.  For poly_order > 0 ``table`` is the
    concatenation of one Legendre block per distinct chunk length (``legendres[n]``,
    :206-213) and ``table_off[s]`` the offset of chunk s's block.
    """
    starts, lens = [], []
    offset = 0
    for subsc, ts, ns, nb in zip(subscans, tstart, nsamples, nbolos):
        sub = np.asarray(subsc, dtype=np.int64).reshape(-1)
        t0 = np.asarray(ts, dtype=np.int64).reshape(-1)
        if sub.size != t0.size:
            raise lp.ShapeError("%d sub-scan sizes but %d sub-scan starts" % (sub.size, t0.size))
        bolo = np.arange(int(nb), dtype=np.int64)[:, None] * int(ns) + offset
        starts.append((t0[None, :] + bolo).reshape(-1))
        lens.append(np.tile(sub, int(nb)))
        offset += int(nb) * int(ns)
    starts = np.concatenate(starts) if starts else np.zeros(0, dtype=np.int64)
    lens = np.concatenate(lens) if lens else np.zeros(0, dtype=np.int64)
    order = np.argsort(starts, kind="stable")
    starts, lens = np.ascontiguousarray(starts[order]), np.ascontiguousarray(lens[order])
    if poly_order == 0:
        return starts, lens, None, None

    K = poly_order + 1
    table_off = np.zeros(starts.size, dtype=np.int64)
    blocks, at = [], 0
    for n in np.unique(lens):
        n = int(n)
        if n == 0:
            continue
        L = np.ascontiguousarray(legendres[n], dtype=np.float64)
        if L.shape != (n, K):
            raise lp.ShapeError("Legendre table for length %d has shape %r" % (n, L.shape))
        blocks.append(L.reshape(-1))
        table_off[lens == n] = at
        at += n * K
    table = np.concatenate(blocks) if blocks else np.zeros(0)
    return starts, lens, table_off, table


class _FilterHandle(object):
    def __init__(self, handle, keep):
        self.h = handle
        self.keep = keep                   # tensors the C side borrows

    def __del__(self):
        if self.h:
            try:
                _hip.load().cm2_filter_destroy(self.h)
            except Exception:
                pass
            self.h = None


class FilterLO(_DeviceOp):
    """
    Sub-scan filter of a time stream (reference: linearoperators.py:94-283).

    ``FilterLO(size, subscan_nsample, samples_per_bolopair, bolos_per_ces, pix_samples,
    poly_order=0, npool=4)``: ``subscan_nsample = [sizes, starts]`` of the sub-scans of
    one detector pair (lists of such arrays, with lists of ``samples_per_bolopair`` /
    ``bolos_per_ces``, for several CES, :263-273).  ``poly_order=0`` subtracts the mean
    of the unflagged samples of each chunk (``mult``, :129-168); ``poly_order>0``
    removes the Legendre polynomials up to that order (``polyfilter_multithreads`` ->
    ``globalprocsfilter``, :246-261, :286-322).  Flagged samples (pixel < 0) do not
    enter the fit.  ``npool`` sized the reference's multiprocessing pool and is
    ignored: one wavefront per chunk does the work.

    The flags are snapshotted at construction (the reference re-reads ``pix_samples``
    on every call).  Chunks must not overlap.
    """

    def __init__(self, size, subscan_nsample, samples_per_bolopair, bolos_per_ces, pix_samples,
                 poly_order=0, npool=4):
        self.n = int(size)
        self.nsamples = samples_per_bolopair
        self.nbolos = bolos_per_ces
        self.subscans = subscan_nsample[0]
        self.tstart = subscan_nsample[1]
        if not (type(self.nsamples) is list):
            self.nsamples = [self.nsamples]
            self.nbolos = [self.nbolos]
            self.subscans = [self.subscans]
            self.tstart = [self.tstart]
        self.pixels = pix_samples
        self.poly_order = int(poly_order)
        if self.poly_order < 0:
            raise ValueError("poly_order must be >= 0, got %r" % (poly_order,))
        if self.poly_order > 7:
            raise ValueError("poly_order up to 7 is supported, got %d" % self.poly_order)
        D.require_gpu()
        self._d_pix = D.i32(pix_samples)
        if self._d_pix.numel() != self.n:
            raise lp.ShapeError("pix_samples has %d entries, expected size=%d"
                                % (self._d_pix.numel(), self.n))
        self._handles = {}
        if self.poly_order > 0:
            self.compute_legendres()
        mv = self.mult if self.poly_order == 0 else self.polyfilter_multithreads
        self._plan_for(self.poly_order)
        super(FilterLO, self).__init__(nargin=self.n, nargout=self.n, matvec=mv, symmetric=False)

    def compute_legendres(self):
        """``self.legendres = {chunk length: normalised Legendre table}`` (:206-213)."""
        from ..utilities.linear_algebra_funcs import get_legendre_polynomials
        sizes = []
        for array in self.subscans:
            for i in np.asarray(array).reshape(-1):
                if int(i) not in sizes:
                    sizes.append(int(i))
        self.legendres = {n: get_legendre_polynomials(self.poly_order, n) for n in sizes}

    def _plan_for(self, order):
        h = self._handles.get(order)
        if h is not None:
            return h
        starts, lens, toff, table = filter_plan(
            self.subscans, self.tstart, self.nsamples, self.nbolos, order,
            getattr(self, "legendres", None))
        if starts.size and (starts[0] < 0 or (starts + lens).max() > self.n):
            raise lp.ShapeError("a sub-scan chunk lies outside the %d samples" % self.n)
        p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        handle = ctypes.c_void_p()
        _hip.call("cm2_filter_create", ctypes.byref(handle), self.n, int(starts.size), p(starts),
                  p(lens), D.ptr(self._d_pix), int(order), p(toff), p(table),
                  0 if table is None else int(table.size), D.stream())
        h = self._handles[order] = _FilterHandle(handle, self._d_pix)
        return h

    def filter_info(self):
        info = (ctypes.c_int64 * 7)()
        _hip.call("cm2_filter_info", self._plan_for(self.poly_order).h, info)
        return dict(nt=info[0], nchunks=info[1], poly_order=info[2], covered=info[3],
                    skipped=info[4], unflagged=info[5], flagged=info[6])

    def _apply(self, order, d):
        x = D.f64(d)
        if x.numel() != self.n:
            raise lp.ShapeError("time-domain vector has %d entries, expected %d"
                                % (x.numel(), self.n))
        out = D.empty(self.n)
        _hip.call("cm2_filter_apply", self._plan_for(order).h, D.ptr(x), D.ptr(out), D.stream())
        return D.like_input(out, d)

    def mult(self, d):
        """Offset removal per chunk (:129-168), whatever ``poly_order`` is."""
        return self._apply(0, d)

    def _apply_all(self, d):
        """The operator's own matvec on a device vector (used by the tile-order chain)."""
        return self._apply(self.poly_order, d)

    def _tile_compatible(self, P):
        """The tile order has no slot for the samples ``P`` flags; the filter may run there only
        if it flags exactly the same samples (the reference passes ``P.pairs``)."""
        cache = self.__dict__.get("_flag_match")
        if cache is None or cache[0]() is not P:          # weak reference: an id() can be reused
            a, b = self._d_pix, P._d_pix
            ok = a.numel() == b.numel() and (
                a.data_ptr() == b.data_ptr() or bool(torch.equal(a < 0, b < 0)))
            cache = self._flag_match = (weakref.ref(P), ok)
        return cache[1]

    def _apply_tiles(self, T, d_in_tb, d_out_tb):
        """Filter a TOD held in the tile-bucketed order ``T``; False if the chunks do not fit
        the windowed kernel (the caller then goes through the time order)."""
        done = ctypes.c_int(0)
        _hip.call("cm2_filter_apply_tiles", self._plan_for(self.poly_order).h, T.h, D.ptr(d_in_tb),
                  D.ptr(d_out_tb), ctypes.byref(done), D.stream())
        return bool(done.value)

    def polyfilter(self, d):
        """Legendre filtering up to ``poly_order`` (:170-204)."""
        if self.poly_order == 0:
            raise AttributeError("polyfilter needs poly_order > 0 (no Legendre tables were built)")
        return self._apply(self.poly_order, d)

    polyfilter_multithreads = polyfilter


# ============================================================== GroundFilterLO ===
class GroundFilterLO(_DeviceOp):
    """
    Ground-template filter ``v - G (G^T G)^-1 G^T v`` (reference: linearoperators.py:24-61):
    ``ground[t]`` is the azimuth bin of sample t (-1 = none), ``G`` the pol=1 pointing
    onto ``max(ground)+1`` bins and ``(G^T G)^-1`` the inverse hit count per bin (empty
    bins give 0, :788-790).
    """

    def counts_in_groundbins(self, g):
        """Samples per ground bin (:26-46), as the GPU scatter of a vector of ones."""
        ones = D.empty(self.n)
        ones.fill_(1.0)
        return D.to_host(self._bin_sums(ones))

    LDS_BINS = 8192          # cm2_ground_bin_sums keeps one histogram per workgroup in LDS

    def _bin_sums(self, x):
        """G^T x: LDS histogram when the bins fit, else the pixel-major P^T of SparseLO."""
        if self.nbins > self.LDS_BINS:
            return self._G.rmult(x)
        sums = D.empty(self.nbins)
        _hip.call("cm2_ground_bin_sums", self.n, self.nbins, D.ptr(self._G._d_pix), D.ptr(x),
                  D.ptr(sums), D.stream())
        return sums

    def mult(self, v):
        x = D.f64(v)
        if x.numel() != self.n:
            raise lp.ShapeError("time-domain vector has %d entries, expected %d"
                                % (x.numel(), self.n))
        binned = self._invGtG.mult(self._bin_sums(x))
        out = D.empty(self.n)
        _hip.call("cm2_ground_subtract", self.n, D.ptr(self._G._d_pix), D.ptr(binned), D.ptr(x),
                  D.ptr(out), D.stream())
        return D.like_input(out, v)

    _apply_all = mult

    def _tile_compatible(self, P):
        return self.n == P.nrows

    def _apply_tiles(self, T, d_in_tb, d_out_tb):
        """The same filter on a TOD held in the tile-bucketed order ``T``: binning and
        subtraction do not care about the order of the samples, only the bin labels have to
        be carried over once.  Flagged-pixel samples are absent there; they hold 0 in the
        chain P^T (.) P, so the bin sums are the same."""
        if self.nbins > self.LDS_BINS or T.nt != self.n:
            return False
        cache = getattr(self, "_bin_tb", None)
        if cache is None or cache[0] is not T:
            lab = D.empty(max(T.nvalid, 1), torch.int32)
            _hip.call("cm2_i32_time_to_tiles", T.h, D.ptr(self._G._d_pix), D.ptr(lab), D.stream())
            cache = self._bin_tb = (T, lab)
        lab = cache[1]
        st = D.stream()
        sums = D.empty(self.nbins)
        _hip.call("cm2_ground_bin_sums", T.nvalid, self.nbins, D.ptr(lab), D.ptr(d_in_tb),
                  D.ptr(sums), st)
        binned = self._invGtG.mult(sums)
        _hip.call("cm2_ground_subtract", T.nvalid, D.ptr(lab), D.ptr(binned), D.ptr(d_in_tb),
                  D.ptr(d_out_tb), st)
        return True

    def __init__(self, ground):
        g = ground.detach().cpu().numpy() if D.is_tensor(ground) else np.asarray(ground)
        self.nbins = int(g.max()) + 1
        self.n = len(g)
        self._G = G = SparseLO(self.nbins, self.n, ground)
        G.counts = self.counts_in_groundbins(ground)
        self._invGtG = invGtG = BlockDiagonalPreconditionerLO(G, self.nbins)
        self.Pg = G * invGtG * G.T
        super(GroundFilterLO, self).__init__(nargin=self.n, nargout=self.n, matvec=self.mult,
                                             symmetric=True)
