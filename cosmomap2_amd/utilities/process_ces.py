"""
ProcessTimeSamples: one-off preprocessing of a pointing stream (reference:
utilities/process_ces.py:20-555) -- per-pixel weight sums, the mask of unobserved
or badly conditioned pixels, compaction of the pixel numbering and flagging of the
samples that fall in removed pixels.  All of it runs on the GPU:

* weight sums: per-pixel fixed-order reduction over a stable (pixel -> samples) sort
  (cm2_weights_accumulate) instead of the serial scatter loops at :480-539;
* mask: the NumPy lines :491 / :544-555 as one kernel (cm2_pixel_mask);
* compaction: prefix sum (cm2_pixel_compact) instead of the O(Nold*Nm) search at
  :205-228, same ``old2new`` and compacted arrays;
* flagging: cm2_flag_samples (:411-418); the caller's ``pixs`` array is then updated
  IN PLACE, as the reference does, because the same object is handed to SparseLO.

``cos``/``sin`` of host angles are evaluated with NumPy exactly as :493-494 so that they
match the reference bit for bit; angles already in HBM use the device kernel.
"""
import ctypes

import numpy as np

from .. import _hip
from .. import device as D
from .utilities_functions import bash_colors, is_sorted

torch = D.torch

__all__ = ["ProcessTimeSamples"]

_FIELDS = ("counts", "cosine", "sine", "cos2", "sin2", "sincos")
_NEED = {1: ("counts",), 2: ("cos2", "sin2", "sincos"), 3: _FIELDS}


class _NewPixel(tuple):
    """``(npix, obspix)`` whose second element is produced on first access."""

    def __new__(cls, npix, owner):
        self = tuple.__new__(cls, (npix, None))
        self._owner = owner
        return self

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self)[i]
        return self._owner.obspix if i in (1, -1) else tuple.__getitem__(self, i)

    def __iter__(self):
        yield tuple.__getitem__(self, 0)
        yield self._owner.obspix

    def __eq__(self, other):
        return tuple(iter(self)) == tuple(other)

    __hash__ = None


class ProcessTimeSamples(object):
    """
    ``ProcessTimeSamples(pixs, npix, obspix=None, pol=1, phi=None, w=None, ground=None,
    threshold_cond=1.e3, obspix2=None)`` -- signature of process_ces.py:58.

    Attributes as in the reference: ``counts, cosine, sine, cos2, sin2, sincos`` (per
    compacted pixel), ``cos, sin`` (per sample), ``mask``, ``old2new``, ``obspix``,
    ``nsamples``, ``oldnpix``, and the property ``get_new_pixel -> (npix', obspix)``.
    Array attributes are NumPy arrays fetched from HBM on first access.

    ``allreduce``: optional callable applied in place to each per-pixel sum before the
    mask is formed (sums over TOD shards on other GPUs, see sharding.py).
    """

    def __init__(self, pixs, npix, obspix=None, pol=1, phi=None, w=None, ground=None,
                 threshold_cond=1.e3, obspix2=None, allreduce=None):
        D.require_gpu()
        if pol not in (1, 2, 3):
            raise RuntimeError("No valid polarization key set!\t=>\tpol=%r" % (pol,))
        self.pixs = pixs
        self.oldnpix = int(npix)
        self.nsamples = len(pixs)
        self.pol = pol
        self.bashc = bash_colors()
        self.threshold = threshold_cond
        self._allreduce = allreduce
        self._host = {}
        # (obspix=None: the reference sizes the default by samples, :67-68; the pixel count is what
        # the compaction indexes, so arange(npix) is what stands in -- materialised on first use)
        self._obspix = None if obspix is None else np.asarray(obspix)
        self._obspix_pending = False
        if ground is not None:                             # :70-73
            neg = np.asarray(ground) < 0
            ground[neg] = -1
            pixs[neg] = -1
        if D.is_dev(pixs) and pixs.dtype == torch.int32 and pixs.is_contiguous():
            self._d_pix = pixs                             # flagged in place, in HBM
        else:
            self._d_pix = D.i32(pixs)                      # device copy (int32)
        self._d_w = None if w is None else D.f64(w)
        self._d_cos = self._d_sin = None
        if pol > 1:
            if phi is None:
                raise RuntimeError("pol=%d needs the polarisation angles phi" % pol)
            if D.is_tensor(phi):
                dphi = D.f64(phi)
                self._d_cos, self._d_sin = D.empty(self.nsamples), D.empty(self.nsamples)
                _hip.call("cm2_cos_sin_2phi", self.nsamples, D.ptr(dphi), D.ptr(self._d_cos),
                          D.ptr(self._d_sin), D.stream())
            else:
                self._host["cos"] = np.cos(2. * np.asarray(phi))      # :493-494
                self._host["sin"] = np.sin(2. * np.asarray(phi))
                self._d_cos = D.f64(self._host["cos"])
                self._d_sin = D.f64(self._host["sin"])
        if obspix2 is None:
            self.initializeweights()
            self.new_repixelization()
            self.flagging_samples()
        else:
            self.SetObspix(obspix2)
            self.flagging_samples()
            self.compute_arrays()
        if ground is not None:                             # :85-89
            flagged = np.asarray(D.to_host(self._d_pix)) == -1
            ground[flagged] = -1
            self.ground = ground

    # ------------------------------------------------------------------ helpers ---
    def _accumulate(self, npix):
        out = {k: (D.empty(npix) if k in _NEED[self.pol] else None) for k in _FIELDS}
        _hip.call("cm2_weights_accumulate", int(self.pol), self.nsamples, int(npix),
                  D.ptr(self._d_pix), D.ptr(self._d_w), D.ptr(self._d_cos), D.ptr(self._d_sin),
                  *([D.ptr(out[k]) for k in _FIELDS] + [D.stream()]))
        if self._allreduce is not None:
            for k in _NEED[self.pol]:
                self._allreduce(out[k])
        return out

    # Host copies of per-pixel results (mask, old2new, obspix) are made when they are first read:
    # the device-resident pipeline (SparseLO, M_BD, the tile plan) never needs them, and at nside
    # 256 fetching and re-indexing them costs more than the kernels that produced them.
    @property
    def obspix(self):
        if self._obspix is None:
            self._obspix = np.arange(self.oldnpix)
        if self._obspix_pending:
            self._obspix = self._obspix[:self.oldnpix][D.to_host(self._d_keep) != 0]
            self._obspix_pending = False
        return self._obspix

    @obspix.setter
    def obspix(self, value):
        self._obspix = np.asarray(value)
        self._obspix_pending = False

    def __getattr__(self, name):
        if name == "mask":                                  # :491, :544-555
            keep = self.__dict__.get("_d_keep")
            if keep is None:
                raise AttributeError(name)
            self.__dict__["mask"] = np.where(D.to_host(keep) != 0)[0]
            return self.__dict__["mask"]
        if name == "old2new":
            o2n = self.__dict__.get("_d_old2new")
            if o2n is None:
                raise AttributeError(name)
            self.__dict__["old2new"] = D.to_host(o2n).astype(np.int64)
            return self.__dict__["old2new"]
        # lazily fetched NumPy views of device-side results
        if name in _FIELDS:
            dw = self.__dict__.get("_dev_weights", {})
            if dw.get(name) is None:
                raise AttributeError(name)
            host = self.__dict__["_host"]
            if name not in host:
                host[name] = D.to_host(dw[name])
            return host[name]
        if name in ("cos", "sin"):
            t = self.__dict__.get("_d_" + name)
            if t is None:
                raise AttributeError(name)
            host = self.__dict__["_host"]
            if name not in host:
                host[name] = D.to_host(t)
            return host[name]
        raise AttributeError(name)

    @property
    def get_new_pixel(self):
        """(number of pixels kept, their external ids) -- process_ces.py:90-92.  The ids are
        fetched from the device when the second element is read."""
        return _NewPixel(self.__new_npix, self)

    # ----------------------------------------------------------- reference steps ---
    def initializeweights(self, phi=None, w=None):
        """Weight sums on the original pixelisation and the pixel mask (:426-555)."""
        sums = self._accumulate(self.oldnpix)
        self._raw_weights = sums
        keep = D.empty(self.oldnpix, torch.uint8)
        _hip.call("cm2_pixel_mask", int(self.pol), self.oldnpix, D.ptr(sums["counts"]),
                  D.ptr(sums["cos2"]), D.ptr(sums["sin2"]), D.ptr(sums["sincos"]),
                  float(self.threshold), D.ptr(keep), D.stream())
        self._d_keep = keep
        self.__dict__.pop("mask", None)

    def new_repixelization(self):
        """Compact the pixel numbering to the masked set, preserving order (:192-349)."""
        o2n = D.empty(self.oldnpix, torch.int32)
        newn = ctypes.c_int64(0)
        _hip.call("cm2_pixel_compact", self.oldnpix, D.ptr(self._d_keep), D.ptr(o2n),
                  ctypes.byref(newn), D.stream())
        self._d_old2new = o2n
        self.__new_npix = int(newn.value)
        self.__dict__.pop("old2new", None)
        dw = {}
        for k in _FIELDS:
            src = self._raw_weights[k]
            if src is None:
                dw[k] = None
                continue
            dst = D.empty(self.__new_npix)
            if self.__new_npix:
                _hip.call("cm2_compact_f64", self.oldnpix, D.ptr(o2n), D.ptr(src), D.ptr(dst),
                          D.stream())
            dw[k] = dst
        self._dev_weights = dw
        self._raw_weights = None
        self._host = {k: v for k, v in self._host.items() if k in ("cos", "sin")}
        self._obspix_pending = True                         # compacted when first read
        self.n_removed_pix = self.oldnpix - self.__new_npix

    repixelization = new_repixelization

    def SetObspix(self, new_obspix):
        """Adopt an externally given pixel set (:94-111): ``old2new`` maps the old
        ``obspix`` onto positions in ``new_obspix``."""
        new_obspix = np.asarray(new_obspix)
        old2new = np.full(self.oldnpix, -1, dtype=np.int32)
        if not (is_sorted(self.obspix) and is_sorted(new_obspix)):
            order = np.argsort(self.obspix, kind='quicksort')
            self.obspix = self.obspix[order]
        idx = np.searchsorted(self.obspix, new_obspix)
        old2new[idx] = np.arange(len(idx), dtype=np.int32)
        self.old2new = old2new.astype(np.int64)
        self._d_old2new = D.i32(old2new)
        self.obspix = new_obspix
        self.__new_npix = len(new_obspix)

    def flagging_samples(self):
        """pixs[i] = old2new[pixs[i]] (-1 stays) on the device, then written back into
        the caller's array in place (:403-425)."""
        _hip.call("cm2_flag_samples", self.nsamples, D.ptr(self._d_pix), D.ptr(self._d_old2new),
                  D.stream())
        if isinstance(self.pixs, np.ndarray):
            self.pixs[:] = D.to_host(self._d_pix)
        elif D.is_tensor(self.pixs) and self.pixs is not self._d_pix:
            self.pixs.copy_(self._d_pix.to(self.pixs.device, self.pixs.dtype))

    def compute_arrays(self, phi=None, w=None):
        """Weight sums on the already compacted pixelisation (:113-189)."""
        self._dev_weights = self._accumulate(self.__new_npix)
        self._host = {k: v for k, v in self._host.items() if k in ("cos", "sin")}
