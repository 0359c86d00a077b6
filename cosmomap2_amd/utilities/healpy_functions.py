"""
Map-domain helpers on either side of the solve, with the reference's names and argument
order: ``obspix2mask`` and ``reorganize_map`` (utilities/healpy_functions.py:22-102) and
``full2cutskymap`` (utilities/IOfiles.py:377-393).  The scatter / gather between the
solution vector ``[I0,Q0,U0, I1,...]`` over the observed pixels and full-sky HEALPix maps
runs on the GPU (cm2_cutsky_to_fullsky / cm2_fullsky_to_cutsky).

healpy is not a dependency: ``nside2npix`` is ``12 * nside**2`` (HEALPix definition) and the
``fname`` arguments write the maps with :func:`cosmomap2_amd.utilities.healpix_fits.write_map`
(the FITS binary table ``hp.write_map`` produces; ``read_map`` reads such files back).
"""
from .. import _hip
from .. import device as D

torch = D.torch

from .healpix_fits import read_map, write_map                # noqa: F401  (hp.read_map / hp.write_map)

__all__ = ["obspix2mask", "reorganize_map", "full2cutskymap", "nside2npix", "read_map", "write_map"]


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def _write_fits(fname, maps):
    """hp.write_map(fname, maps) of the reference (:46, :103)."""
    if fname is not None:
        from .healpix_fits import write_map
        host = [D.to_host(m) if D.is_tensor(m) else m for m in maps]
        write_map(fname, host[0] if len(host) == 1 else host)


def _i64(a):
    return D.to_dev(a, torch.int64)


def obspix2mask(obspix, nside, fname=None):
    """Binary full-sky mask, 1 on the observed pixels (reference :22-45); with ``fname`` it is
    also written as a HEALPix FITS file (:46)."""
    n = int(_i64(obspix).numel())
    ones = D.empty(n)
    ones.fill_(1.0)
    mask = reorganize_map(ones, obspix, n, nside, 1)[0]
    _write_fits(fname, [mask])
    return mask if D.is_tensor(obspix) else D.to_host(mask)


def reorganize_map(mapin, obspix, npix, nside, pol, fname=None):
    """Solution vector -> list of ``pol`` full-sky HEALPix maps (I / Q,U / I,Q,U), zero
    outside the observed pixels (reference :47-102).  NumPy in, NumPy out; HBM tensor in,
    HBM tensors out.  With ``fname`` the maps are also written as a HEALPix FITS file (:103)."""
    if pol not in (1, 2, 3):
        raise RuntimeError("No valid polarization key set!\t=>\tpol=%r" % (pol,))
    D.require_gpu()
    x = D.f64(mapin).reshape(-1)
    op = _i64(obspix).reshape(-1)
    npix = int(npix)
    if op.numel() != npix or x.numel() != pol * npix:
        raise ValueError("reorganize_map: %d map entries and %d observed pixels for npix=%d, pol=%d"
                         % (x.numel(), op.numel(), npix, pol))
    nfull = nside2npix(nside)
    full = D.empty(pol * nfull)
    _hip.call("cm2_cutsky_to_fullsky", int(pol), npix, D.ptr(op), D.ptr(x), nfull, D.ptr(full),
              D.stream())
    maps = [full[k * nfull:(k + 1) * nfull] for k in range(pol)]
    _write_fits(fname, maps)
    if D.is_tensor(mapin):
        return maps
    return [D.to_host(m) for m in maps]


def full2cutskymap(hp_map, pol, npix, observpix):
    """Sequence of full-sky maps [I, Q, U] -> concatenated cut-sky vector
    ``I0,Q0,U0, I1,Q1,U1, ...`` (reference IOfiles.py:377-393).  For ``pol == 1`` the
    reference returns the list ``[map[observpix]]``; so does this function."""
    D.require_gpu()
    maps = [D.f64(m).reshape(-1) for m in hp_map]
    if len(maps) < pol:
        raise ValueError("full2cutskymap: %d maps given for pol=%d" % (len(maps), pol))
    nfull = int(maps[0].numel())
    full = torch.cat(maps[:pol]) if pol > 1 else maps[0]
    op = _i64(observpix).reshape(-1)
    npix = int(npix)
    if op.numel() != npix:
        raise ValueError("full2cutskymap: %d observed pixels for npix=%d" % (op.numel(), npix))
    out = D.empty(pol * npix)
    _hip.call("cm2_fullsky_to_cutsky", int(pol), npix, D.ptr(op), D.ptr(full), nfull, D.ptr(out),
              D.stream())
    host = not any(D.is_tensor(m) for m in hp_map)
    res = D.to_host(out) if host else out
    return [res] if pol == 1 else res
