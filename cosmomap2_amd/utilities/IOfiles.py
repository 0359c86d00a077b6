"""
Files either side of the solve (reference: utilities/IOfiles.py).

Built: the CES readers (IOfiles.py:18-210: one HDF5 file per constant-elevation scan with
``obspix``, ``n_bolo_pair``, ``n_sample_ces``, ``subscans/{n_sample,t_start}`` and one group
``bolo_pair_<i>`` per detector pair holding ``pixel, pol_angle, ground, sum, dif, weight_sum,
weight_dif``), the sample-flagging helpers, the Ritz-vector checkpoint (:214-256), so that the
deflation space of a scanning strategy is computed once and reloaded by later runs, the
``bolo_pair`` test files of ``write_to_hdf5`` / ``read_from_hdf5`` (:277-300, :317-330), the
observed-pixel list (:258-275) and the map lists (:351-375).

Container: HDF5 through :mod:`cosmomap2_amd.utilities.hdf5_lite`, a small pure-Python
reader / writer of the HDF5 subset the reference's files use (h5py is not part of this
image); the group and dataset names are the reference's.  The reader is checked on the
reference's own h5py-written files; the writer emits the same object-header messages, but its
files have only been read back by this reader (neither h5py nor libhdf5 exists in this image),
so "h5py opens them" is intended, not verified.
"""
import os

import numpy as np

from .. import device as D
from . import hdf5_lite as h5

__all__ = ["write_ritz_eigenvectors_to_hdf5", "read_ritz_eigenvectors_from_hdf5",
           "write_ritz_eigenvectors", "read_ritz_eigenvectors", "write_to_hdf5",
           "read_from_hdf5", "write_obspix_to_hdf5", "read_obspix_from_hdf5",
           "read_from_data", "read_from_data_with_subscan_resize", "read_multiple_ces",
           "flagging_not_in_allCES", "flagging_subscan", "save_maplist", "read_maplist"]


def _host(a):
    return D.to_host(a) if D.is_tensor(a) else np.asarray(a)


def write_ritz_eigenvectors_to_hdf5(z, filename, eigvals=None):
    """Save the Ritz vectors ``z`` (n x r, NumPy or HBM tensor) and optionally their Ritz
    values, with the reference's layout (IOfiles.py:214-238): group ``Ritz_eigenvectors`` with
    the scalar ``n_eigenvectors`` (big-endian int32) and the dataset ``Eigenvectors``; root
    dataset ``Ritz_eigenvalues``."""
    zh = np.ascontiguousarray(_host(z), dtype=np.float64)
    if zh.ndim != 2:
        raise ValueError("z must be 2-d (size_eigenvectors x n_eigenvals)")
    tree = {"Ritz_eigenvectors": {"n_eigenvectors": np.array(zh.shape[1], dtype=">i4"),
                                  "Eigenvectors": zh}}
    if eigvals is not None:
        tree["Ritz_eigenvalues"] = np.ascontiguousarray(_host(eigvals), dtype=np.float64)
    h5.write_file(filename, tree)


def read_ritz_eigenvectors_from_hdf5(filename, eigvals=False):
    """``(z, n_eigenvals)`` or ``(z, n_eigenvals, eigenvals)`` (IOfiles.py:240-256)."""
    f = h5.read_file(filename)
    z = np.ascontiguousarray(f["Ritz_eigenvectors"]["Eigenvectors"], dtype=np.float64)
    n_eigenvals = f["Ritz_eigenvectors"]["n_eigenvectors"]
    if eigvals:
        return z, n_eigenvals, np.asarray(f["Ritz_eigenvalues"], dtype=np.float64)
    return z, n_eigenvals


def write_ritz_eigenvectors(z, filename, eigvals=None):
    """Checkpoint of a deflation basis; ``filename`` gets ``.hdf5`` appended if it has no
    extension."""
    if not os.path.splitext(filename)[1]:
        filename = filename + ".hdf5"
    write_ritz_eigenvectors_to_hdf5(z, filename, eigvals=eigvals)
    return filename


def read_ritz_eigenvectors(filename, eigvals=False, device=False):
    """Inverse of :func:`write_ritz_eigenvectors`: ``Z`` or ``(Z, eigenvalues)``; with
    ``device=True`` Z comes back as a row-major float64 tensor in HBM (bit-identical to what
    was written)."""
    if not os.path.exists(filename) and os.path.exists(filename + ".hdf5"):
        filename = filename + ".hdf5"
    out = read_ritz_eigenvectors_from_hdf5(filename, eigvals=eigvals)
    z = D.f64(out[0]) if device else out[0]
    if eigvals:
        return z, out[2]
    return z


def write_to_hdf5(filename, obs_pixels, noise_values, d, phi=None):
    """The ``bolo_pair`` test file of the reference (IOfiles.py:277-300): big-endian int32
    ``pixel``, big-endian float64 ``weight``, ``sum`` and (optional) ``pol_angle``."""
    grp = {"pixel": np.asarray(obs_pixels).astype(">i4"),
           "weight": np.asarray(noise_values).astype(">f8"),
           "sum": np.asarray(d).astype(">f8")}
    if phi is not None:
        grp["pol_angle"] = np.asarray(phi).astype(">f8")
    h5.write_file(filename, {"bolo_pair": grp})


def read_from_hdf5(filename):
    """``(det, obs_pix, polang, weight)`` of a ``bolo_pair`` file (IOfiles.py:317-330)."""
    g = h5.read_file(filename)["bolo_pair"]
    return (np.asarray(g["sum"], dtype=np.float64), np.asarray(g["pixel"]),
            np.asarray(g["pol_angle"], dtype=np.float64), np.asarray(g["weight"], dtype=np.float64))


def write_obspix_to_hdf5(filename, obspix):
    """Observed-pixel list (IOfiles.py:268-275)."""
    h5.write_file(filename, {"obspix": np.asarray(obspix).astype(">i4")})


def read_obspix_from_hdf5(filename):
    """(IOfiles.py:258-267)."""
    return np.asarray(h5.read_file(filename)["obspix"]).astype(np.int32)


# ------------------------------------------------------------------ CES files -------
def _read_ces(filename, pol, npairs, resize):
    f = h5.read_file(filename)
    hp_pixs = f["obspix"]
    n_bolo_pair = int(f["n_bolo_pair"])
    n_ces = int(f["n_sample_ces"])
    subscan = None
    if resize:
        subscan = [np.asarray(f["subscans"]["n_sample"]), np.asarray(f["subscans"]["t_start"])]
    n_to_read = n_bolo_pair if npairs is None else int(npairs)
    pixs_pair, polang_pair, d_pair, weight_pair, ground_pair = [], [], [], [], []
    for i in range(n_to_read):
        group = f["bolo_pair_" + str(i)]
        pix = np.array(group["pixel"])
        if resize:
            flagging_subscan(pix, subscan)
        pixs_pair.append(pix)
        polang_pair.append(np.asarray(group["pol_angle"]))
        ground_pair.append(np.asarray(group["ground"]).astype("int"))
        if pol == 1:
            d_pair.append(np.asarray(group["sum"]))
            weight_pair.append(group["weight_sum"])
        elif pol == 3 or pol == 2:
            d_pair.append(np.asarray(group["dif"]))
            weight_pair.append(group["weight_dif"])
    out = (np.concatenate(d_pair), np.array(weight_pair), np.concatenate(polang_pair),
           np.concatenate(pixs_pair), hp_pixs, np.concatenate(ground_pair), n_ces)
    if resize:
        return out + (n_to_read, subscan)
    return out


def read_from_data(filename, pol, npairs=None):
    """One constant-elevation scan (IOfiles.py:18-72): ``pol=1`` reads the ``sum`` streams and
    their weights, ``pol=2,3`` the ``dif`` streams.  Returns ``d, weight, polang, pixs, hp_pixs,
    ground, n_ces`` with the detector pairs concatenated."""
    return _read_ces(filename, pol, npairs, False)


def read_from_data_with_subscan_resize(filename, pol, npairs=None):
    """The same with every sample outside a sub-scan flagged (pixel = -1) (IOfiles.py:158-210).
    Returns ``d, weight, polang, pixs, hp_pixs, ground, n_ces, n_to_read, subscan``."""
    return _read_ces(filename, pol, npairs, True)


def read_multiple_ces(filelist, pol, npairs=None, filtersubscan=True):
    """Several scans concatenated (IOfiles.py:73-126).  With ``filtersubscan`` the result ends
    with ``subscan, tstart, samples_per_bolopair, bolopairs_per_ces`` -- the arguments of
    :class:`FilterLO`."""
    readf = read_from_data_with_subscan_resize if filtersubscan else read_from_data
    subscan, tstart = [], []
    bolopairs_per_ces, samples_per_bolopair = [], []
    pixs, polang, d, weight, ground, hp_pixs = [], [], [], [], [], []
    outdata = None
    for fname in filelist:
        outdata = readf(fname, pol, npairs=npairs)
        d.append(outdata[0])
        weight.append(outdata[1])
        polang.append(outdata[2])
        pixs.append(outdata[3])
        ground.append(outdata[5])
        if filtersubscan:
            samples_per_bolopair.append(outdata[6])
            bolopairs_per_ces.append(outdata[7])
            subscan.append(outdata[8][0])
            tstart.append(outdata[8][1])
    hp_pixs.append(outdata[4])                      # of the last file, as the reference (:115)
    head = (np.concatenate(d), np.concatenate(weight), np.concatenate(polang),
            np.concatenate(pixs), np.concatenate(hp_pixs), np.concatenate(ground))
    if filtersubscan:
        return head + (subscan, tstart, samples_per_bolopair, bolopairs_per_ces)
    return head + (samples_per_bolopair, bolopairs_per_ces)


def flagging_not_in_allCES(CES_pixs):
    """Flag (in place) the samples whose pixel is not seen by every scan (IOfiles.py:128-141)."""
    inters = CES_pixs[0]
    for pix in CES_pixs:
        inters = np.intersect1d(inters, pix)
    for pixs in CES_pixs:
        pixs[np.isin(pixs, inters, invert=True)] = -1


def flagging_subscan(unflagged_pix, subscan):
    """Flag (in place) every sample before / between the sub-scans (IOfiles.py:145-156; like the
    reference, samples after the last sub-scan are left alone)."""
    k = 0
    for t, n in zip(subscan[1], subscan[0]):
        unflagged_pix[k:t] = -1
        k = t + n


def save_maplist(maplist, filename):
    """A list of maps, e.g. the iterates of a solve (IOfiles.py:351-363)."""
    tree = {"Nmaps": np.array(len(maplist), dtype=">i4")}
    for i, m in enumerate(maplist):
        tree["Map" + str(i)] = np.asarray(_host(m)).astype(">f8")
    h5.write_file(filename, tree)


def read_maplist(filename):
    """(IOfiles.py:365-375)"""
    f = h5.read_file(filename)
    nmaps = int(f["Nmaps"])
    return [np.array(f["Map" + str(i)]).T for i in range(nmaps)], nmaps
