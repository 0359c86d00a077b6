"""
HEALPix maps in FITS files without healpy / astropy: what ``hp.read_map`` and ``hp.write_map``
do for the reference (input maps such as data/cmb_r0.2_3.5arcmin_128.fits read at
src/test_BD_precond_onto_real_data.py:31-37; output maps written by ``obspix2mask`` /
``reorganize_map``, utilities/healpy_functions.py:46, :103).

A HEALPix map file is a FITS primary header without data followed by ONE binary-table
extension: TFIELDS columns (one per Stokes map), each row holding ``repeat`` consecutive
pixels of every column (``TFORMn = '1024E'``), values big-endian; keywords PIXTYPE = 'HEALPIX',
ORDERING = 'RING' | 'NESTED', NSIDE, FIRSTPIX, LASTPIX, INDXSCHM = 'IMPLICIT'.
``read_map`` reads such a file (written by healpy, by the HEALPix Fortran / IDL tools, or by
``write_map`` below); ``write_map`` writes what ``healpy.write_map(fname, m)`` writes by default:
float32 columns of 1024 pixels per row named TEMPERATURE, Q_POLARISATION, U_POLARISATION.
"""
import numpy as np

__all__ = ["read_map", "write_map", "FitsFormatError", "nest2ring", "ring2nest", "reorder"]

_BLOCK = 2880
_TYPES = {"E": ">f4", "D": ">f8", "J": ">i4", "K": ">i8", "I": ">i2", "B": "u1", "L": "u1"}


class FitsFormatError(RuntimeError):
    pass


# ------------------------------------------------------------ RING <-> NESTED ordering ------
# Integer arithmetic of the HEALPix pixelisation (Gorski et al. 2005, ApJ 622, 759; what
# hp.reorder / hp.read_map(nest=...) do for the reference): a NESTED index is
# face * nside^2 + (bits of ix and iy interleaved); a RING index counts along iso-latitude rings.
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)   # ring of a face's north corner / nside
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)   # its longitude index, in units of pi/4


def _check_nside(nside):
    nside = int(nside)
    if nside < 1 or nside & (nside - 1) or nside > (1 << 29):
        raise ValueError("nside must be a power of two (NESTED ordering needs it), got %r" % nside)
    return nside


def _compact_bits(v):
    """Every second bit of v (bits 0, 2, 4, ...) packed together."""
    v = v & 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def nest2ring(nside, ipix):
    """RING index of NESTED pixel(s) ``ipix`` (``hp.nest2ring``)."""
    nside = _check_nside(nside)
    ip = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    if ip.size and (ip.min() < 0 or ip.max() >= npix):
        raise ValueError("pixel index outside [0, %d)" % npix)
    nl4, ncap = 4 * nside, 2 * nside * (nside - 1)
    face = ip // (nside * nside)
    ipf = ip % (nside * nside)
    ix, iy = _compact_bits(ipf), _compact_bits(ipf >> 1)
    jr = _JRLL[face] * nside - ix - iy - 1                   # ring, 1 .. 4 nside - 1
    north, south = jr < nside, jr > 3 * nside
    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    n_before = np.where(north, 2 * nr * (nr - 1),
                        np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4))
    kshift = np.where(north | south, 0, (jr - nside) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, np.where(jp < 1, jp + nl4, jp))
    out = n_before + jp - 1
    return out if out.ndim else int(out)


def ring2nest(nside, ipix):
    """NESTED index of RING pixel(s) ``ipix`` (``hp.ring2nest``), through the inverse of the
    whole permutation (a table of 12 nside^2 entries)."""
    nside = _check_nside(nside)
    npix = 12 * nside * nside
    inv = np.empty(npix, dtype=np.int64)
    inv[nest2ring(nside, np.arange(npix, dtype=np.int64))] = np.arange(npix, dtype=np.int64)
    ip = np.asarray(ipix, dtype=np.int64)
    if ip.size and (ip.min() < 0 or ip.max() >= npix):
        raise ValueError("pixel index outside [0, %d)" % npix)
    out = inv[ip]
    return out if out.ndim else int(out)


def reorder(m, n2r=False, r2n=False):
    """``hp.reorder``: a full-sky map in the other ordering (exactly one of n2r / r2n)."""
    if bool(n2r) == bool(r2n):
        raise ValueError("give exactly one of n2r=True (NESTED -> RING) or r2n=True")
    m = np.asarray(m)
    nside = int(round((m.shape[-1] / 12.0) ** 0.5))
    if 12 * nside * nside != m.shape[-1]:
        raise ValueError("map length %d is not 12 * nside**2" % m.shape[-1])
    perm = nest2ring(nside, np.arange(m.shape[-1], dtype=np.int64))   # perm[nest] = ring
    out = np.empty_like(m)
    if n2r:
        out[..., perm] = m
    else:
        out[...] = m[..., perm]
    return out


def _parse_header(data, pos):
    """Cards of the header starting at ``pos`` -> (dict of keyword -> value, offset after the
    header's last 2880-byte block)."""
    cards = {}
    while True:
        if pos + _BLOCK > len(data):
            raise FitsFormatError("truncated FITS header")
        block = data[pos:pos + _BLOCK]
        pos += _BLOCK
        done = False
        for i in range(0, _BLOCK, 80):
            card = block[i:i + 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] != "= " or key in ("COMMENT", "HISTORY", ""):
                continue
            val = card[10:]
            if val.lstrip().startswith("'"):
                v = val.lstrip()[1:]
                end = v.find("'")
                while end != -1 and v[end:end + 2] == "''":          # doubled quote inside a string
                    end = v.find("'", end + 2)
                cards[key] = v[:end].rstrip() if end != -1 else v.rstrip()
            else:
                v = val.split("/")[0].strip()
                if v in ("T", "F"):
                    cards[key] = v == "T"
                else:
                    try:
                        cards[key] = int(v)
                    except ValueError:
                        try:
                            cards[key] = float(v.replace("D", "E"))
                        except ValueError:
                            cards[key] = v
        if done:
            return cards, pos


def read_map(filename, field=0, dtype=np.float64, nest=False, h=False):
    """
    ``hp.read_map(filename, field=...)``: the map(s) of the first binary-table extension.
    ``field`` is a column index or a sequence of indices (``field=[0, 1, 2]`` for I, Q, U;
    ``None`` for all).  ``nest=False`` (healpy's default) returns RING ordering, ``nest=True``
    NESTED, whatever the file's ORDERING keyword says (the map is reordered when they differ);
    ``nest=None`` takes the file's ordering as it is.
    With ``h=True`` the header keywords are returned as the last element.
    """
    with open(filename, "rb") as f:
        data = f.read()
    if data[:6] != b"SIMPLE":
        raise FitsFormatError("not a FITS file")
    try:
        return _read_map(data, field, dtype, nest, h)
    except FitsFormatError:
        raise
    except (KeyError, ValueError, IndexError, TypeError, OverflowError, MemoryError) as e:
        raise FitsFormatError("corrupt or truncated FITS file (%s: %s)" % (type(e).__name__, e))


def _read_map(data, field, dtype, nest, h):
    prim, pos = _parse_header(data, 0)
    naxis = int(prim.get("NAXIS", 0))
    nbytes = abs(int(prim.get("BITPIX", 8))) // 8
    for i in range(naxis):
        nbytes *= int(prim["NAXIS%d" % (i + 1)])
    if naxis:
        pos += -(-nbytes // _BLOCK) * _BLOCK
    hdr, pos = _parse_header(data, pos)
    if hdr.get("XTENSION") != "BINTABLE":
        raise FitsFormatError("first extension is %r, expected a BINTABLE" % hdr.get("XTENSION"))
    width, nrows, nf = int(hdr["NAXIS1"]), int(hdr["NAXIS2"]), int(hdr["TFIELDS"])
    cols, off = [], 0
    for i in range(1, nf + 1):
        form = str(hdr["TFORM%d" % i]).strip()
        rep = int(form[:-1]) if form[:-1] else 1
        code = form[-1]
        if code not in _TYPES:
            raise FitsFormatError("column format %r is not supported" % form)
        dt = np.dtype(_TYPES[code])
        cols.append((off, rep, dt))
        off += rep * dt.itemsize
    if off != width:
        raise FitsFormatError("column formats add up to %d bytes per row, NAXIS1 says %d" % (off, width))
    ordering = str(hdr.get("ORDERING", "RING")).strip().upper()
    convert = nest is not None and (ordering == "NESTED") != bool(nest)
    table = np.frombuffer(data, dtype=np.uint8, count=width * nrows, offset=pos).reshape(nrows, width)
    fields = range(nf) if field is None else ([field] if np.isscalar(field) else list(field))
    maps = []
    for fi in fields:
        o, rep, dt = cols[fi]
        col = np.ascontiguousarray(table[:, o:o + rep * dt.itemsize]).view(dt).reshape(-1)
        maps.append(col.astype(dtype))
    nside = hdr.get("NSIDE")
    if nside is not None and any(m.size != 12 * int(nside) ** 2 for m in maps):
        raise FitsFormatError("column length does not match NSIDE = %r" % nside)
    if convert:
        maps = [reorder(m, n2r=not nest, r2n=bool(nest)) for m in maps]
    out = maps[0] if (field is not None and np.isscalar(field)) else tuple(maps)
    if h:
        return (out, hdr) if (field is not None and np.isscalar(field)) else tuple(maps) + (hdr,)
    return out


def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, (int, np.integer)):
        v = "%20d" % value
    elif isinstance(value, float):
        v = "%20s" % ("%.12E" % value)
    else:
        v = "'%-8s'" % str(value)
        v = "%-20s" % v
    card = "%-8s= %s" % (key, v)
    if comment:
        card += " / " + comment
    return ("%-80s" % card)[:80].encode("ascii")


def _header_bytes(cards):
    raw = b"".join(cards) + ("%-80s" % "END").encode("ascii")
    return raw + b" " * (-len(raw) % _BLOCK)


def write_map(filename, m, nest=False, dtype=np.float32, coord=None, column_names=None):
    """
    ``hp.write_map(filename, m)``: one map or a sequence of maps (I or I, Q, U) of the same
    HEALPix size into a binary table of ``1024``-pixel rows, one column per map.  ``nest`` states
    the ordering of ``m`` (written to the ORDERING keyword; the pixels are stored as given, as
    healpy does).
    """
    maps = [np.asarray(m)] if np.ndim(m) == 1 else [np.asarray(x) for x in m]
    npix = maps[0].size
    nside = int(round((npix / 12.0) ** 0.5))
    if 12 * nside * nside != npix or any(x.size != npix for x in maps):
        raise ValueError("maps must all have 12 * nside**2 pixels")
    dt = np.dtype(dtype).newbyteorder(">")
    code = {"f4": "E", "f8": "D", "i4": "J", "i8": "K", "i2": "I"}.get(dt.str[1:])
    if code is None:
        raise ValueError("dtype %r cannot be written" % (dtype,))
    rep = 1024 if npix % 1024 == 0 else 1
    nrows = npix // rep
    if column_names is None:
        column_names = (["TEMPERATURE", "Q_POLARISATION", "U_POLARISATION"] if len(maps) == 3
                        else ["TEMPERATURE"] if len(maps) == 1
                        else ["COLUMN_%d" % i for i in range(len(maps))])
    prim = [_card("SIMPLE", True, "conforms to FITS standard"),
            _card("BITPIX", 8, "array data type"),
            _card("NAXIS", 0, "number of array dimensions"),
            _card("EXTEND", True)]
    width = rep * dt.itemsize * len(maps)
    ext = [_card("XTENSION", "BINTABLE", "binary table extension"),
           _card("BITPIX", 8, "array data type"),
           _card("NAXIS", 2, "number of array dimensions"),
           _card("NAXIS1", width, "length of dimension 1"),
           _card("NAXIS2", nrows, "length of dimension 2"),
           _card("PCOUNT", 0, "number of group parameters"),
           _card("GCOUNT", 1, "number of groups"),
           _card("TFIELDS", len(maps), "number of table fields")]
    for i, name in enumerate(column_names):
        ext.append(_card("TTYPE%d" % (i + 1), name))
        ext.append(_card("TFORM%d" % (i + 1), "%d%s" % (rep, code)))
    ext += [_card("PIXTYPE", "HEALPIX", "HEALPIX pixelisation"),
            _card("ORDERING", "NESTED" if nest else "RING",
                  "Pixel ordering scheme, either RING or NESTED"),
            _card("EXTNAME", "xtension", "name of this binary table extension"),
            _card("NSIDE", nside, "Resolution parameter of HEALPIX"),
            _card("FIRSTPIX", 0, "First pixel # (0 based)"),
            _card("LASTPIX", npix - 1, "Last pixel # (0 based)"),
            _card("INDXSCHM", "IMPLICIT", "Indexing: IMPLICIT or EXPLICIT"),
            _card("OBJECT", "FULLSKY", "Sky coverage, either FULLSKY or PARTIAL")]
    if coord:
        ext.append(_card("COORDSYS", coord, "Ecliptic, Galactic or Celestial (equatorial)"))
    table = np.empty((nrows, len(maps), rep), dtype=dt)
    for i, x in enumerate(maps):
        table[:, i, :] = x.reshape(nrows, rep)
    payload = table.tobytes()
    with open(filename, "wb") as f:
        f.write(_header_bytes(prim))
        f.write(_header_bytes(ext))
        f.write(payload)
        f.write(b"\0" * (-len(payload) % _BLOCK))
