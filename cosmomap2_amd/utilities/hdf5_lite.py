"""
A small pure-Python reader / writer for the subset of HDF5 the reference's files use
(utilities/IOfiles.py of the reference writes them with h5py, which this image lacks).

Supported, after the HDF5 File Format Specification 2.0 / 1.1:
  * superblock version 0 / 1, 8-byte offsets and lengths;
  * "old style" groups: symbol-table message -> v1 B-tree (node type 0) -> SNOD leaves ->
    local heap for the link names;
  * version-1 object headers incl. continuation blocks;
  * datasets with a simple or scalar dataspace, fixed-point / IEEE float / fixed-length string
    datatypes of either byte order, and compact, contiguous or chunked (unfiltered) layout
    (layout message versions 1-3; chunk index = v1 B-tree node type 1).
That is what h5py's default ("earliest") file format produces for the reference's
``create_dataset`` calls (IOfiles.py:214-300), including ``chunks=True``.

``read_file(path)`` returns nested dicts (groups) of NumPy arrays in native byte order.
``write_file(path, tree)`` writes nested dicts of arrays with contiguous layout (or chunked,
for arrays wrapped in ``Chunked``); an array's dtype byte order is kept (the reference stores big-endian ``STD_I32BE`` / ``IEEE_F64BE``).
Written files are checked against this reader and, structurally, against the reference's own
h5py-written test files; h5py itself is not available here to read them back.
"""
import mmap
import struct

import numpy as np

__all__ = ["read_file", "write_file", "Chunked", "Hdf5FormatError"]

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(RuntimeError):
    pass


class Chunked(object):
    """Marks an array of the tree given to :func:`write_file` for chunked storage:
    ``{"Eigenvectors": Chunked(z, (4096, 8))}``."""

    def __init__(self, array, chunks):
        self.array, self.chunks = np.asarray(array), tuple(chunks)


# =============================================================================== reader ===
class _Reader(object):
    def __init__(self, data):
        self.d = data
        if data[:8] != _SIG:
            raise Hdf5FormatError("not an HDF5 file (bad signature)")
        ver = data[8]
        if ver not in (0, 1):
            raise Hdf5FormatError("superblock version %d is not supported (0 and 1 are)" % ver)
        if data[13] != 8 or data[14] != 8:
            raise Hdf5FormatError("only 8-byte offsets / lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self.u64(p)
        root = p + 32                               # base, free space, EOF, driver info
        self.root_header = self.u64(root + 8)

    def u16(self, p):
        return struct.unpack_from("<H", self.d, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.d, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.d, p)[0]

    # ---- object header (version 1) -> list of (type, payload offset, size)
    def messages(self, addr):
        addr += self.base
        if self.d[addr] != 1:
            raise Hdf5FormatError("object header version %d is not supported" % self.d[addr])
        nmsg = self.u16(addr + 2)
        hsize = self.u32(addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        seen = set()
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            if p in seen or p < 0 or p + left > len(self.d):
                raise Hdf5FormatError("object header continuation outside the file or cyclic")
            seen.add(p)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self.u16(p), self.u16(p + 2)
                body = p + 8
                if mtype == 0x10:                   # continuation
                    blocks.append((self.u64(body) + self.base, self.u64(body + 8)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    def heap_name(self, heap_addr, off):
        h = heap_addr + self.base
        if self.d[h:h + 4] != b"HEAP":
            raise Hdf5FormatError("local heap signature missing")
        seg = self.u64(h + 24) + self.base
        end = self.d.find(b"\0", seg + off)
        if end < 0:
            raise Hdf5FormatError("unterminated link name in the local heap")
        return self.d[seg + off:end].decode("ascii")

    def group_entries(self, btree_addr, heap_addr):
        """[(name, object header address)] of an old-style group."""
        out = []
        stack = [btree_addr]
        seen = set()
        while stack:
            a = stack.pop() + self.base
            if a in seen:
                raise Hdf5FormatError("cyclic group B-tree")
            seen.add(a)
            if self.d[a:a + 4] == b"TREE":
                ntype, used = self.d[a + 4], self.u16(a + 6)
                if ntype != 0:
                    raise Hdf5FormatError("group B-tree node of type %d" % ntype)
                p = a + 24
                kids = [self.u64(p + 8 + 16 * i) for i in range(used)]
                stack.extend(reversed(kids))
            elif self.d[a:a + 4] == b"SNOD":
                nsym = self.u16(a + 6)
                for i in range(nsym):
                    e = a + 8 + 40 * i
                    out.append((self.heap_name(heap_addr, self.u64(e)), self.u64(e + 8)))
            else:
                raise Hdf5FormatError("unexpected node in a group B-tree")
        return out

    def dtype(self, p):
        cv = self.d[p]
        cls, bits0 = cv & 0x0F, self.d[p + 1]
        size = self.u32(p + 4)
        if cls == 0:
            order = ">" if bits0 & 1 else "<"
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", size))
        if cls == 1:
            if bits0 & 0x40:
                raise Hdf5FormatError("VAX floating point is not supported")
            return np.dtype("%sf%d" % (">" if bits0 & 1 else "<", size))
        if cls == 3:
            return np.dtype("S%d" % size)
        raise Hdf5FormatError("datatype class %d is not supported" % cls)

    def dataspace(self, p):
        ver, rank, flags = self.d[p], self.d[p + 1], self.d[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
        else:
            raise Hdf5FormatError("dataspace message version %d" % ver)
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def read_chunks(self, btree, rank, chunk, dt, out):
        stack = [btree]
        seen = set()
        while stack:
            if stack[-1] in seen:
                raise Hdf5FormatError("cyclic chunk B-tree")
            seen.add(stack[-1])
            a = stack.pop() + self.base
            if self.d[a:a + 4] != b"TREE" or self.d[a + 4] != 1:
                raise Hdf5FormatError("bad chunk B-tree node")
            level, used = self.d[a + 5], self.u16(a + 6)
            ksz = 8 + 8 * (rank + 1)
            p = a + 24
            for i in range(used):
                key = p + i * (ksz + 8)
                nbytes, fmask = self.u32(key), self.u32(key + 4)
                offs = [self.u64(key + 8 + 8 * j) for j in range(rank)]
                child = self.u64(key + ksz)
                if level > 0:
                    stack.append(child)
                    continue
                if fmask:
                    raise Hdf5FormatError("filtered chunks are not supported")
                blockv = np.frombuffer(self.d, dtype=dt, count=int(np.prod(chunk)),
                                       offset=child + self.base).reshape(chunk)
                sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, out.shape))
                sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
                out[sl_out] = blockv[sl_in]

    def dataset(self, msgs):
        shape = dt = None
        layout = None
        for mtype, p, size in msgs:
            if mtype == 0x01:
                shape = self.dataspace(p)
            elif mtype == 0x03:
                dt = self.dtype(p)
            elif mtype == 0x0B:
                if self.d[p + 1] != 0:
                    raise Hdf5FormatError("filtered (compressed) datasets are not supported")
            elif mtype == 0x08:
                layout = p
        if shape is None or dt is None or layout is None:
            raise Hdf5FormatError("object is neither a group nor a readable dataset")
        count = 1
        for extent in shape:                        # python ints: no overflow on a corrupt extent
            count *= int(extent)
        # a chunked dataset may be larger than the file (unallocated chunks read as zeros), but
        # not absurdly so: refuse what can only be a corrupt dataspace before allocating it
        if count * dt.itemsize > 64 * len(self.d) + (1 << 24):
            raise Hdf5FormatError("dataset of %d bytes in a file of %d" % (count * dt.itemsize, len(self.d)))
        ver = self.d[layout]
        p = layout
        if ver == 3:
            cls = self.d[p + 1]
            if cls == 0:
                n = self.u16(p + 2)
                if n < count * dt.itemsize:
                    raise Hdf5FormatError("compact dataset holds %d bytes, its dataspace needs %d"
                                          % (n, count * dt.itemsize))
                raw = np.frombuffer(self.d, dtype=dt, count=count, offset=p + 4)
            elif cls == 1:
                addr = self.u64(p + 2)
                if addr == _UNDEF:
                    raw = np.zeros(count, dtype=dt)
                else:
                    raw = np.frombuffer(self.d, dtype=dt, count=count, offset=addr + self.base)
            elif cls == 2:
                ndim = self.d[p + 2]
                bt = self.u64(p + 3)
                chunk = tuple(self.u32(p + 11 + 4 * i) for i in range(ndim - 1))
                out = np.zeros(shape, dtype=dt)
                if bt != _UNDEF:
                    self.read_chunks(bt, ndim - 1, chunk, dt, out)
                raw = out
            else:
                raise Hdf5FormatError("layout class %d" % cls)
        elif ver in (1, 2):
            ndim, cls = self.d[p + 1], self.d[p + 2]
            q = p + 8
            addr = None
            if cls != 0:
                addr = self.u64(q)
                q += 8
            dims = [self.u32(q + 4 * i) for i in range(ndim)]
            q += 4 * ndim
            if cls == 1:
                raw = (np.zeros(count, dtype=dt) if addr == _UNDEF else
                       np.frombuffer(self.d, dtype=dt, count=count, offset=addr + self.base))
            elif cls == 0:
                raw = np.frombuffer(self.d, dtype=dt, count=count, offset=q + 4)
            elif cls == 2:
                chunk = tuple(dims[:-1])
                out = np.zeros(shape, dtype=dt)
                if addr != _UNDEF:
                    self.read_chunks(addr, ndim - 1, chunk, dt, out)
                raw = out
            else:
                raise Hdf5FormatError("layout class %d" % cls)
        else:
            raise Hdf5FormatError("layout message version %d" % ver)
        arr = np.asarray(raw).reshape(shape)
        if arr.dtype.kind in "iuf":
            arr = arr.astype(arr.dtype.newbyteorder("="))          # native order, own memory
        else:
            arr = arr.copy()
        return arr

    def obj(self, addr, _depth=0, _open=None):
        _open = set() if _open is None else _open
        if addr in _open or _depth > 64:
            raise Hdf5FormatError("cyclic or too deeply nested groups")
        _open = _open | {addr}
        msgs = self.messages(addr)
        for mtype, p, size in msgs:
            if mtype == 0x11:                       # symbol table -> group
                bt, heap = self.u64(p), self.u64(p + 8)
                return {name: self.obj(a, _depth + 1, _open) for name, a in self.group_entries(bt, heap)}
        return self.dataset(msgs)


def read_file(path):
    """The whole file as nested dicts: groups -> dicts, datasets -> NumPy arrays (scalar
    datasets -> 0-d arrays).  A truncated or corrupt file raises :class:`Hdf5FormatError`."""
    # the file is mapped, not read: every dataset is copied once, from the page cache into its own
    # array (a 600 MB Ritz-vector checkpoint is not held twice); the mapping goes with `data`
    with open(path, "rb") as f:
        try:
            data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        except (ValueError, OSError):               # empty file, or a file system without mmap
            data = f.read()
    if len(data) < 64:
        raise Hdf5FormatError("not an HDF5 file (too short)")
    try:
        return _Reader(data).obj(struct.unpack_from("<Q", data, 24 + (4 if data[8] == 1 else 0) + 40)[0])
    except Hdf5FormatError:
        raise
    except (IndexError, struct.error, ValueError, TypeError, OverflowError, KeyError,
            UnicodeDecodeError, RecursionError, MemoryError, AssertionError) as e:
        raise Hdf5FormatError("corrupt or truncated HDF5 file (%s: %s)" % (type(e).__name__, e))


# =============================================================================== writer ===
_LEAF_K, _INT_K = 32, 16                            # symbols per SNOD = 2 K_leaf; children = 2 K_int


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    be = 1 if (dt.byteorder == ">" or (dt.byteorder == "=" and not np.little_endian)) else 0
    if dt.kind in "iu":
        bits = be | (8 if dt.kind == "i" else 0)
        return struct.pack("<BBBBI", 0x10, bits, 0, 0, dt.itemsize) + \
            struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 8:
            prop = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            sign = 63
        else:
            prop = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            sign = 31
        return struct.pack("<BBBBI", 0x11, be | 0x20, sign, 0, dt.itemsize) + prop
    raise Hdf5FormatError("cannot write dtype %r" % (dt,))


class _Writer(object):
    """Lays the file out as a list of (address, bytes-like) pieces; dataset payloads are kept as
    views of the caller's arrays and go to the file with one write each."""

    def __init__(self):
        self.pieces = []
        self.end = 96                               # superblock, written last

    def alloc(self, data):
        self.end += -self.end % 8
        addr = self.end
        self.pieces.append((addr, data))
        self.end += len(data) if not isinstance(data, memoryview) else data.nbytes
        return addr

    def header(self, msgs):
        body = b"".join(msgs)
        return self.alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def dataset(self, arr, chunks=None):
        arr = np.asarray(arr)
        if arr.dtype.kind not in "iuf":
            raise Hdf5FormatError("only integer and float datasets can be written")
        if not arr.flags.c_contiguous:                   # (ascontiguousarray would make 0-d arrays 1-d)
            arr = np.array(arr, order="C")
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape)
        # fill value: version 2, late allocation, written if set, defined with size 0 -- the
        # message h5py's create_dataset leaves (reference test files, data/testcase_block_diag_*)
        fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)
        if chunks is None or arr.ndim == 0 or arr.size == 0:
            nbytes = arr.nbytes
            daddr = self.alloc(memoryview(arr.reshape(-1).view(np.uint8))) if nbytes else _UNDEF
            layout = struct.pack("<BBQQ", 3, 1, daddr, nbytes)
        else:
            layout = self._chunked(arr, tuple(int(c) for c in chunks))
        return self.header([_msg(0x01, space), _msg(0x03, _dtype_msg(arr.dtype), flags=1),
                            _msg(0x05, fill, flags=1), _msg(0x08, layout, flags=1)])

    def _chunked(self, arr, chunks):
        """Chunked, unfiltered storage (what ``create_dataset(..., chunks=True)`` of the
        reference's Ritz writer, IOfiles.py:232, produces): every chunk stored whole (edge chunks
        padded), indexed by ONE leaf node of a v1 B-tree of type 1."""
        rank = arr.ndim
        if len(chunks) != rank or any(c < 1 for c in chunks):
            raise Hdf5FormatError("chunk shape must have one positive entry per dimension")
        grid = [-(-s // c) for s, c in zip(arr.shape, chunks)]
        nchunks = int(np.prod(grid))
        if nchunks > 64:
            # one B-tree leaf indexes at most 64 chunks: coarser chunks along the first dimension
            # until they fit (any chunk shape is a valid HDF5 file; readers see the same array)
            chunks = list(chunks)
            while int(np.prod([-(-s // c) for s, c in zip(arr.shape, chunks)])) > 64:
                d = max(range(rank), key=lambda i: -(-arr.shape[i] // chunks[i]))
                chunks[d] = min(arr.shape[d], 2 * chunks[d])
            chunks = tuple(chunks)
            grid = [-(-s // c) for s, c in zip(arr.shape, chunks)]
            nchunks = int(np.prod(grid))
        esz = arr.dtype.itemsize
        csize = int(np.prod(chunks)) * esz
        keys = []
        for idx in np.ndindex(*grid):
            off = [i * c for i, c in zip(idx, chunks)]
            block = np.zeros(chunks, dtype=arr.dtype)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunks, arr.shape))
            block[tuple(slice(0, x.stop - x.start) for x in sl)] = arr[sl]
            keys.append((off, self.alloc(block.tobytes())))
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 1, 0, nchunks, _UNDEF, _UNDEF))
        for off, addr in keys:
            node += struct.pack("<II", csize, 0) + b"".join(struct.pack("<Q", o) for o in off)
            node += struct.pack("<Q", 0) + struct.pack("<Q", addr)
        node += struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape)
        node += struct.pack("<Q", 0)
        node += b"\0" * (24 + 65 * (8 + 8 * (rank + 1)) + 64 * 8 - len(node))
        bt = self.alloc(bytes(node))
        return struct.pack("<BBB", 3, 2, rank + 1) + struct.pack("<Q", bt) + \
            b"".join(struct.pack("<I", c) for c in chunks) + struct.pack("<I", esz)

    def group(self, tree):
        """-> (object header address, B-tree address, heap address)"""
        names = sorted(tree)
        if len(names) > 2 * _LEAF_K * 2 * _INT_K:
            raise Hdf5FormatError("too many entries in one group")
        children = {}
        for name in names:
            node = tree[name]
            if isinstance(node, dict):
                children[name] = self.group(node)
            elif isinstance(node, Chunked):
                children[name] = (self.dataset(node.array, chunks=node.chunks),)
            else:
                children[name] = (self.dataset(node),)
        # local heap: empty string at offset 0, then the names, 8-byte aligned
        heap = bytearray(8)
        offs = {}
        for name in names:
            offs[name] = len(heap)
            heap += _pad8(name.encode("ascii") + b"\0")
        seg_addr = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, seg_addr))
        # symbol nodes
        snods, keys = [], [0]
        per = 2 * _LEAF_K
        for i in range(0, max(len(names), 1), per):
            part = names[i:i + per]
            body = bytearray()
            for name in part:
                c = children[name]
                if len(c) == 3:
                    body += struct.pack("<QQII", offs[name], c[0], 1, 0) + struct.pack("<QQ", c[1], c[2])
                else:
                    body += struct.pack("<QQII16x", offs[name], c[0], 0, 0)
            body += b"\0" * (40 * (per - len(part)))
            snods.append(self.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + bytes(body)))
            keys.append(offs[part[-1]] if part else 0)
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), _UNDEF, _UNDEF))
        for i, s in enumerate(snods):
            node += struct.pack("<QQ", keys[i], s)
        node += struct.pack("<Q", keys[len(snods)])
        node += b"\0" * (24 + (2 * 2 * _INT_K + 1) * 8 - len(node))
        bt_addr = self.alloc(bytes(node))
        hdr = self.header([_msg(0x11, struct.pack("<QQ", bt_addr, heap_addr))])
        return hdr, bt_addr, heap_addr

    def finish(self, root, path):
        hdr, bt, heap = root
        sb = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INT_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, self.end, _UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, heap)
        assert len(sb) == 96
        with open(path, "wb") as f:
            f.write(sb)
            for addr, data in self.pieces:
                f.seek(addr)
                f.write(data)
            f.truncate(self.end)


def write_file(path, tree):
    """Write nested dicts (groups) of arrays (datasets) as an HDF5 file."""
    w = _Writer()
    w.finish(w.group(tree), path)
