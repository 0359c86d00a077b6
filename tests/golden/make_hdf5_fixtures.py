"""
Input fixtures from the only data files the reference holds for this path:
/root/reference/data/testcase_block_diag_{3,4}.hdf5 (written by the reference's
utilities/IOfiles.py:277-300 ``write_to_hdf5``: group ``bolo_pair`` with ``pixel`` (i32[100]),
``pol_angle`` (f64[100]), ``sum`` (f64[100]) and ``weight`` (f64[2,2] / f64[2])).

Run in the build container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_hdf5_fixtures.py

It copies the two data files (data, not source) next to this script, so that the HDF5 reader
of cosmomap2_amd.utilities.hdf5_lite is tested on files h5py wrote, and stores the arrays it
parses from them in reference_inputs.npz.  The parser here is a separate, minimal walk of
exactly these files' structure (superblock v0 -> root symbol table -> one group -> four
contiguous datasets), so that the fixture does not depend on the code under test.
"""
import os
import shutil
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/data"


def parse(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89HDF\r\n\x1a\n" and d[8] == 0 and d[13] == 8 and d[14] == 8
    u16 = lambda p: struct.unpack_from("<H", d, p)[0]
    u32 = lambda p: struct.unpack_from("<I", d, p)[0]
    u64 = lambda p: struct.unpack_from("<Q", d, p)[0]

    def messages(a):
        assert d[a] == 1
        n, size = u16(a + 2), u32(a + 8)
        p, out = a + 16, []
        while len(out) < n and p < a + 16 + size:
            out.append((u16(p), p + 8, u16(p + 2)))
            p += 8 + u16(p + 2)
        return out

    def entries(bt, heap):
        seg = u64(heap + 24)
        assert d[bt:bt + 4] == b"TREE" and d[bt + 5] == 0          # a single leaf level
        out = []
        for i in range(u16(bt + 6)):
            sn = u64(bt + 24 + 8 + 16 * i)
            assert d[sn:sn + 4] == b"SNOD"
            for j in range(u16(sn + 6)):
                e = sn + 8 + 40 * j
                off = u64(e)
                name = d[seg + off:d.index(b"\0", seg + off)].decode()
                out.append((name, u64(e + 8)))
        return out

    def symtab(a):
        for t, p, s in messages(a):
            if t == 0x11:
                return u64(p), u64(p + 8)
        raise AssertionError("no symbol table")

    root = u64(24 + 32 + 8)
    (gname, gaddr), = entries(*symtab(root))
    assert gname == "bolo_pair"
    arrays = {}
    for name, a in entries(*symtab(gaddr)):
        shape = dt = addr = None
        for t, p, s in messages(a):
            if t == 0x01:
                rank = d[p + 1]
                shape = tuple(u64(p + 8 + 8 * i) for i in range(rank))
            elif t == 0x03:
                cls, be, size = d[p] & 15, d[p + 1] & 1, u32(p + 4)
                dt = np.dtype((">" if be else "<") + ("i" if cls == 0 else "f") + str(size))
            elif t == 0x08:
                assert d[p] == 3 and d[p + 1] == 1                  # contiguous
                addr = u64(p + 2)
        n = int(np.prod(shape))
        arrays[name] = np.frombuffer(d, dtype=dt, count=n, offset=addr).reshape(shape).astype(dt.newbyteorder("="))
    return arrays


def main():
    out = {}
    for k in (3, 4):
        name = "testcase_block_diag_%d.hdf5" % k
        shutil.copyfile(os.path.join(SRC, name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
        for key, arr in parse(os.path.join(SRC, name)).items():
            out["case%d_%s" % (k, key)] = arr
    np.savez(os.path.join(HERE, "reference_inputs.npz"), **out)
    for k, v in sorted(out.items()):
        print(k, v.dtype, v.shape)


if __name__ == "__main__":
    main()
