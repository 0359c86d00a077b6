"""
Fixture from the HEALPix map the reference holds for its runs:
/root/reference/data/cmb_r0.2_3.5arcmin_128.fits (synfast output, 3 x 196608 float32 pixels in
a binary table of 1024-pixel rows; read by hp.read_map at src/test_BD_precond_onto_real_data.py:
31-37).  The file is 2.3 MB, so instead of a copy this script stores what pins a reader:
the two headers' keyword cards as written, and the pixel values of every column at 4096 seeded
pixel numbers plus per-column sums, all taken with a few lines of NumPy that know this one
file's layout (two header units of 2880 and 5760 bytes, then 192 rows of 3 x 1024 big-endian
float32).  Build container only:

    python tests/golden/make_fits_fixture.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/data/cmb_r0.2_3.5arcmin_128.fits"


def main():
    d = open(SRC, "rb").read()
    # header units end with a card that starts with 'END' in an 80-byte slot
    ends = [i for i in range(0, 3 * 2880 + 2880, 80) if d[i:i + 8] == b"END     "]
    h1_end = (ends[0] // 2880 + 1) * 2880
    h2_end = (ends[1] // 2880 + 1) * 2880
    table = np.frombuffer(d, dtype=">f4", count=192 * 3 * 1024, offset=h2_end).reshape(192, 3, 1024)
    cols = [table[:, i, :].reshape(-1).astype(np.float64) for i in range(3)]
    rng = np.random.default_rng(20161202)
    idx = np.sort(rng.choice(196608, 4096, replace=False))
    cards = [d[i:i + 80].decode("ascii") for i in range(0, h2_end, 80)]
    keyed = [c for c in cards if c[8:10] == "= "]
    np.savez(os.path.join(HERE, "reference_cmb_map_samples.npz"),
             pixels=idx, I=cols[0][idx], Q=cols[1][idx], U=cols[2][idx],
             sums=np.array([c.sum() for c in cols]), sumsq=np.array([(c * c).sum() for c in cols]),
             header_cards=np.frombuffer("".join(keyed).encode("ascii"), dtype=np.uint8).reshape(-1, 80), header_bytes=np.int64(h2_end),
             primary_bytes=np.int64(h1_end))
    print("header units:", h1_end, h2_end - h1_end, "cards:", len(keyed))


if __name__ == "__main__":
    main()
