"""Go / no-go probe for a CHUNK-LOCAL tile order (round 4): the overlap-save kernel
on the global tile order [tile][time] against the same kernel on the order [noise block][tile][time].

No library change is needed for the probe: the chunked order is the tile order of VIRTUAL pixels
pix' = block * npix + pix (npix is a whole number of tiles), i.e. nb * ntiles virtual tiles.  With that
many tiles the run tables do not fit LDS, so both orders are timed on PLAIN lists (6 bytes an entry);
the global order is timed on run-coded lists as well (the shipped default).  P / P^T are not touched.
Every result is checked against the time-order application of the same operator.

    PROBE_NT, PROBE_NB, PROBE_NSIDE, PROBE_TP as os_probe.py;  PROBE_ORDERS="global,chunked"
"""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from bench import toeplitz_band

nside = int(os.environ.get("PROBE_NSIDE", "256"))
nt, nb, lam = int(os.environ.get("PROBE_NT", "100000000")), int(os.environ.get("PROBE_NB", "100")), 2048
tp = int(os.environ.get("PROBE_TP", "1536"))
npix = 12 * nside * nside
assert npix % tp == 0 and nt % nb == 0 and npix * ((nt + 2**17 - 1) // 2**17) < 2**31
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
rng = np.random.default_rng(0)
bands = [toeplitz_band(lam, rng) for _ in range(nb)]
tod = torch.rand(nt, generator=g, device=dev, dtype=torch.float64)
N = BlockLO(nt // nb, bands, offdiag=True, method=3)
want_time = N * tod
torch.cuda.synchronize()

# PROBE_ORDERS: "global" or "chunked[:samples per chunk]" (default: one chunk per noise block; chunks need
# not be aligned with blocks or windows -- a window that crosses a chunk boundary just has more runs)
for order in os.environ.get("PROBE_ORDERS", "global,chunked").split(","):
    if order.startswith("chunked"):
        clen = int(order.split(":")[1]) if ":" in order else nt // nb
        nch = (nt + clen - 1) // clen
        blk = (torch.arange(nt, device=dev, dtype=torch.int64) // clen)
        vp = (blk * npix + pix.to(torch.int64)).to(torch.int32)
        del blk
        P = SparseLO(npix * nch, nt, vp, pol=1)
    else:
        P = SparseLO(npix, nt, pix, pol=1)
    h = ctypes.c_void_p()
    _hip.call("cm2_tiles_create", ctypes.byref(h), D.ptr(P._d_pix), D.ptr(P._d_cos), D.ptr(P._d_sin),
              P.nrows, P.ncols, 1, tp, 12207, D.stream())
    T = L._TileHandle(h)
    a = D.empty(T.nvalid)
    want = D.empty(T.nvalid)
    _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(tod), D.ptr(a), D.stream())
    _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(want_time), D.ptr(want), D.stream())
    for lists in (("plain",) if order.startswith("chunked") else ("plain", "rc")):
        os.environ["CM2_OS_LISTS"] = lists
        Nv = BlockLO(nt // nb, bands, offdiag=True, method=3)
        b = torch.zeros_like(a)

        def run():
            _hip.call("cm2_noise_apply_tiles", Nv._noise.h, T.h, D.ptr(a), D.ptr(b), D.stream())
        run()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for x, y in evs:
            x.record()
            run()
            y.record()
        torch.cuda.synchronize()
        err = float((b - want).norm() / want.norm())
        print(json.dumps({"order": order, "lists": lists, "tiles": T.ntiles, "tile_pixels": T.tile_pixels,
                          "ms": round(float(np.median([x.elapsed_time(y) for x, y in evs])), 4),
                          "rel_l2_vs_time_order": err, **Nv.tile_kernel_info()}), flush=True)
        del Nv, b
    del a, want, T, P
