"""One P^T application on a small tile plan (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, linearoperators as L
nt, npix, tp, pol = 300000, 5000, 2048, 1
rng = np.random.default_rng(1)
pix = rng.integers(0, npix, nt).astype(np.int32)
pix[rng.random(nt) < 0.1] = -1
P = SparseLO(npix, nt, pix, pol=pol)
T = L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)
print("tiles", T.ntiles, T.nvalid, flush=True)
v = D.f64(rng.standard_normal(T.nvalid)); out = D.empty(pol * npix)
_hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v), D.ptr(out), D.stream())
print("launched", T.fixed_order_info(), flush=True)
torch.cuda.synchronize()
print("done", float(out.sum()), flush=True)
