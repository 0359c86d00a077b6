"""Time cm2_noise_apply_tiles (k_overlap_save_reg on the tile order) alone on a C4-sized plan."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from bench import toeplitz_band
nside, nt, nb, lam = 256, 100_000_000, 100, 2048
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
P = SparseLO(npix, nt, pix, pol=1)
T = L._sparse_tiles(P, tile_pixels=int(os.environ.get("PROBE_TP", "1536")))
rng = np.random.default_rng(0)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
a = torch.rand(T.nvalid + 8, generator=g, device=dev, dtype=torch.float64); b = torch.empty_like(a)
def run():
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), D.stream())
run(); torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for x, y in evs:
    x.record(); run(); y.record()
torch.cuda.synchronize()
print(json.dumps({ "tile_pixels": T.tile_pixels,
                  "ms": round(float(np.median([x.elapsed_time(y) for x, y in evs])), 4)}))
