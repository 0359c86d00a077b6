"""Time cm2_noise_apply_tiles (the overlap-save kernel on the tile order) alone on a C4-sized plan,
for every kernel / list-format variant (CM2_OS_KERNEL, CM2_OS_LISTS), and check each against the
segment-pair kernel's output.  PROBE_VARIANTS="pair,real16:rc,..." selects."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from bench import toeplitz_band
nside = int(os.environ.get("PROBE_NSIDE", "256"))
nt, nb, lam = int(os.environ.get("PROBE_NT", "100000000")), int(os.environ.get("PROBE_NB", "100")), 2048
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
P = SparseLO(npix, nt, pix, pol=1)
tp = os.environ.get("PROBE_TP")
T = L._sparse_tiles(P, tile_pixels=int(tp) if tp else (1536 if nside == 256 else None))
rng = np.random.default_rng(0)
bands = [toeplitz_band(lam, rng) for _ in range(nb)]
a = torch.rand(T.nvalid + 8, generator=g, device=dev, dtype=torch.float64)
ref = None
for var in os.environ.get("PROBE_VARIANTS", "pair,real16:rc,real16:plain,real32:rc,real32:plain").split(","):
    kern, _, lists = var.partition(":")
    os.environ["CM2_OS_KERNEL"] = kern
    os.environ["CM2_OS_LISTS"] = lists or "rc"
    N = BlockLO(nt // nb, bands, offdiag=True, method=3)
    b = torch.zeros_like(a)
    def run():
        _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), D.stream())
    run(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for x, y in evs:
        x.record(); run(); y.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = b.clone()
    err = float((b - ref).norm() / ref.norm())
    info = N.tile_kernel_info() if hasattr(N, "tile_kernel_info") else {}
    print(json.dumps({"variant": var, "tile_pixels": T.tile_pixels, "tiles": T.ntiles,
                      "ms": round(float(np.median([x.elapsed_time(y) for x, y in evs])), 4),
                      "rel_l2_vs_first": err, **info}), flush=True)
    del N, b
