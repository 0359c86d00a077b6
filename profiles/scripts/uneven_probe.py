"""Stage times of P^T N^-1 P at C4 size for the uniform, the uneven (half of the samples on a tenth of
the map) and the hot-pixel (5 % on one pixel) hit maps; plan builders selected by the environment
(CM2_OS_LIST_BUILD, CM2_FX_BUILD, CM2_TILE_BUILD, CM2_PT_SLICE).  One JSON line per hit map."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from cosmomap2_amd.utilities import ProcessTimeSamples
from bench import toeplitz_band
nside, nt, nb, lam, pol = 256, 100_000_000, 100, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
which = sys.argv[1:] or ["uniform", "uneven", "hot"]

def ev(fn, reps=7):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

for name in which:
    g = torch.Generator(device=dev).manual_seed(20161203)
    pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
    if name == "uneven":
        hot = torch.rand(nt, generator=g, device=dev) < 0.5
        pix[hot] = pix[hot] % (npix // 10)
    elif name == "hot":
        pix[torch.rand(nt, generator=g, device=dev) < 0.05] = npix // 3
    phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi)
    n = ces.get_new_pixel[0]
    P = SparseLO(n, nt, pix, pol=pol, angle_processed=ces)
    T = L._sparse_tiles(P)
    x = torch.rand(pol * n, device=dev, dtype=torch.float64)
    tb, tb2, out = D.empty(T.nvalid), D.empty(T.nvalid), D.empty(pol * n)
    st = D.stream()
    r = {"hit_map": name, "tiles": int(T.ntiles), "env": {k: v for k, v in os.environ.items() if k.startswith("CM2_")}}
    r["P"] = round(ev(lambda: _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(tb), st)), 4)
    r["N^-1"] = round(ev(lambda: _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(tb), D.ptr(tb2), st)), 4)
    r["P^T"] = round(ev(lambda: _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(tb), D.ptr(out), st)), 4)
    S_, bytes_ = T.fixed_order_info()
    r["plan"] = {"fx_slice": S_, "fx_bytes_per_sample": round(bytes_ / max(T.nvalid, 1), 2)}
    r["os"] = N.tile_kernel_info()
    print(json.dumps(r, default=str), flush=True)
    del P, T, ces, pix, phi, x, tb, tb2, out
