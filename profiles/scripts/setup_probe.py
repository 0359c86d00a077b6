"""Wall-clock split of the one-off setup at C4 size (1e8 samples, nside 256, IQU, Toeplitz 2048)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import (SparseLO, BlockLO, BlockDiagonalPreconditionerLO,
                                      linearoperators as L)
from cosmomap2_amd.utilities import ProcessTimeSamples
from bench import toeplitz_band
nside, nt, nb, lam, pol = 256, 100_000_000, 100, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
sync = torch.cuda.synchronize
g = torch.Generator(device=dev).manual_seed(1)
rng = np.random.default_rng(0)
torch.empty(1, device=dev); _hip.load(); sync()
tm = {}
def lap(name, t0):
    sync(); tm[name] = round(time.time() - t0, 4)
for rep in range(2):
    pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
    phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    bands = [toeplitz_band(lam, rng) for _ in range(nb)]
    sync()
    t0 = time.time(); N = BlockLO(nt // nb, bands, offdiag=True, method=3); lap("BlockLO", t0)
    t0 = time.time(); ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi); lap("ProcessTimeSamples", t0)
    del phi
    npc = ces.get_new_pixel[0]
    t0 = time.time(); P = SparseLO(npc, nt, pix, pol=pol, angle_processed=ces); lap("SparseLO", t0)
    t0 = time.time(); M = BlockDiagonalPreconditionerLO(ces, npc, pol=pol); lap("M_BD", t0)
    A = P.T * N * P
    x = torch.rand(pol * npc, device=dev, dtype=torch.float64)
    t0 = time.time(); T = L._sparse_tiles(P); lap("tile_plan", t0)
    d1 = D.empty(T.nvalid); d2 = D.empty(T.nvalid); y = D.empty(pol * npc); sync()
    t0 = time.time(); _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d1), D.stream()); lap("first_P_tiles", t0)
    t0 = time.time(); _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d1), D.ptr(d2), D.stream()); lap("first_N_tiles(fft lists)", t0)
    t0 = time.time(); _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(d2), D.ptr(y), D.stream()); lap("first_Pt_tiles(fixed-order lists)", t0)
    t0 = time.time(); A * x; lap("first_operator_matvec", t0)
    t0 = time.time(); A * x; lap("second_operator_matvec", t0)
    tm["total"] = round(sum(v for k, v in tm.items() if k not in ("total", "second_operator_matvec")), 4)
    print(json.dumps({"rep": rep, **tm}), flush=True)
    del N, ces, P, M, A, T, d1, d2, y, x, pix
    tm = {}

if os.environ.get("PROBE_CPROFILE"):
    import cProfile, pstats
    pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
    phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    sync()
    pr = cProfile.Profile(); pr.enable()
    ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi); sync()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
