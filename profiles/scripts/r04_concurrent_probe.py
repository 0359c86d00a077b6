"""Would P / P^T of one time chunk hide beside the overlap-save kernel of another?  No library change: the
three kernels of a C4-size step on independent buffers, (a) one after the other on one stream, (b) N^-1 on
one stream and P + P^T on a second one, K repetitions each, wall clock between synchronisations.  (b) is the
ceiling of any chunk-pipelined matvec (which would add fill / drain and a combine of the P^T pieces)."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from bench import toeplitz_band

nside, nt, nb, lam, pol = 256, 100000000, 100, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
ang = type("A", (), {})()
ang.cos, ang.sin = torch.cos(2 * phi), torch.sin(2 * phi)
del phi
rng = np.random.default_rng(0)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
P = SparseLO(npix, nt, pix, pol=pol, angle_processed=ang)
T = L._sparse_tiles(P)
_hip.call("cm2_noise_prepare_tiles", N._noise.h, T.h, D.stream())
x = torch.rand(pol * npix, generator=g, device=dev, dtype=torch.float64)
a, b, c, d = (D.empty(T.nvalid) for _ in range(4))
y = D.empty(pol * npix)
for t in (a, d):
    t.uniform_()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def P_(s):
    _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(c), s.cuda_stream)


def N_(s):
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), s.cuda_stream)


def Pt_(s):
    _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(d), D.ptr(y), s.cuda_stream)


def wall(fn, K=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


def sequential():
    P_(s1); N_(s1); Pt_(s1)


def two_streams():
    N_(s1); P_(s2); Pt_(s2)


def three_streams(s3=torch.cuda.Stream()):
    N_(s1); P_(s2); Pt_(s3)


for name, fn in (("P alone", lambda: P_(s1)), ("N^-1 alone", lambda: N_(s1)), ("P^T alone", lambda: Pt_(s1)),
                 ("sequential, one stream", sequential), ("N^-1 | P + P^T", two_streams),
                 ("N^-1 | P | P^T", three_streams), ("sequential again", sequential)):
    print(json.dumps({"case": name, "ms": round(wall(fn), 4)}), flush=True)
