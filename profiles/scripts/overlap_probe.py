"""Does a bandwidth-bound kernel with a small footprint (a plain copy: few VGPRs, no LDS) run
beside k_overlap_save_reg on another stream?  Times FFT alone, copy alone, both together."""
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
T = L._sparse_tiles(P, tile_pixels=1536)
rng = np.random.default_rng(0)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
a = torch.rand(T.nvalid + 8, generator=g, device=dev, dtype=torch.float64); b = torch.empty_like(a)
src = torch.rand(125_000_000, device=dev, dtype=torch.float64); dst = torch.empty_like(src)   # 1 GB each
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def fft():
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), s1.cuda_stream)
def cp():
    with torch.cuda.stream(s2):
        dst.copy_(src)
def timed(fns, reps=10):
    for f in fns: f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        for f in fns: f()
        d1, d2 = torch.cuda.Event(), torch.cuda.Event()
        d1.record(s1); d2.record(s2)
        torch.cuda.current_stream().wait_event(d1); torch.cuda.current_stream().wait_event(d2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return round(float(np.median(ts)), 4)
out = {"fft_ms": timed([fft]), "copy_2GB_ms": timed([cp]), "both_fft_first_ms": timed([fft, cp]),
       "both_copy_first_ms": timed([cp, fft])}
print(json.dumps(out))
