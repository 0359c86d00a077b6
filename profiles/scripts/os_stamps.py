"""Phase timing inside k_os_real (diagnostic build: CM2_EXTRA_HIPCC_FLAGS=-DCM2_OS_STAMPS python -m
cosmomap2_amd.build --force): mean s_memtime deltas per workgroup between the phase boundaries."""
import os, sys, json, ctypes
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
bands = [toeplitz_band(lam, rng) for _ in range(nb)]
a = torch.rand(T.nvalid + 8, generator=g, device=dev, dtype=torch.float64)
lib = ctypes.CDLL(_hip.LIB_PATH)
lib.cm2_os_debug_stamps.argtypes = [ctypes.c_void_p]
names = ["load window (lists, gathers, staging)", "forward passes", "pairing + spectrum", "inverse passes",
         "result staging + list + stores"]
for var in os.environ.get("PROBE_VARIANTS", "real16:rc,real16:plain,real32:rc,real32:plain").split(","):
    kern, _, lists = var.partition(":")
    os.environ["CM2_OS_KERNEL"], os.environ["CM2_OS_LISTS"] = kern, lists or "rc"
    N = BlockLO(nt // nb, bands, offdiag=True, method=3)
    b = torch.zeros_like(a)
    hop = 4096 if kern == "real16" else 12288
    nwin = nb * (-(-(nt // nb) // hop))
    st = torch.zeros(nwin * 8, dtype=torch.int64, device=dev)
    def run():
        _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), D.stream())
    lib.cm2_os_debug_stamps(None)
    run(); run(); torch.cuda.synchronize()
    lib.cm2_os_debug_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.cm2_os_debug_stamps(None)
    s = st.cpu().numpy().reshape(nwin, 8).astype(np.float64)
    dt = np.diff(s[:, :6], axis=1)
    tot = s[:, 5] - s[:, 0]
    span = (s[:, 5].max() - s[:, 0].min())
    out = {"variant": var, "ms": round(e0.elapsed_time(e1), 4), "windows": nwin,
           "ticks_per_window_mean": round(float(tot.mean()), 1), "kernel_span_ticks": float(span),
           "ticks_per_us_estimate": round(float(span) / (1e3 * e0.elapsed_time(e1)), 1)}
    for i, nme in enumerate(names):
        out[nme] = {"mean": round(float(dt[:, i].mean()), 1), "p10": round(float(np.percentile(dt[:, i], 10)), 1),
                    "p90": round(float(np.percentile(dt[:, i], 90)), 1)}
    print(json.dumps(out), flush=True)
    del N, b, st
