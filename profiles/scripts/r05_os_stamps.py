#!/usr/bin/env python3
"""Where a window's time goes in k_os_real on the tile order: s_memtime at the phase boundaries (diagnostic build
-DCM2_OS_STAMPS of cm2_overlap_save.hip, loaded through CM2_LIB_PATH; every lane stores its stamp so that the kernel
stays one basic block).  Phases: load window (lists, gathers, staging) | forward passes | pairing | inverse passes |
results.  Prints mean / median clocks per phase over all windows of one launch at C4 size, and the busiest 1 %.
    python3 profiles/scripts/build_variant.py stamps --flags=-DCM2_OS_STAMPS
    CM2_LIB_PATH=profiles/scripts/_variants/lib_stamps.so python3 profiles/scripts/r05_os_stamps.py [c4|c5]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402
from cosmomap2_amd import _hip, device as D                                  # noqa: E402
from cosmomap2_amd.interfaces import SparseLO, BlockLO                       # noqa: E402
from cosmomap2_amd.interfaces import linearoperators as L                    # noqa: E402
from cosmomap2_amd.utilities import ProcessTimeSamples                       # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "c4"
cfg = bench.CONFIGS[key]
pol, nside, nt, nb, lam = 3, cfg["nside"], cfg["nt"], cfg["nb"], cfg["lam"]
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
inp = bench.synth_inputs(torch, dev, npix, nt, nb, lam, rank=0)
pix, phi = inp["pix"], inp.pop("phi")
N = BlockLO(nt // nb, inp["bands"], offdiag=True, method=3)
ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi)
del phi
n = ces.get_new_pixel[0]
P = SparseLO(n, nt, pix, pol=pol, angle_processed=ces)
A = P.T * N * P
x = torch.rand(pol * n, generator=torch.Generator(device=dev).manual_seed(7), device=dev, dtype=torch.float64)
for _ in range(5):
    A * x
torch.cuda.synchronize()
lib = _hip.load()
fn = getattr(lib, "cm2_os_debug_stamps")
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
nwin_max = 1 << 20
buf = torch.zeros(8 * nwin_max, dtype=torch.int64, device=dev)
assert fn(buf.data_ptr()) == 0
T = L._sparse_tiles(P)
tb, tb2 = D.empty(T.nvalid), D.empty(T.nvalid)
_hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(tb), D.stream())
_hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(tb), D.ptr(tb2), D.stream())
torch.cuda.synchronize()
fn(None)
st = buf.cpu().numpy().reshape(-1, 8)
st = st[(st[:, 0] != 0) & (st[:, 5] != 0)]
names = ["load window", "forward passes", "pairing", "inverse passes", "results"]
d = np.diff(st[:, :6].astype(np.float64), axis=1)
tot = st[:, 5] - st[:, 0]
out = {"config": key, "windows": int(st.shape[0]), "counter": "s_memtime",
       "mean": {nm: round(float(d[:, i].mean()), 1) for i, nm in enumerate(names)},
       "median": {nm: round(float(np.median(d[:, i])), 1) for i, nm in enumerate(names)},
       "window_total_mean": round(float(tot.mean()), 1), "window_total_p99": round(float(np.percentile(tot, 99)), 1),
       "launch_span": int(st[:, 5].max() - st[:, 0].min())}
out["share"] = {nm: round(out["mean"][nm] / out["window_total_mean"], 3) for nm in names}
if (st[:, 6] != 0).all():
    # inside the load phase: start -> lists and run tables in (one round trip + barrier) -> first half's gathers in
    # (address decode of both halves + second round trip) -> end of the phase (staging of both halves, four barriers)
    a, b, c = st[:, 6] - st[:, 0], st[:, 7] - st[:, 6], st[:, 1] - st[:, 7]
    out["load_window_split_mean"] = {"lists arrive": round(float(a.mean()), 1), "decode + gathers arrive": round(float(b.mean()), 1),
                                     "staging to registers": round(float(c.mean()), 1)}
print(json.dumps(out))
