"""Time cm2_Pt_tiles_apply alone on a C4-sized tile plan (development probe)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, linearoperators as L
from types import SimpleNamespace
nside = int(os.environ.get("PROBE_NSIDE", "256")); nt = int(float(os.environ.get("PROBE_NT", "1e8")))
npix = 12 * nside * nside; pol = 3
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
skew = float(os.environ.get("PROBE_SKEW", "0"))          # this share of the samples on 10 % of the map
if skew > 0:
    hot = torch.rand(nt, generator=g, device=dev) < skew
    pix[hot] = (pix[hot].to(torch.int64) % (npix // 10)).to(torch.int32)
phi = 0.3 + 0.0785 * torch.arange(nt, device=dev, dtype=torch.float64)
ang = SimpleNamespace(cos=torch.cos(2 * phi), sin=torch.sin(2 * phi))
ang._d_cos, ang._d_sin = ang.cos, ang.sin
P = SparseLO(npix, nt, pix, pol=pol, angle_processed=ang)
T = L._sparse_tiles(P)
v = torch.rand(T.nvalid, generator=g, device=dev, dtype=torch.float64)
out = D.empty(pol * npix)
def run():
    _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v), D.ptr(out), D.stream())
res = {}
for dbg in ["0"]:
    run(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    res[dbg] = round(float(np.median([a.elapsed_time(b) for a, b in evs])), 4)
print(json.dumps({"tile_pixels": T.tile_pixels, "tiles": T.ntiles, "fixed": T.pt_fixed, "fixed_order_info": T.fixed_order_info(), "nvalid": T.nvalid, "ms": res}))
