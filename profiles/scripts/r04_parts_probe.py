"""What does sharing a tile's slices out to several workgroups cost by itself?  Uniform pointing at C4 size
(every tile 195 k samples), the fixed-order P^T with every tile whole, cut into 2, into 4 parts
(CM2_TILE_BALANCE=parts, CM2_PT_PARTS=<samples>): kernel time alone, 10 repetitions, one process."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, linearoperators as L

nside, nt, pol = 256, int(os.environ.get("PROBE_NT", "100000000")), 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
if os.environ.get("PROBE_UNEVEN"):                      # half of the samples on the first tenth of the map
    hot = torch.rand(nt, generator=g, device=dev) < 0.5
    pix[hot] = pix[hot] % (npix // 10)
    del hot
phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
ang = type("A", (), {})()
ang.cos, ang.sin = torch.cos(2 * phi), torch.sin(2 * phi)
del phi
v = torch.rand(nt, generator=g, device=dev, dtype=torch.float64)
if os.environ.get("PROBE_HOT"):                         # 5 % of the samples on one pixel
    pix[torch.rand(nt, generator=g, device=dev) < 0.05] = npix // 3
ref = None
plans = (("whole tiles", {}), ("2 parts", {"CM2_TILE_BALANCE": "parts", "CM2_PT_PARTS": "100000"}),
         ("4 parts", {"CM2_TILE_BALANCE": "parts", "CM2_PT_PARTS": "49000"}), ("whole tiles again", {}))
if os.environ.get("PROBE_HOT"):
    plans = (("default", {}), ("cut", {"CM2_TILE_BALANCE": "cut"}), ("default again", {}))
elif os.environ.get("PROBE_UNEVEN"):
    plans = (("default", {}), ("cut", {"CM2_TILE_BALANCE": "cut"}), ("parts 200000", {"CM2_PT_PARTS": "200000"}),
             ("parts 120000", {"CM2_PT_PARTS": "120000"}), ("parts 54000", {"CM2_PT_PARTS": "54000"}),
             ("one workgroup per tile", {"CM2_TILE_BALANCE": "0"}), ("default again", {}))
for name, env in plans:
    for k in ("CM2_TILE_BALANCE", "CM2_PT_PARTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    P = SparseLO(npix, nt, pix, pol=pol, angle_processed=ang)
    T = L._sparse_tiles(P)
    v_tb, out = D.empty(T.nvalid), D.empty(pol * npix)
    _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(v), D.ptr(v_tb), D.stream())

    def run():
        _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), D.stream())
    run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(json.dumps({"plan": name, "tiles": T.ntiles, "ms": round(float(np.median([a.elapsed_time(b) for a, b in evs])), 4),
                      "fx_slice": T.fixed_order_info()[0], "designed_GB": round(T.fixed_order_info()[1] / 1e9, 4), "max_diff_vs_whole": err, **T.pt_parts()}), flush=True)
    del T, P, v_tb, out
