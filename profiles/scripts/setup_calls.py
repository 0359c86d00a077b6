"""Where the wall clock of the one-off setup goes at C4 size: time per C entry point (each call
followed by a synchronisation) and the Python-side remainder."""
import os, sys, json, time, collections, gc
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
acc = collections.defaultdict(float)
cnt = collections.Counter()
orig = _hip.call
def timed_call(name, *a):
    sync(); t0 = time.perf_counter(); orig(name, *a); sync()
    acc[name] += time.perf_counter() - t0; cnt[name] += 1
for rep in range(2):
    acc.clear(); cnt.clear()
    pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
    phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    bands = [toeplitz_band(lam, rng) for _ in range(nb)]
    x = torch.rand(pol * npix, device=dev, dtype=torch.float64)
    sync()
    _hip.call = timed_call
    for mod in (L, sys.modules["cosmomap2_amd.utilities.process_ces"]):
        pass
    gc.disable()                # (a generation-2 collection inside the timed region costs 10-20 ms)
    t0 = time.perf_counter()
    stmts = {}
    def lap(name, t=[t0]):
        sync(); now = time.perf_counter(); stmts[name] = round(now - t[0], 4); t[0] = now
    N = BlockLO(nt // nb, bands, offdiag=True, method=3); lap("BlockLO")
    ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi); lap("ProcessTimeSamples")
    npc = ces.get_new_pixel[0]; lap("get_new_pixel")
    P = SparseLO(npc, nt, pix, pol=pol, angle_processed=ces); lap("SparseLO")
    M = BlockDiagonalPreconditionerLO(ces, npc, pol=pol); lap("M_BD")
    A = P.T * N * P
    T = L._sparse_tiles(P); lap("tile plan")
    y = A * x[:pol * npc]; lap("first matvec")
    sync()
    total = time.perf_counter() - t0
    gc.enable()
    _hip.call = orig
    inc = sum(acc.values())
    print(json.dumps({"rep": rep, "total_s": round(total, 4), "in_C_entry_points_s": round(inc, 4),
                      "python_and_torch_s": round(total - inc, 4), "statements_s": stmts,
                      "calls": {k: [cnt[k], round(v, 4)] for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:14]}}), flush=True)
    del N, ces, P, M, A, T, y, pix, phi, x
    gc.collect(); sync()        # (operator graphs hold reference cycles: without this the previous
                                #  build's plans are destroyed somewhere inside the next build's timing)
