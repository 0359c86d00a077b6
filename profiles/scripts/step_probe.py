"""Where does the step spend time beyond its three kernels?  (development probe)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from cosmomap2_amd.utilities import ProcessTimeSamples
from bench import toeplitz_band
nside, nt, nb, lam, pol = 256, 100_000_000, 100, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
phi = 0.3 + 0.0785 * torch.arange(nt, device=dev, dtype=torch.float64)
ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi); del phi
P = SparseLO(npix, nt, pix, pol=pol, angle_processed=ces)
rng = np.random.default_rng(0)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
A = P.T * N * P
x = torch.rand(pol * npix, generator=g, device=dev, dtype=torch.float64)
T = L._sparse_tiles(P)
d_tb, v_tb, out = D.empty(T.nvalid), D.empty(T.nvalid), D.empty(pol * npix)
st = D.stream()
def direct():
    _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st)
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d_tb), D.ptr(v_tb), st)
    _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st)
def timed(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / k
def host_only(fn, k=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); return 1e3 * (t1 - t0) / k
def direct_fresh_out():
    o = D.empty(pol * npix)
    _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st)
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d_tb), D.ptr(v_tb), st)
    _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(o), st)
    return o
op = A._compiled()[0]
w0, w1 = op._work if op._work is not None else (None, None)
def direct_op_buffers():
    a, b = op._work
    _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(a), st)
    _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), st)
    _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(b), D.ptr(out), st)
A * x
res = {"operator_ms": timed(lambda: A * x), "direct_calls_ms": timed(direct),
       "direct_fresh_out_ms": timed(direct_fresh_out), "direct_op_buffers_ms": timed(direct_op_buffers),
       "op_mult_ms": timed(lambda: op._mult(x)),
       "operator_host_ms": host_only(lambda: A * x), "direct_host_ms": host_only(direct)}
for name, fn in (("P", lambda: _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st)),
                 ("N", lambda: _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d_tb), D.ptr(v_tb), st)),
                 ("Pt", lambda: _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st))):
    res[name + "_alone_ms"] = timed(fn)
res["operator_again_ms"] = timed(lambda: A * x)
res["op_mult_again_ms"] = timed(lambda: op._mult(x))
res["operator_third_ms"] = timed(lambda: A * x)
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
