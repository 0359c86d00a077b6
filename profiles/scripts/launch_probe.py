"""Host-side cost of one P^T N^-1 P application (operator layer + three launches), measured on a
problem small enough that the GPU is never the bottleneck: what bounds a strongly scaled shard."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D
from cosmomap2_amd.interfaces import SparseLO, BlockLO, linearoperators as L
from cosmomap2_amd.utilities import ProcessTimeSamples
from bench import toeplitz_band
nside, nt, nb, lam, pol = 64, 1 << 20, 8, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rng = np.random.default_rng(0)
pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
phi = 0.3 + 0.07 * torch.arange(nt, device=dev, dtype=torch.float64)
ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi)
n = ces.get_new_pixel[0]
P = SparseLO(n, nt, pix, pol=pol, angle_processed=ces)
N = BlockLO(nt // nb, [toeplitz_band(lam, rng) for _ in range(nb)], offdiag=True, method=3)
A = P.T * N * P
x = torch.rand(pol * n, device=dev, dtype=torch.float64)
assert L._use_tiles(P)
for _ in range(20): y = A * x
torch.cuda.synchronize()
K = 2000
t0 = time.perf_counter()
for _ in range(K): y = A * x
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps({"nt": nt, "host_us_per_matvec": round(1e6 * t_issue / K, 1),
                  "wall_us_per_matvec": round(1e6 * t_all / K, 1)}))

# ---- the PCG iteration: wall time per iteration against the kernels of an iteration ---------
# (PROBE_PCG_NT samples, C4's pixelisation by default; the gap is what the host adds between
# iterations: 0.3 ms before the stop test's read of ||r||^2 was deferred behind the launches of
# the next iteration, solvers.py)
import cosmomap2_amd
from cosmomap2_amd.interfaces import BlockDiagonalPreconditionerLO
nt2 = int(os.environ.get("PROBE_PCG_NT", "100000000"))
nside2, nb2 = int(os.environ.get("PROBE_PCG_NSIDE", "256")), 100
npix2 = 12 * nside2 * nside2
del A, P, N, ces, pix, phi, x
pix = torch.randint(0, npix2, (nt2,), generator=g, device=dev, dtype=torch.int32)
phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt2, device=dev, dtype=torch.float64)
ces = ProcessTimeSamples(pix, npix2, pol=pol, phi=phi)
del phi
n = ces.get_new_pixel[0]
P = SparseLO(n, nt2, pix, pol=pol, angle_processed=ces)
N = BlockLO(nt2 // nb2, [toeplitz_band(lam, rng) for _ in range(nb2)], offdiag=True, method=3)
A = P.T * N * P
M = BlockDiagonalPreconditionerLO(ces, n, pol=pol)
b = P.T * (N * torch.rand(nt2, generator=g, device=dev, dtype=torch.float64))
x = torch.rand(pol * n, device=dev, dtype=torch.float64)


def ev_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a_, b_ in e:
        a_.record(); fn(); b_.record()
    torch.cuda.synchronize()
    return float(np.median([a_.elapsed_time(b_) for a_, b_ in e]))


t_mv = ev_ms(lambda: A * x)
t_m = ev_ms(lambda: M * x)
t_dot = ev_ms(lambda: D.dot_dev(x, x))
cosmomap2_amd.cg(A, b, M=M, rtol=1e-6, maxiter=2)
torch.cuda.synchronize()
K2 = 12                                   # fixed number of iterations: rtol far below reach
its = []
t0 = time.perf_counter()
cosmomap2_amd.cg(A, b, M=M, rtol=1e-30, maxiter=K2, callback=lambda v: its.append(1))
torch.cuda.synchronize()
t_it = 1e3 * (time.perf_counter() - t0) / K2
kern = t_mv + t_m + 3 * t_dot + 2 * t_dot        # matvec, M_BD, three dots, the two fused updates (~a dot each)
print(json.dumps({"pcg_nt": nt2, "iterations": len(its), "ms_per_iteration_wall": round(t_it, 4),
                  "matvec_ms": round(t_mv, 4), "M_BD_ms": round(t_m, 4), "dot_ms": round(t_dot, 4),
                  "kernels_per_iteration_ms_estimate": round(kern, 4),
                  "host_gap_per_iteration_ms": round(t_it - kern, 4)}))
