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
