#!/usr/bin/env python3
"""N^-1 on the TIME order (k_os_real<32, 0, false>: the same transform, pairing and exchanges, streaming loads and
stores, no lists) at C4 size: the compute side of the tile-order kernel plus the cheapest possible memory side."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                # noqa: E402
import bench                # noqa: E402
from cosmomap2_amd.interfaces import BlockLO                       # noqa: E402
cfg = bench.CONFIGS["c4"]
nt, nb, lam = cfg["nt"], cfg["nb"], cfg["lam"]
dev = torch.device("cuda", 0)
inp = bench.synth_inputs(torch, dev, 12 * cfg["nside"] ** 2, nt, nb, lam, rank=0)
N = BlockLO(nt // nb, inp["bands"], offdiag=True, method=3)
d = inp["d"]
for _ in range(5):
    N * d
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    N * d
e1.record()
torch.cuda.synchronize()
print(json.dumps({"N^-1 on the time order, ms per 1e8 samples (incl. the output allocation of N * d)": round(e0.elapsed_time(e1) / 20, 4)}))
