#!/usr/bin/env python3
"""Is the timed loop of bench.py a burst figure?  K matvecs back to back take 30 ms at C4 (20 x 1.46 ms); the chip's
power management reacts on a longer scale.  This probe runs the same operator for several seconds, one HIP event
pair per matvec, and prints the time per matvec by position in the run, the three kernels' shares at the start and
at the end (event-timed stage launches), and the engine clock rocm-smi reports while the loop runs.
    python3 profiles/scripts/r05_sustained_probe.py [c4|c5] [seconds]
"""
import json
import os
import subprocess
import sys
import threading
import time

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
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
cfg = bench.CONFIGS["c5" if key == "c5w" else key]
pol, nside, nt, nb, lam = 3, cfg["nside"], cfg["nt"], cfg["nb"], cfg["lam"]
if key == "c5w":                                        # C5 WHOLE on one GPU: 1e9 samples, 64 detector blocks
    nt, nb = cfg["total"], cfg["total_nb"]
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
A * x
torch.cuda.synchronize()
T = L._sparse_tiles(P)
tb, tb2, out = D.empty(T.nvalid), D.empty(T.nvalid), D.empty(pol * n)
stages = [lambda: _hip.call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(tb), D.stream()),
          lambda: _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(tb), D.ptr(tb2), D.stream()),
          lambda: _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(tb2), D.ptr(out), D.stream())]


def stage_times(reps=5):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for r in range(reps):
        ev[r][0].record()
        for k, f in enumerate(stages):
            f()
            ev[r][k + 1].record()
    torch.cuda.synchronize()
    return [round(float(np.median([ev[r][k].elapsed_time(ev[r][k + 1]) for r in range(reps)])), 4) for k in range(3)]


def alone_times(reps=5):
    """every kernel on its own: `reps` launches back to back between two events; and N^-1 once behind an idle gap"""
    res = []
    for f in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / reps, 4))
    stages[0]()
    torch.cuda.synchronize()
    time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    stages[1]()
    e1.record()
    torch.cuda.synchronize()
    return res, round(e0.elapsed_time(e1), 4)


clocks, stop = [], threading.Event()


def sample_clocks():
    while not stop.is_set():
        t = time.perf_counter()
        try:
            o = subprocess.run([sys.executable, os.path.realpath("/opt/rocm/bin/rocm-smi"), "--showclocks", "--json"],
                               capture_output=True, text=True, timeout=5).stdout
            d = json.loads(o)
            card = next(iter(d.values()))
            clocks.append((round(t - t_start, 2), {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower()}))
        except Exception as exc:                         # noqa: BLE001
            clocks.append((round(t - t_start, 2), "rocm-smi: %s" % type(exc).__name__))
        stop.wait(0.25)


time.sleep(1.0)                                          # idle: the state a fresh bench process starts from
t_start = time.perf_counter()
cold_stages = stage_times()
th = threading.Thread(target=sample_clocks)
th.start()
est = sum(cold_stages)
nrep = int(seconds * 1e3 / est)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(nrep + 1)]
ev[0].record()
for i in range(nrep):
    A * x
    ev[i + 1].record()
torch.cuda.synchronize()
hot_stages = stage_times()
alone, n_after_idle = alone_times()
stop.set()
th.join()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(nrep)])
marks = [0, 20, 50, 100, 200, 400, 800, 1600, 3200]
series = {"%d-%d" % (a, a + 20): round(float(ms[a:a + 20].mean()), 4) for a in marks if a + 20 <= nrep}
series["last20"] = round(float(ms[-20:].mean()), 4)
print(json.dumps({"config": key, "matvecs": nrep, "seconds": round(float(ms.sum()) / 1e3, 2),
                  "ms_per_matvec_by_position": series,
                  "stages_ms_after_idle [P, N^-1, P^T]": cold_stages, "stages_ms_after_the_run": hot_stages,
                  "each_kernel_alone_back_to_back_ms": alone, "N^-1_once_behind_a_50ms_idle_gap_ms": n_after_idle,
                  "clocks_during_the_run": clocks[:: max(1, len(clocks) // 12)]}))
