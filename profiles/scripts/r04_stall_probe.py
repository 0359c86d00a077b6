"""Where the 12-35 ms of one cm2_bd_det_mask call go in a warm setup.
Every C entry point of a C4-sized setup is timed as (launch: the call returns) + (wait: the device is
idle again), the wait done in one of two ways, alternating per repetition:
  block : torch.cuda.synchronize() -- a blocking wait of the runtime;
  spin  : the host polls hipStreamQuery until the stream is idle (no blocking wait at all).
Calls longer than 3 ms are listed with both parts, with the entry points that ran just before, and
with the library's allocation counters (driver allocations = cache misses) around them."""
import collections, ctypes, gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosmomap2_amd import device as D, _hip
from cosmomap2_amd.interfaces import SparseLO, BlockLO, BlockDiagonalPreconditionerLO, linearoperators as L
from cosmomap2_amd.utilities import ProcessTimeSamples
from bench import toeplitz_band
nside, nt, nb, lam, pol = 256, 100_000_000, 100, 2048, 3
npix = 12 * nside * nside
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rng = np.random.default_rng(0)
torch.empty(1, device=dev); _hip.load(); torch.cuda.synchronize()
orig = _hip.call
mode = "block"
log = []


def misses():
    m = (ctypes.c_int64 * 4)()
    orig("cm2_device_memory_info", m)
    return int(m[3])


def wait_idle():
    if mode == "block":
        torch.cuda.synchronize()
    else:
        st = torch.cuda.current_stream()
        while not st.query():
            pass


def timed_call(name, *a):
    wait_idle()
    m0 = misses()
    t0 = time.perf_counter(); orig(name, *a); t1 = time.perf_counter(); wait_idle(); t2 = time.perf_counter()
    log.append((name, t1 - t0, t2 - t1, misses() - m0))


reps = int(os.environ.get("PROBE_REPS", "12"))
for rep in range(reps):
    # PROBE_REP1 = block | spin: how repetition 1 (the first warm build, where the stall was seen in
    # round 3) waits; the modes alternate from there
    first = os.environ.get("PROBE_REP1", "block")
    mode = first if rep % 2 == 1 else ("spin" if first == "block" else "block")
    del log[:]
    pix = torch.randint(0, npix, (nt,), generator=g, device=dev, dtype=torch.int32)
    phi = 0.3 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    bands = [toeplitz_band(lam, rng) for _ in range(nb)]
    x = torch.rand(pol * npix, device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    _hip.call = timed_call
    gc.disable()
    t0 = time.perf_counter()
    N = BlockLO(nt // nb, bands, offdiag=True, method=3)
    ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi)
    npc = ces.get_new_pixel[0]
    P = SparseLO(npc, nt, pix, pol=pol, angle_processed=ces)
    extra = {}
    if os.environ.get("PROBE_NOOP"):
        # a trivial kernel on memory that exists already, launched where cm2_bd_det_mask would be: does
        # the delay belong to the moment or to cm2_bd_det_mask's own (new) output buffers?
        wait_idle(); tn = time.perf_counter(); x[:16].fill_(1.0); wait_idle()
        extra["noop_kernel_ms"] = round(1e3 * (time.perf_counter() - tn), 3)
    seg0 = torch.cuda.memory_stats().get("segment.all.allocated", 0)
    M = BlockDiagonalPreconditionerLO(ces, npc, pol=pol)
    extra["torch_segments_allocated_by_M_BD"] = torch.cuda.memory_stats().get("segment.all.allocated", 0) - seg0
    A = P.T * N * P
    T = L._sparse_tiles(P)
    y = A * x[:pol * npc]
    wait_idle()
    total = time.perf_counter() - t0
    gc.enable()
    _hip.call = orig
    slow = [{"call": n, "launch_ms": round(1e3 * a, 3), "wait_ms": round(1e3 * b, 3), "driver_allocations": m,
             "previous": [log[j][0] for j in range(max(0, i - 2), i)]}
            for i, (n, a, b, m) in enumerate(log) if a + b > 3e-3 and n not in
            ("cm2_weights_accumulate", "cm2_tiles_create", "cm2_tiles_prepare_pt")]
    print(json.dumps({"rep": rep, "wait": mode, "total_s": round(total, 4),
                      "sum_calls_s": round(sum(a + b for _, a, b, _ in log), 4),
                      "bd_det_mask_ms": [round(1e3 * (a + b), 3) for n, a, b, _ in log if n == "cm2_bd_det_mask"],
                      "calls_that_went_to_the_driver": [[n, m] for n, _, _, m in log if m],
                      "torch_segments_total": torch.cuda.memory_stats().get("segment.all.allocated", 0),
                      "slow_calls": slow, **extra}), flush=True)
    if os.environ.get("PROBE_KEEP"):
        # nothing of a repetition is released (host or device): does the delay need the previous build's
        # objects to be destroyed?
        keep = globals().setdefault("_keep", [])
        keep.append((N, ces, P, M, A, T, y, pix, phi, x, bands))
    del N, ces, P, M, A, T, y, pix, phi, x
    gc.collect(); torch.cuda.synchronize()
