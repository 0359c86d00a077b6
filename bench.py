#!/usr/bin/env python3
"""
bench.py -- TOD samples/s through one P^T N^-1 P matvec (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]
                    [--scaling weak|strong]

A "step" is one application of A = P^T N^-1 P to a map-domain vector with all inputs
already resident in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU; a step is
the local matvec of the rank's block-aligned TOD shard followed by the RCCL all-reduce of the
map.  --scaling weak (default): every rank owns a shard of the configuration's per-GPU size
(C4: 1e8 samples per GPU); --scaling strong: the configuration's TOTAL is cut into N shards
(C4: 1e8 samples over N ranks; C5: 1e9 over 8).  With N > 1 the JSON line carries the other
mode's point too ("other_scaling_point"), the backend and world size torch.distributed
reports, and the exposed all-reduce time (step minus the local matvec).  Rank 0 prints ONE
JSON line.

Workloads (BASELINE.json configs; synthetic inputs after utilities_functions.py:99-212:
uniform-random pixel per sample, HWP angle ramp, d ~ U[0,1)):
  c2  nside 128 IQU, 1e7 samples, diagonal N (100 blocks)        -> fused single kernel
  c3  nside 128 IQU, 1e8 samples, banded-Toeplitz N, lambda 2048 -> P, overlap-save FFT, P^T
  c4  nside 256 IQU, 1e8 samples/GPU, Toeplitz lambda 2048       -> the configuration the
      north_star target (>= 40 % HBM roofline) is quoted on; DEFAULT.
  c5  one GPU's share of C5: nside 512 IQU, 1.25e8 samples (8 detector blocks of 15 625 000);
      `--config c5 --scaling strong --gpus 1` is C5 WHOLE on one GPU (1e9 samples, 64 blocks)

Byte accounting per stage: "bytes_survey" is SURVEY.md 8(d)'s algorithmic figure (P 28, N^-1 16,
P^T 28 B/sample + map traffic) -- the figure `roofline.achieved` is computed from;
"bytes_designed" is what the kernel is built to move (half-angle storage, address lists, window
overlap), which is what to compare the PMC traffic with.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    "c2": dict(nside=128, nt=10_000_000, nb=100, lam=0, total=10_000_000, total_nb=100,
               label="C2 nside128 IQU 1e7 diag-N"),
    "c3": dict(nside=128, nt=100_000_000, nb=100, lam=2048, total=100_000_000, total_nb=100,
               label="C3 nside128 IQU 1e8 Toeplitz(2048)"),
    "c4": dict(nside=256, nt=100_000_000, nb=100, lam=2048, total=100_000_000, total_nb=100,
               label="C4 nside256 IQU 1e8/GPU Toeplitz(2048)"),
    # one GPU's share of C5 (1e9 samples, 64 detector blocks over 8 GPUs): 8 blocks of 15 625 000
    "c5": dict(nside=512, nt=125_000_000, nb=8, lam=2048, total=1_000_000_000, total_nb=64,
               label="C5 share: nside512 IQU 1.25e8/GPU, 8 detector blocks, Toeplitz(2048)"),
}


def toeplitz_band(lam, rng, fknee=0.02, alpha=1.5):
    """First row of an SPD banded-Toeplitz INVERSE noise covariance with a 1/f knee
    (build-defined input, SURVEY 8d: the reference's noise_val draws i.i.d. uniforms, which
    is not positive definite at lambda = 2048).  Inverse spectrum 1 / (1 + (fknee/f)^alpha)
    (high-pass: low frequencies are down-weighted), transformed to lags, tapered to `lam`
    lags with a Hann window, and lifted so that the truncated spectrum stays positive."""
    Lg = 8 * lam
    f = np.fft.rfftfreq(Lg)
    H = 1.0 / (1.0 + (fknee / np.maximum(f, 0.25 * f[1])) ** alpha)
    a = np.fft.irfft(H, Lg)[:lam].copy()
    a *= 0.5 * (1.0 + np.cos(np.pi * np.arange(lam) / lam))
    g = np.zeros(Lg)
    g[:lam] = a
    g[Lg - lam + 1:] = a[1:][::-1]
    Ht = np.fft.rfft(g).real
    floor = 2e-3 * Ht.max()
    if Ht.min() < floor:
        a[0] += floor - Ht.min()
    return a * (1.0 + 0.1 * rng.random())


def synth_inputs(torch, dev, npix, nt, nb, lam, rank=0):
    """The synthetic inputs of one rank's shard, generated in HBM (seeded; the seed differs per rank),
    after the reference's generators (utilities/utilities_functions.py:99-212): uniform-random pixel
    per sample (pairs_gen :111-122), HWP angle ramp phi = theta0 + 2 pi 2.5/200 i (angles_gen :99-109),
    d ~ U[0,1) (:197), one banded-Toeplitz first row (`toeplitz_band`) or one diagonal weight in
    [0.5, 1.5) per noise block.  Shared by bench.py and the full-size parity tests
    (tests/test_gpu_fullsize.py), which hand the same arrays to the CPU oracle."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(20161202 + 1000 * rank)
    rng = np.random.default_rng(20161202 + 1000 * rank)
    pix = torch.randint(0, npix, (nt,), generator=gen, device=dev, dtype=torch.int32)
    theta0 = float(rng.uniform(0, np.pi))
    phi = theta0 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    d = torch.rand(nt, generator=gen, device=dev, dtype=torch.float64)
    bands = [toeplitz_band(lam, rng) for _ in range(nb)] if lam else None
    diag = None if lam else rng.random(nb) + 0.5
    return dict(gen=gen, rng=rng, theta0=theta0, pix=pix, phi=phi, d=d, bands=bands, diag=diag)


def hbm_budget_gb(nt_rank, n, r, arnoldi_steps, world, layout):
    """Device memory one rank needs, from the per-sample table of DESIGN.md section 2 (measured at C5 whole:
    profiles/r04_bench_c5_whole.json `hbm_memory`) and the map-domain vectors of either layout.  Printed for
    the strong-scaling series so that the first multi-GPU run of a configuration cannot fail on memory."""
    per_sample = {"inputs pix i32 + d f64": 12.0, "cos 2phi, sin 2phi": 16.0,
                  "tile plan (pl u16, half angle f64, tb_dst u32)": 14.0,
                  "fixed-order P^T lists": 12.2, "overlap-save lists": 5.1,
                  "TOD scratch of the operator (two tile-order buffers)": 16.0,
                  "transient: phi f64 + ProcessTimeSamples' pixel-sorted index": 28.0}
    rows = n / world if layout == "rows" else n
    vec = {"PCG vectors (x, r, p, z, q, b) + exchange buffers": 8.0 * (6 * rows + 2 * n),
           "Arnoldi basis, 2 x (steps + 1) vectors": 8.0 * 2 * (arnoldi_steps + 1) * rows,
           "Z, AZ": 8.0 * 2 * r * rows,
           "per-pixel weights + M_BD blocks (13 arrays of npix)": 8.0 * 13 * n / 3}
    samples = sum(per_sample.values()) * nt_rank
    total = samples + sum(vec.values())
    return {"samples_GB": round(samples / 1e9, 1), "map_domain_GB": round(sum(vec.values()) / 1e9, 1),
            "total_GB": round(total / 1e9, 1), "fits_288_GB": bool(total < 0.9 * 288e9)}


def measure_traffic(kernel_regex, child_args, timeout=420):
    """HBM traffic of one kernel per launch from the PMC counters, measured by THIS run: two child processes of this
    script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need a pass each: the TCC block has four counter
    slots, FETCH_SIZE takes three, WRITE_SIZE two -- /opt/skills/guides/MI355X_MICROARCH.md, rocprofv3 PMC slots;
    counters only, no other trace domain than the kernel trace).  The children repeat the workload with few steps
    and every extra leg switched off; the median over the kernel's launches is taken.  Both counters are KiB per
    dispatch; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so traffic = 2 * FETCH_SIZE + WRITE_SIZE
    (the guide's HBM section).  Children are started with subprocess (never exec), program after `--` is python3."""
    import csv
    import glob
    import statistics
    import tempfile
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--kernel-include-regex", kernel_regex, "-d", d,
                   "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__)] + child_args
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr and kernel_regex in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                raise RuntimeError("rocprofv3 --pmc %s gave no row for %s (rc %d): %s"
                                   % (ctr, kernel_regex, res.returncode, (res.stderr or res.stdout)[-300:]))
            got[ctr] = {"bytes": statistics.median(vals) * 1024.0, "launches": len(vals)}
    return 2.0 * got["FETCH_SIZE"]["bytes"] + got["WRITE_SIZE"]["bytes"], got


class Heartbeat(object):
    """A line on stderr every `period` seconds while a long host-side leg runs (the oracle's PCG at 1e9 samples takes
    minutes: a GPU box takes seven silent minutes for a hang and ends the run)."""

    def __init__(self, what, period=60.0):
        import threading
        self.what, self.period, self.t0 = what, period, time.perf_counter()
        self.stop = threading.Event()
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop.wait(self.period):
            print("bench.py: %s, %.0f s so far" % (self.what, time.perf_counter() - self.t0), file=sys.stderr, flush=True)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()
        return False


def under_profiler():
    keys = " ".join(k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")))
    return bool(keys) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def device_state():
    """What rocm-smi says about the card this process runs on (compute / memory partition mode, VRAM in use,
    clocks): recorded with every line because the SAME commit measured 11.3 ms and 8.1 ms for N^-1 at C5 whole on
    two boxes of the same pool (profiles/r05_c5_whole.md) -- a property of the box, not of the run."""
    out = {}
    if under_profiler():
        # (rocm-smi is a `#!/usr/bin/env python3` script: under rocprofv3 the profiler's preloaded library initialises
        #  the GPU in every process of that exec chain, and a process that has initialised the GPU must not exec)
        return {"skipped": "inside a profiled run"}
    try:
        import shutil
        smi = os.path.realpath(shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi")
        res = subprocess.run([sys.executable, smi, "--showcomputepartition", "--showmemorypartition", "--showmeminfo",
                              "vram", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        cards = json.loads(res.stdout)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        out["cards_listed"] = len([k for k in cards if k.startswith("card")])
        keep = ("partition", "vram", "sclk", "mclk")
        for name, card in sorted(cards.items()):
            if not name.startswith("card"):
                continue
            out[name] = {k.strip(): v for k, v in card.items() if any(w in k.lower() for w in keep)}
        if vis:
            out["visible_devices"] = vis
    except Exception as exc:                                  # noqa: BLE001
        out["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:120])
    return out


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(gpus, argv, port):
    """The command `python bench.py --gpus N` runs for N > 1 when it was NOT started by
    torch.distributed.run itself: one rank per GPU of this node, rendezvous on 127.0.0.1 --
    the same line the driver uses (bench.py's own arguments are passed through unchanged)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(int(gpus)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def launch_ranks(gpus, argv):
    """Start the N ranks as a CHILD process (never exec: nothing in this process has touched the
    GPU, and it must stay that way), relay rank 0's JSON line, and fail unless the job really ran
    on N ranks.  Returns the exit code."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launcher_command(gpus, argv, _free_port())
    print("bench.py: starting %d ranks: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print("bench.py: torch.distributed.run exited with %d" % proc.returncode, file=sys.stderr)
        return proc.returncode or 1
    if line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    rec = json.loads(line)
    ws = (rec.get("distributed") or {}).get("world_size")
    if rec.get("n_gpus") != gpus or ws != gpus:
        print("bench.py: asked for %d ranks, the job reports n_gpus=%r world_size=%r"
              % (gpus, rec.get("n_gpus"), ws), file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("CM2_BENCH_CONFIG", "c4"),
                    choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--layout", default="replicated", choices=["replicated", "rows"],
                    help="N > 1: map-domain vectors replicated on every rank (one all-reduce of the map per "
                         "matvec) or row-sharded (all-gather + reduce-scatter; M_BD, M2 and the vector updates "
                         "on a rank's rows only); the other layout's step is measured too (other_layout_point)")
    ap.add_argument("--nt", type=int, default=0, help="override samples per GPU")
    ap.add_argument("--lam", type=int, default=0, help="override the Toeplitz band length")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pcg", action="store_true", help="skip the PCG iteration count")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the full-size comparison of the timed path with the CPU oracle")
    ap.add_argument("--parity-host-seconds", type=float, default=150.0,
                    help="host time the oracle's PCG solves of the parity_full_size block may take (estimated "
                         "from its first matvec); C5 whole needs about 400")
    ap.add_argument("--parity-arnoldi", action="store_true",
                    help="parity_full_size also runs the oracle's own Arnoldi (reference recurrence, "
                         "interfaces/deflationlib.py:17-113) for --arnoldi-steps steps on the host and reports the "
                         "principal angles between its Ritz space and the GPU's (minutes of host time)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child runs (copied from the newest "
                         "committed profile instead); implied by --no-cpu and inside a profiled run")
    ap.add_argument("--no-raster", action="store_true",
                    help="skip the secondary run with a coherent raster-scan pointing")
    ap.add_argument("--no-filters", action="store_true",
                    help="skip the FilterLO / GroundFilterLO timing (SURVEY 8f rows)")
    ap.add_argument("--no-other-point", action="store_true",
                    help="N > 1: skip the other scaling mode's point")
    ap.add_argument("--deflation", type=int, default=32,
                    help="rank of the deflation space of the two-level PCG run (0 = skip)")
    ap.add_argument("--arnoldi-steps", type=int, default=96)
    ap.add_argument("--fft-len", type=int, default=0)
    ap.add_argument("--toeplitz", default="fused", choices=["fused", "rocfft"],
                    help="overlap-save implementation for the Toeplitz configs")
    args = ap.parse_args()
    if args.fft_len:
        os.environ["CM2_FFT_LEN"] = str(args.fft_len)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before torch is imported
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # roofline.traffic: the two rocprofv3 --pmc passes are CHILD processes of this script, started HERE -- before
    # this process imports torch or touches the GPU (a process that has initialised the GPU must not be the ancestor
    # of an exec chain on this pool: rocprofv3 is a launcher that re-execs its target).  The dominant kernel of a
    # configuration is known from its noise model; the result is matched against the measured one below.
    traffic_pre = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not (args.no_traffic or args.no_cpu)
            and not under_profiler()):
        pre_regex = "k_os_real" if (CONFIGS[args.config]["lam"] and args.toeplitz == "fused") else (
            "k_PtNP_sell" if not CONFIGS[args.config]["lam"] else None)
        child = ["--config", args.config, "--scaling", args.scaling, "--gpus", "1", "--steps", "3", "--warmup", "1",
                 "--toeplitz", args.toeplitz, "--no-cpu", "--no-filters", "--no-raster", "--no-pcg", "--no-parity",
                 "--no-traffic", "--deflation", "0"]
        if args.nt:
            child += ["--nt", str(args.nt)]
        if args.lam:
            child += ["--lam", str(args.lam)]
        if pre_regex:
            t_tr = time.perf_counter()
            try:
                traffic, passes = measure_traffic(pre_regex, child)
                traffic_pre = {"regex": pre_regex, "traffic": traffic, "passes": passes,
                               "seconds": round(time.perf_counter() - t_tr, 1)}
            except Exception as exc:                          # noqa: BLE001  (auxiliary: falls back to the copy)
                traffic_pre = {"regex": pre_regex, "error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CM2_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than
        # ranks (ranks then share a card); the driver's runs use RCCL, one rank per GPU.
        backend = os.environ.get("CM2_DIST_BACKEND", "nccl")
        ndev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank % ndev if backend != "nccl" else local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(2)

    import cosmomap2_amd
    from cosmomap2_amd import device as D
    from cosmomap2_amd import _hip
    from cosmomap2_amd.build import library_stamp
    from cosmomap2_amd.interfaces import SparseLO, BlockLO, BlockDiagonalPreconditionerLO
    from cosmomap2_amd.interfaces import linearoperators as L
    from cosmomap2_amd.utilities import ProcessTimeSamples
    from cosmomap2_amd.sharding import (ShardedLO, make_sync, shard_blocks, RowShards, RowShardedNormalLO,
                                        row_sharded_bd, row_sharded_two_level)

    cfg = dict(CONFIGS[args.config])
    if args.nt:
        cfg["nt"] = args.nt
        cfg["total"] = args.nt
        cfg["total_nb"] = cfg["nb"]
    if args.lam and cfg["lam"]:
        cfg["lam"] = args.lam
        cfg["label"] += " [lambda=%d]" % args.lam
    pol = 3
    npix = 12 * cfg["nside"] ** 2
    lam = cfg["lam"]
    dev = torch.device("cuda", torch.cuda.current_device())

    def sync():
        torch.cuda.synchronize()

    def strong_budget():
        return {"bytes_per_sample_table": "DESIGN.md section 2",
                **{"N=%d" % k: {lay: hbm_budget_gb(-(-cfg["total"] // k), pol * npix, args.deflation,
                                                   args.arnoldi_steps, k, lay) for lay in ("replicated", "rows")}
                   for k in (1, 2, 4, 8)}}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- a rank's shard in one scaling mode, generated in HBM (seed differs per rank) -------
    def shard_geometry(scaling):
        """(samples, noise blocks) of this rank's shard."""
        if scaling == "weak":
            nb = cfg["nb"]
            return (cfg["nt"] // nb) * nb, nb
        # strong: the configuration's TOTAL cut into `world` block-aligned shards (world = 1: the
        # whole of it on one GPU -- C5: 1e9 samples, 64 detector blocks; the N = 1 point of the
        # strong-scaling series)
        tnb = cfg["total_nb"]
        bsz = cfg["total"] // tnb
        b0, b1, s0, s1 = shard_blocks([bsz] * tnb, world, rank)
        return s1 - s0, b1 - b0

    def build_shard(scaling):
        nt, nb = shard_geometry(scaling)
        bsize = nt // nb
        tm = {}
        t0 = time.time()
        inp = synth_inputs(torch, dev, npix, nt, nb, lam, rank)
        gen, rng, theta0, pix, d = (inp[k] for k in ("gen", "rng", "theta0", "pix", "d"))
        phi = inp.pop("phi")
        sync()
        tm["generate_inputs"] = time.time() - t0
        t0 = time.time()
        bands = inp["bands"]
        if lam:
            N = BlockLO(bsize, bands, offdiag=True, method=(3 if args.toeplitz == "fused" else 2))
            w = None
        else:
            N = BlockLO(bsize, list(inp["diag"]), offdiag=False)
            w = N._device_diag()
        sync()
        tm["noise_operator"] = time.time() - t0
        t0 = time.time()
        allred = (lambda t: dist.all_reduce(t)) if world > 1 else None
        ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi, w=w, allreduce=allred)
        del phi
        sync()
        tm["weights_mask_compaction"] = time.time() - t0
        npix_c = ces.get_new_pixel[0]
        t0 = time.time()
        P = SparseLO(npix_c, nt, pix, pol=pol, angle_processed=ces)
        Mbd = BlockDiagonalPreconditionerLO(ces, npix_c, pol=pol)
        sync()
        tm["pointing_operator_and_M_BD"] = time.time() - t0
        A_local = P.T * N * P
        n = pol * npix_c
        x = torch.rand(n, generator=torch.Generator(device=dev).manual_seed(7), device=dev,
                       dtype=torch.float64)
        # the distributed operator in both layouts of the map-domain vectors (same seed on every rank:
        # x is the same whole vector everywhere; x_rows = this rank's rows of it)
        sh = RowShards(npix_c, pol) if world > 1 else None
        # (persistent_output: results and exchange vectors live in buffers the operators keep -- no
        #  allocation, zero-fill or copy per matvec, the same addresses for RCCL every time; the timed
        #  loop, the PCG and the Arnoldi build consume A p before the next application)
        A_repl = ShardedLO(A_local, persistent_output=True) if world > 1 else A_local
        A_rows = RowShardedNormalLO(A_local, sh, persistent_output=True) if world > 1 else A_local
        x_rows = sh.local(x) if world > 1 else x
        A = A_rows if args.layout == "rows" else A_repl
        if lam and L._use_tiles(P):
            t0 = time.time()
            L._sparse_tiles(P)
            sync()
            tm["tile_plan_and_fixed_order_PT_lists"] = time.time() - t0
            t0 = time.time()
            A_local * x            # first application: the overlap-save kernel's address lists
            sync()
            tm["first_matvec_overlap_save_lists"] = time.time() - t0
        # The first ~20 matvecs after an idle period run ~3 % slower than the steady state (the
        # same 20-step loop measured again later in the process is faster by that much): part of
        # the setup is therefore 0.15 s of untimed matvecs, so that the W warm-up steps and the K
        # timed steps that follow see the GPU in the state a solver run keeps it in.
        t0 = time.time()
        while time.time() - t0 < 0.15:
            for _ in range(5):
                A_local * x
            sync()
        tm["steady_state_spinup"] = time.time() - t0
        return dict(nt=nt, nb=nb, bsize=bsize, pix=pix, d=d, bands=bands, N=N, w=w, ces=ces,
                    npix_c=npix_c, P=P, Mbd=Mbd, A_local=A_local, A=A, n=n, x=x, rng=rng,
                    gen=gen, theta0=theta0, setup=tm, sh=sh, A_repl=A_repl, A_rows=A_rows, x_rows=x_rows)

    def timed(A, x, steps, warmup):
        for _ in range(warmup):
            y = A * x
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = A * x
        # The end of the queue is polled before the closing barrier + synchronize.  (Rounds 2-3 added this
        # against "a blocking wait that wakes late"; round 4 found the real cause of those one-off 10-30 ms
        # delays -- the kernel driver evicting the process's GPU queues when pageable host memory that the
        # HIP runtime had pinned for a copy is freed, profiles/r04_stall_probe.md -- and removed it from
        # the library's copies.  The poll stays: it does not change what is timed.)
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        del y
        return elapsed

    if args.scaling == "strong" and cfg["total"] != cfg["nt"]:
        cfg["label"] += " [strong scaling: the whole configuration, %d samples in %d blocks, over %d GPU(s)]" % (
            cfg["total"], cfg["total_nb"], world)
    t_setup = time.time()
    S = build_shard(args.scaling)
    sync()
    t_setup = time.time() - t_setup
    S_setup = S["setup"]
    nt, nb, bsize, npix_c, n = S["nt"], S["nb"], S["bsize"], S["npix_c"], S["n"]
    P, N, A, A_local, Mbd, x, d, pix, ces = (S[k] for k in ("P", "N", "A", "A_local", "Mbd", "x",
                                                               "d", "pix", "ces"))
    bands, w, gen, theta0 = S["bands"], S["w"], S["gen"], S["theta0"]

    sh, rows = S["sh"], (args.layout == "rows" and world > 1)
    x_in = S["x_rows"] if rows else x                # the vector the distributed operator acts on
    # ---- timed region: W warmup + exactly K steps ------------------------------------
    elapsed = timed(A, x_in, args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    nt_all = nt
    if world > 1:
        tot = torch.tensor([float(nt)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        nt_all = int(tot.item())
    value = nt_all / (elapsed / args.steps)

    # exposed all-reduce: the same K steps of the local matvec alone
    dist_info = None
    if world > 1:
        el_local = timed(A_local, x, args.steps, args.warmup)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "layout": args.layout,
                     "local_matvec_ms": round(1e3 * el_local / args.steps, 4),
                     "exposed_collective_ms": round(ms_per_step - 1e3 * el_local / args.steps, 4)}
        if rows:
            dist_info.update({"collectives_per_matvec": "all-gather + reduce-scatter of %d rows per rank" % sh.rows,
                              "bytes_sent_per_rank_per_matvec": sh.bytes_per_matvec()})
        else:
            dist_info.update({"allreduce_bytes": 8 * n,
                              # the collective choice the operator made (max over ranks of the shard sizes)
                              "allreduce_chunks": A.allreduce_chunks(nt),
                              "map_allreduces_per_matvec": A.collectives_issued // (args.steps + args.warmup)})
        # per-rank device memory of THIS configuration's strong-scaling series (total cut over N ranks), both
        # layouts: computed, not measured -- printed so that a first hardware run cannot fail on memory
        dist_info["hbm_budget_per_rank_strong_scaling"] = strong_budget()
        # the same K steps in the OTHER layout of the map-domain vectors
        # (auxiliary: a failure here is reported in the line, it does not take the headline down.  Every
        # rank takes the same branch: an exception on one rank only would leave the others in a collective,
        # which the launcher's timeout then ends.)
        A_o, x_o = (S["A_repl"], x) if rows else (S["A_rows"], S["x_rows"])
        try:
            el_o = timed(A_o, x_o, args.steps, args.warmup)
            dist_info["other_layout_point"] = {"layout": "replicated" if rows else "rows",
                                               "ms_per_step": round(1e3 * el_o / args.steps, 4),
                                               "value": nt_all / (el_o / args.steps), "unit": "TOD samples/s"}
        except Exception as exc:                          # noqa: BLE001
            dist_info["other_layout_point"] = {"layout": "replicated" if rows else "rows",
                                               "error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- per-kernel HIP-event timing on the launch stream (rank 0) --------------------
    def ev_time(fn, reps):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return float(np.mean(ts)), float(ts[len(ts) // 2])

    def seq_time(fns, reps):
        """The callables launched one after the other, `reps` times back to back, an event between
        every two launches: each kernel's duration with the caches in the state the step leaves them
        in (a kernel relaunched on its own inputs finds part of its data in the 256 MB Infinity Cache
        and measures 3-12 % faster than inside a step; profiles/r03_step_gaps.json)."""
        for fn in fns:
            fn()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)] for _ in range(reps)]
        for row in evs:
            row[0].record()
            for i, fn in enumerate(fns):
                fn()
                row[i + 1].record()
        torch.cuda.synchronize()
        out = []
        for i in range(len(fns)):
            ts = sorted(row[i].elapsed_time(row[i + 1]) for row in evs)
            out.append((float(np.mean(ts)), float(ts[len(ts) // 2])))
        # the whole sequence between the same events (first to last of a repetition): the stage times
        # add up to this by construction
        whole = [row[0].elapsed_time(row[-1]) for row in evs]
        seq_time.last_step_ms = float(np.mean(whole))
        return out

    seq_time.last_step_ms = None
    reps = max(5, min(args.steps, 20))
    stages = {}          # name -> ((mean ms, median ms), bytes_survey, bytes_designed)
    stage_timing = "each kernel alone, relaunched on its own inputs"
    map_bytes = 48.0 * npix_c
    tile_info = None
    if lam:
        tod = P * x
        tod2 = N * tod
        if L._use_tiles(P):
            T = L._sparse_tiles(P)
            nv = T.nvalid
            ang = 8.0 if T.half_angle else 16.0          # half-angle double, or cos and sin
            st = D.stream
            d_tb = D.empty(T.nvalid)
            out = D.empty(n)
            call = _hip.call
            tile_info = {"tile_pixels": T.tile_pixels, "tiles": T.ntiles,
                         "half_angle_storage": T.half_angle,
                         "pt_order": {0: "atomic", 1: "fixed (time order per pixel, hot runs in fixed chunks, "
                                      "no atomics)", 2: "exact (time order per pixel)"}[T.pt_mode]}
            pt_name = "P^T tiles (k_Pt_tiles_fixed)" if T.pt_fixed else "P^T tiles (k_Pt_tiles)"
            pt_designed = ((8.0 + 4.0 + ang) if T.pt_fixed else (8.0 + 2.0 + ang)) * nv + map_bytes / 2
            if args.toeplitz == "fused":
                # the three kernels of a step in sequence, events between them
                v_tb = D.empty(T.nvalid)
                p_ms, os_ms, pt_ms = seq_time([
                    lambda: call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st()),
                    lambda: call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d_tb), D.ptr(v_tb), st()),
                    lambda: call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st())], reps)
                stage_timing = "P, N^-1, P^T launched in sequence, events between the kernels"
            else:
                p_ms = ev_time(lambda: call("cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st()), reps)
            stages["P tiles (k_P_tiles)"] = (p_ms, 28.0 * nt + map_bytes / 2,
                                             (2.0 + ang + 8.0) * nv + map_bytes / 2)
            if args.toeplitz == "fused":
                kinfo = N.tile_kernel_info()               # which kernel ran, its list format
                tile_info["overlap_save"] = kinfo
                os_name = ("k_os_real<%s>, one real window of %d samples per workgroup, %s lists"
                           % (kinfo["os_kernel"][4:], kinfo["os_window"], kinfo["os_lists"]))
                stages["N^-1 on tile order (%s)" % os_name] = (
                    os_ms, 16.0 * nt, float(kinfo["tile_bytes_per_sample"]) * nv)
                del v_tb
            else:
                stages["tiles->time (k_tiles_to_time)"] = (ev_time(lambda: call(
                    "cm2_tod_tiles_to_time", T.h, D.ptr(d_tb), D.ptr(tod), st()), reps),
                    16.0 * nt, 22.0 * nt)
                stages["N^-1 (overlap-save rocFFT)"] = (ev_time(lambda: N * tod, reps), 16.0 * nt,
                                                        None)
                stages["time->tiles (k_time_to_tiles)"] = (ev_time(lambda: call(
                    "cm2_tod_time_to_tiles", T.h, D.ptr(tod2), D.ptr(d_tb), st()), reps),
                    16.0 * nt, 22.0 * nt)
            if args.toeplitz != "fused":
                pt_ms = ev_time(lambda: call("cm2_Pt_tiles_apply", T.h, D.ptr(d_tb), D.ptr(out), st()), reps)
            stages[pt_name] = (pt_ms, 28.0 * nt + map_bytes / 2, pt_designed)
            del d_tb, out
        else:
            stages["P (k_P_time)"] = (ev_time(lambda: P * x, reps), 28.0 * nt + map_bytes / 2,
                                      28.0 * nt + map_bytes / 2)
            stages["N^-1"] = (ev_time(lambda: N * tod, reps), 16.0 * nt, 24.0 * nt)
            stages["P^T (k_Pt_sell)"] = (ev_time(lambda: P.T * tod2, reps),
                                         28.0 * nt + map_bytes / 2, 32.0 * nt + map_bytes / 2)
        step_bytes = 72.0 * nt + map_bytes
        del tod, tod2
    else:
        stages["P^T diag(w) P fused (k_PtNP_sell)"] = (ev_time(lambda: A_local * x, reps),
                                                       28.0 * nt + map_bytes, 24.0 * nt + map_bytes)
        step_bytes = 28.0 * nt + map_bytes
    dom = max(stages, key=lambda k: stages[k][0][0])
    (dom_mean, dom_med), dom_bytes, dom_designed = stages[dom]
    # One consistent set of times.  The K timed steps run WITHOUT events between the kernels
    # (ms_per_step); the stage times come from the same three launches with an event between every
    # two, and add up to the step measured between the same events (`step_ms_same_events`, equal to the
    # sum by construction).  An event between two kernels keeps the command processor from fetching the
    # next dispatch while the previous kernel drains: `event_gap_ms` = (step with events - step without)
    # / kernels is what each event interval contains beside its kernel, and `avg_launch_ms` -- the
    # figure rocprofv3's average duration of the same kernel is to be compared with -- is the interval
    # minus that gap (`avg_launch_ms_event_interval` is the raw interval).
    timing = {"kind": stage_timing}
    gap = 0.0
    if seq_time.last_step_ms is not None and world == 1:
        nk = len(stages)
        ssum = sum(v[0][0] for v in stages.values())
        gap = max(0.0, (seq_time.last_step_ms - ms_per_step) / nk)
        timing.update({"stages_sum_ms": round(ssum, 4), "step_ms_same_events": round(seq_time.last_step_ms, 4),
                       "step_ms_no_events": round(ms_per_step, 4), "event_gap_ms": round(gap, 4),
                       "stages_sum_minus_gaps_ms": round(ssum - nk * gap, 4)})
    dom_net = dom_mean - gap
    achieved = dom_bytes / (dom_net * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "traffic_measured_in_run": False,
                "algorithmic_bytes": dom_bytes, "designed_bytes": dom_designed,
                "avg_launch_ms": round(dom_net, 4), "avg_launch_ms_event_interval": round(dom_mean, 4)}
    # HBM traffic of the dominant kernel from the PMC counters.  FETCH_SIZE / WRITE_SIZE need
    # rocprofv3 passes of their own, so the figure is NOT measured in this run: it is copied from
    # the newest committed summary of this command (profiles/make_summary.py), only when workload
    # and size are the same, and labelled with that profile's file and commit.
    if traffic_pre is not None:
        # measured by this run (the two rocprofv3 --pmc children at the top of main), if they profiled the kernel
        # that turned out to be the dominant one
        if "error" in traffic_pre:
            roofline["traffic_measurement_error"] = traffic_pre["error"]
        elif traffic_pre["regex"] in dom:
            passes = traffic_pre["passes"]
            roofline.update({"traffic": traffic_pre["traffic"], "traffic_measured_in_run": True,
                             "traffic_passes": {"FETCH_SIZE_bytes": passes["FETCH_SIZE"]["bytes"],
                                                "WRITE_SIZE_bytes": passes["WRITE_SIZE"]["bytes"],
                                                "launches_per_pass": passes["FETCH_SIZE"]["launches"],
                                                "formula": "2 * FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts "
                                                           "128-byte requests at 64 bytes)",
                                                "how": "two child processes of bench.py under rocprofv3 --pmc, started "
                                                       "before this process touched the GPU",
                                                "seconds": traffic_pre["seconds"]}})
        else:
            roofline["traffic_measurement_error"] = "profiled %s, dominant kernel is %s" % (traffic_pre["regex"], dom)
    if roofline["traffic"] is None:
        try:
            pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_c4.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
            if pmc["workload"] == cfg["label"] and pmc["nt_per_gpu"] == nt:
                for kname, rec in pmc["kernels"].items():
                    if kname in dom and rec.get("hbm_traffic_bytes"):
                        roofline["traffic"] = rec["hbm_traffic_bytes"]
                        roofline["traffic_source"] = "profiles/" + pmcs[-1]
                        roofline["traffic_source_commit"] = pmc.get("commit")
                        roofline["traffic_source_sources_sha16"] = pmc.get("library_sources_sha16")
        except Exception:
            pass
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    stage_report = {}
    for k, v in stages.items():
        ms = v[0][0]
        rec = {"ms": round(ms, 4), "bytes_survey": v[1]}
        if v[2] is None or v[2] >= v[1]:
            # a rate on SURVEY's algorithmic bytes is only a bandwidth when the kernel moves at
            # least that much: k_P_tiles / k_Pt_tiles are built to move fewer bytes than SURVEY
            # counts (half-angle storage), and bytes_survey / time there is a throughput figure
            # that can exceed the HBM peak -- it is reported as samples/s only
            rec["GB/s_survey"] = round(v[1] / (ms * 1e-3) / 1e9, 1)
        else:
            rec["samples_per_s"] = round(nt / (ms * 1e-3), 1)
        if v[2] is not None:
            rec["bytes_designed"] = v[2]
            rec["GB/s_designed"] = round(v[2] / (ms * 1e-3) / 1e9, 1)
            rec["frac_of_hbm_peak_designed"] = round(v[2] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        stage_report[k] = rec

    # ---- PCG iterations to 1e-6 (outside the timed region) ----------------------------
    pcg = None
    if not args.no_pcg and rows:
        # PCG on row-sharded vectors: M_BD and the two-level preconditioner act on this rank's rows, the
        # three dot products of an iteration are 8-byte all-reduces, Z^T r an r-vector all-reduce
        from cosmomap2_amd.interfaces import ritz_deflation_basis
        b_loc = sh.reduce_scatter(P.T * (N * d))
        Mr = row_sharded_bd(ces, sh)

        def solve(Mop):
            cosmomap2_amd.cg(A, b_loc, M=Mop, rtol=1e-6, maxiter=1, dot_reduce=sh.allreduce_)
            torch.cuda.synchronize()
            best, its_ = float("inf"), []
            for _ in range(2):
                its_ = []
                barrier()
                tp = time.perf_counter()
                xs_, info_ = cosmomap2_amd.cg(A, b_loc, M=Mop, rtol=1e-6, maxiter=500,
                                              callback=lambda xk: its_.append(1), dot_reduce=sh.allreduce_)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - tp)
            return xs_, int(info_), len(its_), best
        xs, info, nit, t_pcg = solve(Mr)
        pcg = {"rtol": 1e-6, "layout": "rows", "iters": nit, "info": info, "seconds": round(t_pcg, 4),
               "ms_per_iteration": round(1e3 * t_pcg / max(1, nit), 4), "timed_runs": 2,
               "preconditioner": "block-diagonal (a rank's pixels)"}
        if lam and args.deflation > 0:
            r = args.deflation
            tz = time.perf_counter()
            Zl, theta, AZl = ritz_deflation_basis(A, Mr, b_loc, r, args.arnoldi_steps, with_AZ=True, shards=sh)
            M2 = row_sharded_two_level(Mr, Zl, AZl, sh, apply='eig')
            torch.cuda.synchronize()
            t_build = time.perf_counter() - tz
            xs2, info2, nit2, t_pcg2 = solve(M2)
            pcg["two_level"] = {"rank": r, "arnoldi_steps": args.arnoldi_steps, "iters": nit2, "info": info2,
                                "seconds": round(t_pcg2, 4),
                                "ms_per_iteration": round(1e3 * t_pcg2 / max(1, nit2), 4),
                                "build_seconds": round(t_build, 3),
                                "Z_rows_per_rank": int(sh.rows),
                                "rel_l2_vs_block_diagonal_solution": float(
                                    (sh.gather(xs2) - sh.gather(xs)).norm() / sh.gather(xs).norm())}
            del Zl, AZl, M2, xs2
        del b_loc, xs, Mr
    elif not args.no_pcg:
        b = P.T * (N * d)
        if world > 1:
            dist.all_reduce(b)
        cosmomap2_amd.cg(A, b, M=Mbd, rtol=1e-6, maxiter=1, sync=make_sync())       # warm-up
        torch.cuda.synchronize()
        # (two timed solves, their MEAN reported and both listed)
        t_runs = []
        for _ in range(2):
            its = []
            tp = time.perf_counter()
            xs, info = cosmomap2_amd.cg(A, b, M=Mbd, rtol=1e-6, maxiter=500,
                                        callback=lambda xk: its.append(1), sync=make_sync())
            torch.cuda.synchronize()
            t_runs.append(time.perf_counter() - tp)
        t_pcg = sum(t_runs) / len(t_runs)
        pcg = {"rtol": 1e-6, "iters": len(its), "info": int(info),
               "seconds": round(t_pcg, 4), "ms_per_iteration": round(1e3 * t_pcg / max(1, len(its)), 4),
               "timed_runs": 2, "seconds_each_run": [round(v, 4) for v in t_runs],
               "preconditioner": "block-diagonal",
               "true_relative_residual": float(torch.linalg.vector_norm(b - A * xs)
                                               / torch.linalg.vector_norm(b))}
        if lam and args.deflation > 0:
            # two-level preconditioner with an Arnoldi/Ritz deflation space (BASELINE config C4)
            from cosmomap2_amd.interfaces import (DeflationLO, CoarseLO, TwoLevelPreconditionerLO,
                                                  ritz_deflation_basis, apply_to_columns)
            from cosmomap2_amd.utilities import write_ritz_eigenvectors, read_ritz_eigenvectors
            r = args.deflation
            tz = time.perf_counter()
            # Z = V U and A Z = P (H U) from the Arnoldi relation: both one pass over the stored
            # basis (cm2_panel_gemm); the reference's r extra applications of A are timed beside it
            Z, theta, AZ = ritz_deflation_basis(A, Mbd, b, r, args.arnoldi_steps, with_AZ=True)
            torch.cuda.synchronize()
            t_ritz = time.perf_counter() - tz
            ta = time.perf_counter()
            AZ_explicit = apply_to_columns(A, Z)
            torch.cuda.synchronize()
            t_az = time.perf_counter() - ta
            az_rel = float(torch.linalg.matrix_norm(AZ - AZ_explicit) / torch.linalg.matrix_norm(AZ_explicit))
            del AZ_explicit
            tz += t_az                                   # (the explicit product is not part of the build)
            te = time.perf_counter()
            Zd, AZd = DeflationLO(Z), DeflationLO(AZ)
            E = CoarseLO(Z, AZ, r, apply='eig')
            M2 = TwoLevelPreconditionerLO(Mbd, Zd, AZd, E)
            torch.cuda.synchronize()
            t_e = time.perf_counter() - te
            t_build = time.perf_counter() - tz
            cosmomap2_amd.cg(A, b, M=M2, rtol=1e-6, maxiter=1, sync=make_sync())   # warm-up
            torch.cuda.synchronize()
            t_runs2 = []
            for _ in range(2):
                its2 = []
                tp = time.perf_counter()
                xs2, info2 = cosmomap2_amd.cg(A, b, M=M2, rtol=1e-6, maxiter=500,
                                              callback=lambda xk: its2.append(1), sync=make_sync())
                torch.cuda.synchronize()
                t_runs2.append(time.perf_counter() - tp)
            t_pcg2 = sum(t_runs2) / len(t_runs2)
            rel = float(torch.linalg.vector_norm(xs2 - xs) / torch.linalg.vector_norm(xs))
            # one application of M2: Z^T r, the r x r solve, fused tail over Z and AZ
            rr = torch.rand(n, generator=torch.Generator(device=dev).manual_seed(9), device=dev,
                            dtype=torch.float64)
            m2_mean, m2_med = ev_time(lambda: M2 * rr, reps)
            work = D.reduce_work()
            y32 = D.empty(r)
            zt_mean, _ = ev_time(lambda: _hip.call("cm2_Zt_apply", n, r, D.ptr(Z), D.ptr(rr),
                                                   D.ptr(y32), D.ptr(work), D.stream()), reps)
            zy = D.empty(n)
            z_mean, _ = ev_time(lambda: _hip.call("cm2_Z_apply", n, r, D.ptr(Z), D.ptr(y32),
                                                  D.ptr(zy), D.stream()), reps)
            gw = D.empty(int(_hip.load().cm2_gemm_tn_work_doubles(r, r)))
            dE = D.empty(r * r)
            g_mean, _ = ev_time(lambda: _hip.call("cm2_gemm_tn", n, r, r, D.ptr(Z), D.ptr(AZ),
                                                  D.ptr(dE), D.ptr(gw), D.stream()), reps)
            zbytes = 8.0 * n * r
            m2_bytes = 3 * zbytes + 8.0 * 3 * n + 56.0 * npix_c
            two = {"rank": r, "arnoldi_steps": args.arnoldi_steps,
                   "iters": len(its2), "info": int(info2),
                   "seconds": round(t_pcg2, 4), "seconds_each_run": [round(v, 4) for v in t_runs2],
                   "ms_per_iteration": round(1e3 * t_pcg2 / max(1, len(its2)), 4),
                   "build_seconds": round(t_build, 3),
                   "build_split_seconds": {"arnoldi_ritz_vectors_and_AZ": round(t_ritz, 3),
                                           "coarse_matrix_and_operators": round(t_e, 3)},
                   "AZ_by_r_matvecs_seconds": round(t_az, 3),
                   "AZ_arnoldi_relation_vs_matvecs_rel": az_rel,
                   "smallest_ritz": float(theta[0]), "largest_kept_ritz": float(theta[-1]),
                   "rel_l2_vs_block_diagonal_solution": rel,
                   "M2_apply": {"ms": round(m2_mean, 4), "bytes": m2_bytes,
                                "GB/s": round(m2_bytes / (m2_mean * 1e-3) / 1e9, 1),
                                "frac_of_hbm_peak": round(m2_bytes / (m2_mean * 1e-3) / 1e9
                                                          / HBM_PEAK_GBS, 4)},
                   "kernels": {
                       "Z^T r (cm2_Zt_apply)": {
                           "ms": round(zt_mean, 4),
                           "GB/s": round((zbytes + 8.0 * n) / (zt_mean * 1e-3) / 1e9, 1)},
                       "Z y (cm2_Z_apply)": {
                           "ms": round(z_mean, 4),
                           "GB/s": round((zbytes + 8.0 * n) / (z_mean * 1e-3) / 1e9, 1)},
                       "E = Z^T AZ (cm2_gemm_tn, fp64 MFMA)": {
                           "ms": round(g_mean, 4),
                           "GB/s": round(2 * zbytes / (g_mean * 1e-3) / 1e9, 1),
                           "TFLOP/s": round(2.0 * n * r * r / (g_mean * 1e-3) / 1e12, 2)}}}
            # Ritz-vector checkpoint: what a second run pays instead of the build
            if rank == 0:
                import tempfile
                with tempfile.TemporaryDirectory() as tmp:
                    fn = os.path.join(tmp, "ritz_c4")
                    tw = time.perf_counter()
                    write_ritz_eigenvectors(Z, fn, eigvals=theta)
                    t_w = time.perf_counter() - tw
                    tr = time.perf_counter()
                    Zr, thr = read_ritz_eigenvectors(fn, eigvals=True, device=True)
                    AZr = apply_to_columns(A_local, Zr)
                    CoarseLO(Zr, AZr, r, apply='eig')
                    torch.cuda.synchronize()
                    t_r = time.perf_counter() - tr
                    two["checkpoint"] = {"write_seconds": round(t_w, 3),
                                         "build_seconds_from_checkpoint": round(t_r, 3),
                                         "bit_identical_Z": bool(torch.equal(Zr, Z)),
                                         "note": "local operator only" if world > 1 else None}
                    del Zr, AZr
            pcg["two_level"] = two
            del Z, AZ, Zd, AZd, E, M2, xs2, rr, zy, gw
        del b, xs

    # ---- 8(f) rows: sub-scan and ground filters on the same TOD (rank 0, untimed extras) ----
    filters = None
    if rank == 0 and world == 1 and not args.no_filters:
        from cosmomap2_amd.interfaces import FilterLO, GroundFilterLO
        sub, gap = 2000, 40                       # sub-scan and turnaround lengths (samples)
        starts = np.arange(0, bsize - sub + 1, sub + gap)
        sizes = np.full(starts.size, sub)
        pix_f = pix.clone()
        pix_f[torch.rand(nt, generator=gen, device=dev) < 0.05] = -1
        filters = {"subscan_samples": sub, "chunks": int(starts.size) * nb, "flag_fraction": 0.05,
                   "algorithmic_bytes_per_sample": 20}
        for order in (0, 2):
            F = FilterLO(nt, [sizes, starts], bsize, nb, pix_f, poly_order=order)
            mean_ms, med_ms = ev_time(lambda: F * d, reps)
            filters["poly%d_ms" % order] = round(med_ms, 4)
            filters["poly%d_GBps" % order] = round(20.0 * nt / (med_ms * 1e-3) / 1e9, 1)
            # the production operator A = P^T F P (tile-order P / P^T around the filter)
            P_f = SparseLO(npix_c, nt, pix_f, pol=pol, angle_processed=ces)
            A_f = P_f.T * F * P_f
            _, med_a = ev_time(lambda: A_f * x, reps)
            filters["PtFP_poly%d_ms" % order] = round(med_a, 4)
            if order == 0 and not args.no_pcg:
                # PCG on the production system P^T F P x = P^T F d with M_BD
                b_f = P_f.T * (F * d)
                cosmomap2_amd.cg(A_f, b_f, M=Mbd, rtol=1e-6, maxiter=1)
                torch.cuda.synchronize()
                its_f = []
                t_f = time.perf_counter()
                x_f, info_f = cosmomap2_amd.cg(A_f, b_f, M=Mbd, rtol=1e-6, maxiter=500,
                                               callback=lambda xk: its_f.append(1))
                torch.cuda.synchronize()
                filters["PtFP_poly0_pcg"] = {"rtol": 1e-6, "iters": len(its_f), "info": int(info_f),
                                             "seconds": round(time.perf_counter() - t_f, 4)}
                del b_f, x_f
            del F, A_f, P_f
        az = ((torch.arange(nt, device=dev, dtype=torch.int64) % (2 * (sub + gap))) - (sub + gap)
              ).abs().to(torch.int32)             # triangle-wave azimuth -> sub+gap+1 ground bins
        Fg = GroundFilterLO(az)
        mean_ms, med_ms = ev_time(lambda: Fg * d, reps)
        filters["ground_bins"] = Fg.nbins
        filters["ground_ms"] = round(med_ms, 4)
        A_g = P.T * Fg * P
        _, med_g = ev_time(lambda: A_g * x, reps)
        filters["PtGP_ms"] = round(med_g, 4)
        del A_g
        del Fg, az, pix_f

    # ---- secondary pointing model: coherent raster scan -------------------------
    # The headline pointing is the reference generator's uniform-random pixels (pairs_gen), the
    # cache-hostile worst case; a telescope sweeps: every detector block rasters W = 1024 columns
    # back and forth, 4 samples per pixel, one row per sweep, starting at its own row.
    raster = None
    # (the secondary pointing models build a second and a third pointing plan beside the first: at more than
    #  2.5e8 samples on one GPU -- C5 whole -- they are skipped, 1e9 samples x 3 plans do not fit 288 GB)
    secondary = rank == 0 and world == 1 and lam and not args.no_raster and nt <= 250_000_000
    if secondary and npix % 1024 == 0:
        Wc, dwell = 1024, 4
        Hr = npix // Wc
        tt = torch.arange(nt, device=dev, dtype=torch.int64)
        blk, u = tt // bsize, (tt % bsize) // dwell
        sweep, cc = u // Wc, u % Wc
        col = torch.where(sweep % 2 == 0, cc, Wc - 1 - cc)
        row = ((blk * Hr) // nb + sweep) % Hr
        pix_r = (row * Wc + col).to(torch.int32)
        del tt, blk, u, sweep, cc, col, row
        phi_r = theta0 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
        ces_r = ProcessTimeSamples(pix_r, npix, pol=pol, phi=phi_r)
        del phi_r
        n_r = ces_r.get_new_pixel[0]
        P_r = SparseLO(n_r, nt, pix_r, pol=pol, angle_processed=ces_r)
        A_r = P_r.T * N * P_r
        x_r = torch.rand(pol * n_r, generator=torch.Generator(device=dev).manual_seed(8), device=dev,
                         dtype=torch.float64)
        _, med_r = ev_time(lambda: A_r * x_r, reps)
        raster = {"pointing": "raster: %d columns, %d samples per pixel, one row per sweep" % (Wc, dwell),
                  "npix": int(n_r), "ms_per_step": round(med_r, 4),
                  "value": round(nt / (med_r * 1e-3), 1), "unit": "TOD samples/s",
                  "step_frac_of_hbm_peak": round((72.0 * nt + 48.0 * n_r) / (med_r * 1e-3) / 1e9
                                                 / HBM_PEAK_GBS, 4)}
        del A_r, P_r, ces_r, pix_r, x_r

    # ---- uneven hit map: half of the samples on a tenth of the map (tiles re-cut to equal load) --
    uneven = None
    if secondary:
        try:
            gen_u = torch.Generator(device=dev).manual_seed(20161203)
            pix_u = torch.randint(0, npix, (nt,), generator=gen_u, device=dev, dtype=torch.int32)
            hot = torch.rand(nt, generator=gen_u, device=dev) < 0.5
            pix_u[hot] = pix_u[hot] % (npix // 10)
            del hot
            phi_u = theta0 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
            ces_u = ProcessTimeSamples(pix_u, npix, pol=pol, phi=phi_u)
            del phi_u
            n_u = ces_u.get_new_pixel[0]
            P_u = SparseLO(n_u, nt, pix_u, pol=pol, angle_processed=ces_u)
            A_u = P_u.T * N * P_u
            x_u = torch.rand(pol * n_u, generator=torch.Generator(device=dev).manual_seed(9), device=dev,
                             dtype=torch.float64)
            # the uniform step measured the same way in the same minute (an event pair around every single
            # matvec; the headline ms_per_step is K matvecs back to back between two wall-clock reads and is
            # 2-4 % shorter than this figure): the ratio of the two is the cost of the hit map
            _, med_ref = ev_time(lambda: A_local * x, reps)
            _, med_u = ev_time(lambda: A_u * x_u, reps)
            T_u = L._sparse_tiles(P_u)
            tb_u, tb2_u, out_u = D.empty(T_u.nvalid), D.empty(T_u.nvalid), D.empty(pol * n_u)
            st_u = {}
            if args.toeplitz == "fused":
                su = seq_time([lambda: _hip.call("cm2_P_tiles_apply", T_u.h, D.ptr(x_u), D.ptr(tb_u), D.stream()),
                               lambda: _hip.call("cm2_noise_apply_tiles", N._noise.h, T_u.h, D.ptr(tb_u),
                                                 D.ptr(tb2_u), D.stream()),
                               lambda: _hip.call("cm2_Pt_tiles_apply", T_u.h, D.ptr(tb2_u), D.ptr(out_u), D.stream())], 5)
                st_u = {"P": su[0][0], "N^-1": su[1][0], "P^T": su[2][0]}
            del tb_u, tb2_u, out_u
            uneven = {"pointing": "50 % of the samples on the first tenth of the map, the rest uniform",
                      "tiles": int(T_u.ntiles), "widest_tile_pixels": int(T_u.tile_pixels), "pt_parts": T_u.pt_parts(),
                      "ms_per_step": round(med_u, 4), "value": round(nt / (med_u * 1e-3), 1),
                      "unit": "TOD samples/s", "uniform_ms_same_method": round(med_ref, 4),
                      "over_uniform": round(med_u / med_ref - 1.0, 4),
                      "stages_ms_in_sequence": {k: round(v, 4) for k, v in st_u.items()}}
            del A_u, P_u, ces_u, pix_u, x_u, T_u
            # a stare at a source: 5 % of the samples on ONE pixel.  The default fixed-order P^T sums
            # such a run in fixed chunks (reproducible); "exact" walks it term by term with one thread.
            pix_h = torch.randint(0, npix, (nt,), generator=gen_u, device=dev, dtype=torch.int32)
            pix_h[torch.rand(nt, generator=gen_u, device=dev) < 0.05] = npix // 3
            phi_h = theta0 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
            ces_h = ProcessTimeSamples(pix_h, npix, pol=pol, phi=phi_h)
            del phi_h
            n_h = ces_h.get_new_pixel[0]
            P_h = SparseLO(n_h, nt, pix_h, pol=pol, angle_processed=ces_h)
            A_h = P_h.T * N * P_h
            x_h = torch.rand(pol * n_h, generator=torch.Generator(device=dev).manual_seed(10), device=dev,
                             dtype=torch.float64)
            _, med_h = ev_time(lambda: A_h * x_h, reps)
            T_h = L._sparse_tiles(P_h)
            tb_h, out_h = D.empty(T_h.nvalid), D.empty(pol * n_h)
            _hip.call("cm2_P_tiles_apply", T_h.h, D.ptr(x_h), D.ptr(tb_h), D.stream())
            pt_ms = {}
            for mode, name in ((1, "fixed_chunks"), (2, "exact_time_order")):
                T_h.set_pt_order(mode)
                _, pt_ms[name] = ev_time(lambda: _hip.call("cm2_Pt_tiles_apply", T_h.h, D.ptr(tb_h),
                                                           D.ptr(out_h), D.stream()), 3)
            T_h.set_pt_order(1)
            uneven["hot_pixel"] = {"pointing": "5 % of the samples on one pixel, the rest uniform",
                                   "ms_per_step": round(med_h, 4), "over_uniform": round(med_h / med_ref - 1.0, 4),
                                   "tiles": int(T_h.ntiles),
                                   "pt_parts": T_h.pt_parts(),
                                   "PT_ms": {k: round(v, 4) for k, v in pt_ms.items()}}
            del A_h, P_h, ces_h, pix_h, x_h, T_h, tb_h, out_h
        except (_hip.HipError, torch.cuda.OutOfMemoryError) as exc:      # auxiliary: reported, not fatal
            # (at 1e9 samples on one GPU a second and third pointing plan do not always fit beside the first)
            uneven = dict(uneven or {}, error="%s: %s" % (type(exc).__name__, str(exc)[:200]))
            del exc
            pix_u = phi_u = ces_u = P_u = A_u = x_u = T_u = tb_u = tb2_u = out_u = None     # (whatever existed)
            pix_h = phi_h = ces_h = P_h = A_h = x_h = T_h = tb_h = out_h = None
            torch.cuda.empty_cache()
            D.release_cached_memory()

    fft_len = N.noise_info()["fft_len"] if lam else 0

    # ---- the other scaling mode's point (N > 1) --------------------------------------------
    other = None
    if world > 1 and not args.no_other_point:
        other_mode = "strong" if args.scaling == "weak" else "weak"
        del A, A_local, P, N, Mbd, ces, x, d, pix, S
        torch.cuda.empty_cache()
        S2 = build_shard(other_mode)
        el2 = timed(S2["A"], S2["x_rows"] if rows else S2["x"], args.steps, args.warmup)
        el2_local = timed(S2["A_local"], S2["x"], args.steps, args.warmup)
        tot = torch.tensor([float(S2["nt"])], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        other = {"scaling": other_mode, "nt_this_rank": S2["nt"], "nt_all_ranks": int(tot.item()),
                 "ms_per_step": round(1e3 * el2 / args.steps, 4),
                 "value": float(tot.item()) / (el2 / args.steps), "unit": "TOD samples/s",
                 "local_matvec_ms": round(1e3 * el2_local / args.steps, 4),
                 "exposed_allreduce_ms": round(1e3 * (el2 - el2_local) / args.steps, 4)}
        del S2

    # ---- CPU baseline: the oracle (1 core, reference-unfused) on a bounded sample -----
    cpu = cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as orc
        orc.build()
        ns = min(nt, 1_000_000 if lam else 20_000_000)
        ns = (ns // bsize) * bsize if ns >= bsize else ns
        hp = pix[:ns].cpu().numpy()
        hc, hs = ces._d_cos[:ns].cpu().numpy(), ces._d_sin[:ns].cpu().numpy()
        hx = x.cpu().numpy()
        tcpu0 = time.perf_counter()
        repsc = 0
        if lam:
            nblk = max(1, ns // bsize)
            hb = bands[:nblk]
            while True:
                todc = orc.sparse_mult(pol, hp, hc, hs, hx)
                todc = orc.blocklo_mult(bsize if ns >= bsize else ns, hb, True, todc)
                orc.sparse_rmult(pol, npix_c, hp, hc, hs, todc)
                repsc += 1
                if time.perf_counter() - tcpu0 > 10.0:
                    break
            sample = ("%d-sample slice of the same workload (%d block(s)), direct banded "
                      "Toeplitz lambda=%d as interfaces/linearoperators.py:582-595, cost exactly "
                      "linear in samples" % (ns, nblk, lam))
        else:
            hw = w[:ns].cpu().numpy()
            while True:
                orc.ptnp_diag(pol, npix_c, hp, hc, hs, hw, hx)
                repsc += 1
                if time.perf_counter() - tcpu0 > 10.0:
                    break
            sample = "%d-sample slice of the same workload, unfused P, diag(w), P^T" % ns
        tcpu = time.perf_counter() - tcpu0
        cpu = {"value": round(ns * repsc / tcpu, 1), "unit": "TOD samples/s", "cores": 1,
               "kind": "port", "sample": sample, "host_cores_available": os.cpu_count()}
        # the fair host figure (SURVEY 8d ii): all cores, OpenMP pointing loops, FFT Toeplitz
        if lam:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count()
            nsa = (min(nt, 20_000_000) // bsize) * bsize or min(nt, bsize)
            nblk = max(1, nsa // bsize)
            fast = orc.AllCoresMatvec(pol, npix_c, pix[:nsa].cpu().numpy(),
                                      ces._d_cos[:nsa].cpu().numpy(), ces._d_sin[:nsa].cpu().numpy(),
                                      bsize if nsa >= bsize else nsa, bands[:nblk], cores)
            fast(hx)                                                        # warm-up
            t0c, repsa = time.perf_counter(), 0
            while True:
                fast(hx)
                repsa += 1
                if time.perf_counter() - t0c > 8.0:
                    break
            ta = time.perf_counter() - t0c
            cpu_all = {"value": round(nsa * repsa / ta, 1), "unit": "TOD samples/s", "cores": cores,
                       "kind": "port",
                       "sample": "%d-sample slice (%d block(s)); OpenMP P / P^T with per-thread maps, "
                                 "FFT convolution per noise block (scipy.signal.fftconvolve) on a "
                                 "thread pool" % (nsa, nblk)}

    # ---- parity at full size: the timed operator against the CPU oracle on the SAME inputs -------
    # (outside the timed region; rank 0 of a one-GPU run).  The inputs are generated again from the
    # same seeds (the run's own pixel stream has been flagged and renumbered in place by now), copied
    # to the host, and the oracle does ProcessTimeSamples, one A x, the right-hand side and the PCG
    # solve with M_BD itself (oracle.HostProblem: serial reference-order loops for the weights and
    # M_BD, all host threads for P / N^-1 / P^T).  north_star: maps within 1e-6 relative l2,
    # identical iteration counts.
    parity = None
    if rank == 0 and world == 1 and not args.no_parity and not args.no_cpu:
        from oracle import oracle as orc
        tq = time.perf_counter()
        beat = Heartbeat("parity_full_size: the oracle on the host")
        beat.__enter__()
        inp2 = synth_inputs(torch, dev, npix, nt, nb, lam, rank)
        pix_h, phi_h, d_h = inp2["pix"].cpu().numpy(), inp2["phi"].cpu().numpy(), inp2["d"].cpu().numpy()
        bands_h, diag_h = inp2["bands"], inp2["diag"]
        del inp2
        H = orc.HostProblem(pol, npix, pix_h, phi_h, bsize, bands=bands_h, diag=diag_h)
        del phi_h
        parity = {"oracle": "oracle.HostProblem: ProcessTimeSamples, M_BD serial (reference order); "
                            "P / N^-1 / P^T on %d host threads" % H.threads,
                  "same_observed_pixels": bool(H.n == npix_c)}
        if H.n == npix_c:
            hx = x.cpu().numpy()
            t1 = time.perf_counter()
            yo = H.A(hx)
            t_mv = time.perf_counter() - t1
            y = (A * x).cpu().numpy()
            parity["matvec_rel_l2"] = float(np.linalg.norm(y - yo) / np.linalg.norm(yo))
            parity["host_matvec_seconds"] = round(t_mv, 2)
            # the all-cores host figure at the FULL size of the workload (the bounded slice of the
            # cpu_baseline_all_cores leg keeps fewer threads busy: fewer noise blocks than threads)
            parity["host_all_cores_samples_per_s"] = round(nt / t_mv, 1)
            del y, yo
            budget = args.parity_host_seconds
            if not args.no_pcg and pcg is not None and t_mv * (pcg["iters"] + 3) < budget:
                bo = H.rhs(d_h)
                b = P.T * (N * d)
                parity["rhs_rel_l2"] = float(np.linalg.norm(b.cpu().numpy() - bo) / np.linalg.norm(bo))
                t1 = time.perf_counter()
                xo, info_o, its_o = H.solve(bo, rtol=1e-6, maxiter=500)
                parity["host_pcg_seconds"] = round(time.perf_counter() - t1, 1)
                budget -= time.perf_counter() - t1
                itsg = []
                xg, info_g = cosmomap2_amd.cg(A, b, M=Mbd, rtol=1e-6, maxiter=500, callback=lambda xk: itsg.append(1))
                parity["pcg_iters_gpu"] = len(itsg)
                parity["pcg_iters_oracle"] = int(its_o)
                parity["pcg_iters_identical"] = bool(len(itsg) == its_o and info_g == 0 and info_o == 0)
                parity["map_rel_l2"] = float(np.linalg.norm(xg.cpu().numpy() - xo) / np.linalg.norm(xo))
                del xg
                two_g = (pcg or {}).get("two_level")
                if lam and args.deflation > 0 and two_g:
                    # the configuration AS BASELINE STATES IT (C4, C5: two-level preconditioner, deflation
                    # space of dimension 32): Z from the GPU's Arnoldi, everything after it by the oracle --
                    # Az[:, i] = A Z[:, i], E = Z^T Az with the 'eig' pseudo-inverse, M2 = Mbd R + Zd E Zd^T
                    # (src/test_M2_precond_onto_real_data.py:96-112, interfaces/linearoperators.py:969-1056)
                    # and scipy's recurrence with M2
                    r = args.deflation
                    need = t_mv * (r + two_g["iters"] + 3)
                    if need < budget:
                        from cosmomap2_amd.interfaces import (DeflationLO, CoarseLO, TwoLevelPreconditionerLO,
                                                              ritz_deflation_basis, apply_to_columns)
                        t1 = time.perf_counter()
                        Z, theta, AZ = ritz_deflation_basis(A, Mbd, b, r, args.arnoldi_steps, with_AZ=True)
                        E = CoarseLO(Z, AZ, r, apply='eig')
                        M2 = TwoLevelPreconditionerLO(Mbd, DeflationLO(Z), DeflationLO(AZ), E)
                        its2 = []
                        x2, info2 = cosmomap2_amd.cg(A, b, M=M2, rtol=1e-6, maxiter=500,
                                                     callback=lambda xk: its2.append(1))
                        Zh = Z.cpu().numpy()
                        AZo, Eo, M2o = H.two_level(Zh, apply='eig')
                        x2o, info2o, its2o = H.solve(bo, rtol=1e-6, maxiter=500, M=M2o)

                        def rl2(a_, b_):
                            return float(np.linalg.norm(np.asarray(a_) - b_) / np.linalg.norm(b_))
                        tl = {"rank": r, "arnoldi_steps": args.arnoldi_steps,
                              "AZ_rel_l2_r_matvecs": rl2(apply_to_columns(A, Z).cpu().numpy(), AZo),
                              "AZ_rel_l2_arnoldi_relation": rl2(AZ.cpu().numpy(), AZo),
                              "E_rel_l2": rl2(E.E, Eo.E), "E_pinv_rel_l2": rl2(E.invE, Eo.invE),
                              "pcg_iters_gpu": len(its2), "pcg_iters_oracle": int(its2o),
                              "pcg_iters_identical": bool(len(its2) == its2o and info2 == 0 and info2o == 0),
                              "map_rel_l2": rl2(x2.cpu().numpy(), x2o),
                              "map_rel_l2_vs_oracle_block_diagonal_solution": rl2(x2o, xo)}
                        if args.parity_arnoldi:
                            # the reference's OWN Arnoldi recurrence (modified Gram-Schmidt on Mbd*A, Mbd*b,
                            # src/test_M2_precond_onto_real_data.py:42, deflationlib.py:17-113) on the host for
                            # the same number of steps, Ritz pairs of its Hessenberg matrix (build_hess,
                            # la.eigh(H) as :43-46 does: the lower triangle), smallest r: the angles between
                            # that space and the GPU's Z say how far the two Krylov recurrences agree on the
                            # deflation space (same Krylov space; Euclidean against M^-1-orthogonal Ritz
                            # projection, so only converged Ritz vectors coincide)
                            t2 = time.perf_counter()
                            m_st = args.arnoldi_steps
                            vs, hs, m_o = orc.arnoldi(lambda v: H.M(H.A(v)), H.M(bo), np.zeros(bo.shape[0]),
                                                      tol=0.0, inner_m=m_st, exhausted="return")
                            Hm = orc.build_hess(hs, m_o)
                            th_o, U_o = np.linalg.eigh(Hm)          # (as the reference: la.eigh(H) reads the lower triangle)
                            V = np.column_stack(vs[:m_o])
                            Zo = V.dot(U_o[:, np.argsort(th_o)[:r]])
                            Qg, _ = np.linalg.qr(Zh)
                            Qo, _ = np.linalg.qr(Zo)
                            cosines = np.clip(np.linalg.svd(Qg.T.dot(Qo), compute_uv=False), 0.0, 1.0)
                            ang = np.sort(np.arccos(cosines))
                            tl["oracle_arnoldi"] = {
                                "steps": int(m_o), "seconds": round(time.perf_counter() - t2, 1),
                                "ritz_smallest_oracle": [float(v) for v in np.sort(th_o)[:4]],
                                "ritz_smallest_gpu": [float(v) for v in np.sort(theta)[:4]],
                                "principal_angles_rad_min_median_max": [float(ang[0]), float(np.median(ang)),
                                                                        float(ang[-1])],
                                "angles_below_1e-6": int(np.sum(ang < 1e-6)),
                                "angles_below_1e-3": int(np.sum(ang < 1e-3))}
                            del vs, V, Zo, Qg, Qo
                        tl["seconds"] = round(time.perf_counter() - t1, 1)
                        parity["two_level"] = tl
                        del Z, AZ, E, M2, x2, Zh, AZo, Eo, M2o, x2o
                    else:
                        parity["two_level"] = "skipped (host build + solve would take %.0f s)" % need
                del b, xo, bo
            else:
                parity["pcg"] = "skipped (host solve would take %.0f s; --parity-host-seconds %g)" % (
                    t_mv * ((pcg or {}).get("iters", 10) + 3), budget)
        parity["seconds"] = round(time.perf_counter() - tq, 1)
        beat.__exit__()
        del H, pix_h, d_h

    # device memory at the end of the run: what the library's objects hold, what it keeps cached, and
    # torch's peak (the vectors of the solves and the TOD-sized scratch of the operator)
    mem = (ctypes.c_int64 * 4)()
    _hip.call("cm2_device_memory_info", mem)
    memory = {"library_live_GB": round(mem[0] / 1e9, 3), "library_cached_GB": round(mem[1] / 1e9, 3),
              "torch_peak_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 3),
              "computed_budget_per_rank_strong_scaling_GB": strong_budget()}
    if rank == 0:
        out = {
            "metric": "TOD samples/s through P^T N^-1 P",
            "value": value, "unit": "TOD samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["label"], "nside": cfg["nside"], "pol": pol,
                       "nt_per_gpu": nt, "nt_all_gpus": nt_all, "npix": int(npix_c),
                       "noise": ("toeplitz" if lam else "diag"),
                       "lambda": lam, "blocks_per_gpu": nb, "fft_len": fft_len,
                       "tile_plan": tile_info,
                       "layout": args.layout if world > 1 else None,
                       "parallelism": ("tod-shard x%d + map all-gather / reduce-scatter (row-sharded vectors)" % world
                                       if rows else "tod-shard x%d + map all-reduce" % world)},
            "roofline": roofline,
            "step_algorithmic_GBps": round(step_gbs, 1),
            "step_frac_of_hbm_peak": round(step_gbs / HBM_PEAK_GBS, 4),
            "stages": stage_report, "stage_timing": timing,
            "distributed": dist_info,
            "other_scaling_point": other,
            "pcg": pcg,
            "raster_pointing": raster, "uneven_hit_map": uneven,
            "filters": filters,
            "parity_full_size": parity,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "hbm_memory": memory,
            "device_state": device_state(),
            "library": library_stamp(),
            "setup_seconds": round(t_setup, 2),
            "setup_split_seconds": {k: round(v, 3) for k, v in S_setup.items()},
            "commit": git_head(),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
