#!/usr/bin/env python3
"""
bench.py -- TOD samples/s through one P^T N^-1 P matvec (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4]

A "step" is one application of A = P^T N^-1 P to a map-domain vector with all inputs
already resident in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU; each
rank owns a block-aligned TOD shard of the same size (weak scaling), a step is the local
matvec followed by the RCCL all-reduce of the map.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs; synthetic inputs after utilities_functions.py:99-212:
uniform-random pixel per sample, HWP angle ramp, d ~ U[0,1)):
  c2  nside 128 IQU, 1e7 samples, diagonal N (100 blocks)        -> fused single kernel
  c3  nside 128 IQU, 1e8 samples, banded-Toeplitz N, lambda 2048 -> P, overlap-save FFT, P^T
  c4  nside 256 IQU, 1e8 samples/GPU, Toeplitz lambda 2048       -> the configuration the
      north_star target (>= 40 % HBM roofline) is quoted on; DEFAULT.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    "c2": dict(nside=128, nt=10_000_000, nb=100, lam=0, label="C2 nside128 IQU 1e7 diag-N"),
    "c3": dict(nside=128, nt=100_000_000, nb=100, lam=2048,
               label="C3 nside128 IQU 1e8 Toeplitz(2048)"),
    "c4": dict(nside=256, nt=100_000_000, nb=100, lam=2048,
               label="C4 nside256 IQU 1e8/GPU Toeplitz(2048)"),
    # one GPU's share of C5 (1e9 samples, 64 detector blocks over 8 GPUs): 8 blocks of 15 625 000
    "c5": dict(nside=512, nt=125_000_000, nb=8, lam=2048,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("CM2_BENCH_CONFIG", "c4"),
                    choices=sorted(CONFIGS))
    ap.add_argument("--nt", type=int, default=0, help="override samples per GPU")
    ap.add_argument("--lam", type=int, default=0, help="override the Toeplitz band length")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pcg", action="store_true", help="skip the PCG iteration count")
    ap.add_argument("--no-raster", action="store_true",
                    help="skip the secondary run with a coherent raster-scan pointing")
    ap.add_argument("--no-filters", action="store_true",
                    help="skip the FilterLO / GroundFilterLO timing (SURVEY 8f rows)")
    ap.add_argument("--deflation", type=int, default=32,
                    help="rank of the deflation space of the two-level PCG run (0 = skip)")
    ap.add_argument("--arnoldi-steps", type=int, default=96)
    ap.add_argument("--fft-len", type=int, default=0)
    ap.add_argument("--toeplitz", default="fused", choices=["fused", "rocfft"],
                    help="overlap-save implementation for the Toeplitz configs")
    args = ap.parse_args()
    if args.fft_len:
        os.environ["CM2_FFT_LEN"] = str(args.fft_len)

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
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import cosmomap2_amd
    from cosmomap2_amd import device as D
    from cosmomap2_amd import _hip
    from cosmomap2_amd.interfaces import SparseLO, BlockLO, BlockDiagonalPreconditionerLO
    from cosmomap2_amd.utilities import ProcessTimeSamples
    from cosmomap2_amd.sharding import ShardedLO, make_sync

    cfg = dict(CONFIGS[args.config])
    if args.nt:
        cfg["nt"] = args.nt
    if args.lam and cfg["lam"]:
        cfg["lam"] = args.lam
        cfg["label"] += " [lambda=%d]" % args.lam
    pol = 3
    npix = 12 * cfg["nside"] ** 2
    nb = cfg["nb"]
    nt = (cfg["nt"] // nb) * nb
    bsize = nt // nb
    lam = cfg["lam"]
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---- synthetic shard, generated in HBM (seed differs per rank) -------------------
    t_setup = time.time()
    gen = torch.Generator(device=dev)
    gen.manual_seed(20161202 + 1000 * rank)
    rng = np.random.default_rng(20161202 + 1000 * rank)
    pix = torch.randint(0, npix, (nt,), generator=gen, device=dev, dtype=torch.int32)
    theta0 = float(rng.uniform(0, np.pi))
    phi = theta0 + (2 * np.pi * 2.5 / 200.0) * torch.arange(nt, device=dev, dtype=torch.float64)
    d = torch.rand(nt, generator=gen, device=dev, dtype=torch.float64)
    if lam:
        bands = [toeplitz_band(lam, rng) for _ in range(nb)]
        N = BlockLO(bsize, bands, offdiag=True, method=(3 if args.toeplitz == "fused" else 2))
        nlabel = "N^-1 (k_overlap_save, LDS FFT)" if args.toeplitz == "fused" else "N^-1 (overlap-save rocFFT)"
        w = None
    else:
        N = BlockLO(bsize, list(rng.random(nb) + 0.5), offdiag=False)
        w = N._device_diag()
    allred = None
    if world > 1:
        allred = lambda t: dist.all_reduce(t)
    ces = ProcessTimeSamples(pix, npix, pol=pol, phi=phi, w=w, allreduce=allred)
    del phi
    npix_c = ces.get_new_pixel[0]
    P = SparseLO(npix_c, nt, pix, pol=pol, angle_processed=ces)
    Mbd = BlockDiagonalPreconditionerLO(ces, npix_c, pol=pol)
    A_local = P.T * N * P
    A = ShardedLO(A_local) if world > 1 else A_local
    n = pol * npix_c
    x = torch.rand(n, generator=torch.Generator(device=dev).manual_seed(7), device=dev,
                   dtype=torch.float64)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: W warmup + exactly K steps ------------------------------------
    for _ in range(args.warmup):
        y = A * x
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = A * x
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * nt / (elapsed / args.steps)

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

    reps = max(5, min(args.steps, 20))
    stages = {}
    map_bytes = 48.0 * npix_c
    if lam:
        from cosmomap2_amd.interfaces import linearoperators as L
        tod = P * x
        tod2 = N * tod
        if L._use_tiles(P):
            T = L._sparse_tiles(P)
            st = D.stream
            d_tb = D.empty(T.nvalid)
            out = D.empty(n)
            call = _hip.call
            stages["P tiles (k_P_tiles)"] = (ev_time(lambda: call(
                "cm2_P_tiles_apply", T.h, D.ptr(x), D.ptr(d_tb), st()), reps),
                28.0 * nt + map_bytes / 2)
            if args.toeplitz == "fused":
                v_tb = D.empty(T.nvalid)
                stages["N^-1 on tile order (k_overlap_save_reg, register+LDS FFT)"] = (ev_time(lambda: call(
                    "cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(d_tb), D.ptr(v_tb), st()),
                    reps), 16.0 * nt)
                del v_tb
            else:
                stages["tiles->time (k_tiles_to_time)"] = (ev_time(lambda: call(
                    "cm2_tod_tiles_to_time", T.h, D.ptr(d_tb), D.ptr(tod), st()), reps), 16.0 * nt)
                stages[nlabel] = (ev_time(lambda: N * tod, reps), 16.0 * nt)
                stages["time->tiles (k_time_to_tiles)"] = (ev_time(lambda: call(
                    "cm2_tod_time_to_tiles", T.h, D.ptr(tod2), D.ptr(d_tb), st()), reps),
                    16.0 * nt)
            stages["P^T tiles (k_Pt_tiles)"] = (ev_time(lambda: call(
                "cm2_Pt_tiles_apply", T.h, D.ptr(d_tb), D.ptr(out), st()), reps),
                28.0 * nt + map_bytes / 2)
            del d_tb, out
        else:
            stages["P (k_P_time)"] = (ev_time(lambda: P * x, reps), 28.0 * nt + map_bytes / 2)
            stages[nlabel] = (ev_time(lambda: N * tod, reps), 16.0 * nt)
            stages["P^T (k_Pt_sell)"] = (ev_time(lambda: P.T * tod2, reps),
                                         28.0 * nt + map_bytes / 2)
        step_bytes = 72.0 * nt + map_bytes
        del tod, tod2
    else:
        stages["P^T diag(w) P fused (k_PtNP_sell)"] = (ev_time(lambda: A_local * x, reps),
                                                       28.0 * nt + map_bytes)
        step_bytes = 28.0 * nt + map_bytes
    dom = max(stages, key=lambda k: stages[k][0][0])
    (dom_mean, dom_med), dom_bytes = stages[dom]
    achieved = dom_bytes / (dom_mean * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "algorithmic_bytes": dom_bytes,
                "avg_launch_ms": round(dom_mean, 4)}
    # HBM traffic of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE need
    # rocprofv3 passes of their own, so they are taken from the committed summary of this very
    # command -- profiles/make_summary.py -- and only when the workload is the same)
    try:
        pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_c4.json"))
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
        if pmc["workload"] == cfg["label"] and pmc["nt_per_gpu"] == nt:
            for kname, rec in pmc["kernels"].items():
                if kname in dom and rec.get("hbm_traffic_bytes"):
                    roofline["traffic"] = rec["hbm_traffic_bytes"]
                    roofline["traffic_source"] = "profiles/" + pmcs[-1]
    except Exception:
        pass
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    stage_report = {k: {"ms": round(v[0][0], 4), "GB/s": round(v[1] / (v[0][0] * 1e-3) / 1e9, 1)}
                    for k, v in stages.items()}

    # ---- PCG iterations to 1e-6 (outside the timed region) ----------------------------
    pcg = None
    if not args.no_pcg:
        b = P.T * (N * d)
        if world > 1:
            dist.all_reduce(b)
        cosmomap2_amd.cg(A, b, M=Mbd, rtol=1e-6, maxiter=1, sync=make_sync())       # warm-up
        torch.cuda.synchronize()
        its = []
        tp = time.perf_counter()
        xs, info = cosmomap2_amd.cg(A, b, M=Mbd, rtol=1e-6, maxiter=500,
                                    callback=lambda xk: its.append(1), sync=make_sync())
        torch.cuda.synchronize()
        pcg = {"rtol": 1e-6, "iters": len(its), "info": int(info),
               "seconds": round(time.perf_counter() - tp, 3), "preconditioner": "block-diagonal"}
        if lam and args.deflation > 0:
            # two-level preconditioner with an Arnoldi/Ritz deflation space (BASELINE config C4)
            from cosmomap2_amd.interfaces import (DeflationLO, CoarseLO, TwoLevelPreconditionerLO,
                                                  ritz_deflation_basis)
            r = args.deflation
            tz = time.perf_counter()
            Z, theta = ritz_deflation_basis(A, Mbd, b, r, args.arnoldi_steps)
            AZ = torch.empty_like(Z)
            for j in range(r):
                AZ[:, j] = A * Z[:, j].contiguous()
            Zd, AZd = DeflationLO(Z), DeflationLO(AZ)
            E = CoarseLO(Z, AZ, r, apply='eig')
            M2 = TwoLevelPreconditionerLO(Mbd, Zd, AZd, E)
            torch.cuda.synchronize()
            t_build = time.perf_counter() - tz
            cosmomap2_amd.cg(A, b, M=M2, rtol=1e-6, maxiter=1, sync=make_sync())   # warm-up
            torch.cuda.synchronize()
            its2 = []
            tp = time.perf_counter()
            xs2, info2 = cosmomap2_amd.cg(A, b, M=M2, rtol=1e-6, maxiter=500,
                                          callback=lambda xk: its2.append(1), sync=make_sync())
            torch.cuda.synchronize()
            t_pcg2 = time.perf_counter() - tp
            rel = float(torch.linalg.vector_norm(xs2 - xs) / torch.linalg.vector_norm(xs))
            pcg["two_level"] = {"rank": r, "arnoldi_steps": args.arnoldi_steps,
                                "iters": len(its2), "info": int(info2),
                                "seconds": round(t_pcg2, 4),
                                "build_seconds": round(t_build, 3),
                                "smallest_ritz": float(theta[0]), "largest_kept_ritz": float(theta[-1]),
                                "rel_l2_vs_block_diagonal_solution": rel}

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
    if rank == 0 and world == 1 and lam and not args.no_raster and npix % 1024 == 0:
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

    if rank == 0:
        out = {
            "metric": "TOD samples/s through P^T N^-1 P",
            "value": value, "unit": "TOD samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["label"], "nside": cfg["nside"], "pol": pol,
                       "nt_per_gpu": nt, "npix": int(npix_c), "noise": ("toeplitz" if lam else "diag"),
                       "lambda": lam, "blocks_per_gpu": nb,
                       "fft_len": (N.noise_info()["fft_len"] if lam else 0),
                       "parallelism": "tod-shard x%d + map all-reduce" % world},
            "roofline": roofline,
            "step_algorithmic_GBps": round(step_gbs, 1),
            "step_frac_of_hbm_peak": round(step_gbs / HBM_PEAK_GBS, 4),
            "stages": stage_report,
            "pcg": pcg,
            "raster_pointing": raster,
            "filters": filters,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "setup_seconds": round(t_setup, 2),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
