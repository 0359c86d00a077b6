"""
Round-5 additions, on the GPU: a REAL out-of-device-memory failure of a library call (status code,
retry after the host framework's idle memory went back to the driver, no stale runtime error afterwards),
persistent exchange buffers of the row-sharded layout, the sub-scan and ground filters at BASELINE size
against the oracle.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cm():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import cosmomap2_amd.interfaces as I
    import cosmomap2_amd.utilities as U
    import cosmomap2_amd
    from types import SimpleNamespace
    return SimpleNamespace(I=I, U=U, cg=cosmomap2_amd.cg, torch=torch)


def test_real_out_of_memory_is_reported_retried_and_leaves_no_stale_error(cm, monkeypatch):
    """cm2_core.hip dev_malloc_bytes / CM2_HIP / _hip.call on a real allocation failure (the CPU suite only has
    a fake): with the device filled by the host framework, cm2_weights_accumulate (process_ces.py:426-555;
    its pixel-sorted index needs ~20 B per sample) fails with CM2_ERR_OUT_OF_MEMORY and the library's message;
    the NEXT, unrelated call succeeds (the runtime's sticky last error was cleared: a stale
    hipErrorOutOfMemory would fail its CM2_LAUNCH_OK); with the same memory merely idle in torch's caching
    allocator the call fails once inside _hip.call, torch's cache is emptied and the second attempt gives
    the bits of the undisturbed run."""
    from cosmomap2_amd import _hip, device as D
    t = cm.torch
    dev = t.device("cuda", 0)
    nt, npix = 1 << 28, 1 << 20
    pix = t.randint(0, npix, (nt,), generator=t.Generator(device=dev).manual_seed(5), device=dev, dtype=t.int32)
    counts = D.empty(npix)

    def run():
        _hip.call("cm2_weights_accumulate", 1, nt, npix, D.ptr(pix), None, None, None, D.ptr(counts),
                  None, None, None, None, None, D.stream())
        t.cuda.synchronize()
    run()
    ref = counts.clone()
    assert float(ref.sum()) == float(nt)
    D.release_cached_memory()
    t.cuda.empty_cache()
    free, total = t.cuda.mem_get_info()
    keep_free = 1 << 30                                   # the call needs > 4 GB
    hog = t.empty(free - keep_free, dtype=t.uint8, device=dev)
    attempts = []
    real_free = _hip._free_torch_cache
    monkeypatch.setattr(_hip, "_free_torch_cache", lambda: attempts.append(1) or real_free())
    with pytest.raises(_hip.HipError, match="out of memory"):
        run()
    assert attempts == [1]                                # tried again once (nothing idle to free), then raised
    # no stale error: an unrelated launch-checked call works, and gives the right number
    assert D.dot(ref, ref) == float((ref * ref).sum())
    # the same memory idle in torch's cache: first attempt fails, cache emptied, second succeeds
    del hog
    assert t.cuda.mem_get_info()[0] < 2 * keep_free      # (still with torch, not with the driver)
    counts.zero_()
    run()
    assert attempts == [1, 1]
    assert t.equal(counts, ref)
    assert D.dot(ref, ref) == float((ref * ref).sum())


def test_filters_at_baseline_size_equal_oracle(cm, oracle):
    """SURVEY 8(f) rows 1 and 2 at C4's TOD size (1e8 samples, nside 256 IQU, 100 detector blocks; the
    sub-scan layout of bench.py's `filters` leg: 490 sub-scans of 2000 samples + 40 turnaround samples
    per block, 5 % of the samples flagged): FilterLO poly_order 0 and 2 (interfaces/linearoperators.py:
    94-283) and GroundFilterLO (:24-61) against the oracle on the host -- one application 1e-12 / 1e-11
    relative l2, one P^T F P matvec on the tile order 1e-12, and for P^T F0 P and P^T G P the PCG solve
    with M_BD: IDENTICAL iteration count, map within 1e-6 (north_star).  The oracle's P / P^T are its
    all-cores loops (checked against the serial ones in tests/test_oracle_golden.py), its filters the
    serial restatements (orc_filter_mean, filter_poly, ground_filter)."""
    import os
    import bench
    t = cm.torch
    cfg = bench.CONFIGS["c4"]
    pol, nside, nt, nb = 3, cfg["nside"], cfg["nt"], cfg["nb"]
    npix, bsize = 12 * nside * nside, cfg["nt"] // cfg["nb"]
    dev = t.device("cuda", 0)
    inp = bench.synth_inputs(t, dev, npix, nt, nb, 0, rank=0)
    pix, phi, d = inp["pix"], inp.pop("phi"), inp["d"]
    pix[t.rand(nt, generator=inp["gen"], device=dev) < 0.05] = -1
    pix_h, phi_h, d_h = pix.cpu().numpy(), phi.cpu().numpy(), d.cpu().numpy()
    sub, gap = 2000, 40
    starts = np.arange(0, bsize - sub + 1, sub + gap)
    sizes = np.full(starts.size, sub)
    # ---- product path ----
    ces = cm.U.ProcessTimeSamples(pix, npix, pol=pol, phi=phi)
    del phi
    n = ces.get_new_pixel[0]
    P = cm.I.SparseLO(n, nt, pix, pol=pol, angle_processed=ces)
    M = cm.I.BlockDiagonalPreconditionerLO(ces, n, pol=pol)
    # ---- oracle ----
    ro = oracle.process_time_samples(pix_h, npix, pol=pol, phi=phi_h)
    del phi_h
    assert ro.new_npix == n
    np.testing.assert_array_equal(pix.cpu().numpy(), pix_h)
    threads = len(os.sched_getaffinity(0))
    PP = oracle.AllCoresMatvec(pol, n, pix_h, ro.cos, ro.sin, bsize, None, threads)
    x = t.rand(pol * n, generator=t.Generator(device=dev).manual_seed(7), device=dev, dtype=t.float64)
    x_h = x.cpu().numpy()
    tod_o = PP.P(x_h)
    az = ((t.arange(nt, device=dev, dtype=t.int64) % (2 * (sub + gap))) - (sub + gap)).abs().to(t.int32)
    az_h = az.cpu().numpy()
    host_filters = {
        "F0": lambda v: oracle.filter_mean(v, pix_h, [sizes, starts], bsize, nb),
        "F2": lambda v: oracle.filter_poly(v, pix_h, [sizes, starts], bsize, nb, 2),
        "G": lambda v: oracle.ground_filter(az_h, v)}
    report = {}
    for name in ("F0", "F2", "G"):
        F = (cm.I.GroundFilterLO(az) if name == "G" else
             cm.I.FilterLO(nt, [sizes, starts], bsize, nb, pix, poly_order=(0 if name == "F0" else 2)))
        Fo = host_filters[name]
        fd_o = Fo(d_h)
        e_f = rel_l2((F * d).cpu().numpy(), fd_o)
        assert e_f < (1e-11 if name == "F2" else 1e-12), (name, e_f)
        A = P.T * F * P
        assert [type(op).__name__ for op in A._compiled()] == ["_TiledNormalLO"], name     # the tile-order chain
        y_o = PP.Pt(Fo(tod_o))
        e_a = rel_l2((A * x).cpu().numpy(), y_o)
        assert e_a < (1e-11 if name == "F2" else 1e-12), (name, e_a)
        report[name] = (e_f, e_a)
        if name == "F2":
            continue                                     # (a host application of the order-2 filter takes ~10 s)
        b = P.T * (F * d)
        b_o = PP.Pt(fd_o)
        assert rel_l2(b.cpu().numpy(), b_o) < 1e-12, name
        its, its_o = [], []
        xs, info = cm.cg(A, b, M=M, rtol=1e-6, maxiter=100, callback=lambda v: its.append(1))
        xo, info_o = oracle.cg(lambda v: PP.Pt(Fo(PP.P(v))), b_o, M=lambda v: oracle.bd_precond_mult(pol, ro, v),
                               rtol=1e-6, maxiter=100, callback=lambda v: its_o.append(1))
        assert info == 0 and info_o == 0, (name, info, info_o)
        assert len(its) == len(its_o), (name, len(its), len(its_o))             # identical, strictly
        e_x = rel_l2(xs.cpu().numpy(), xo)
        assert e_x < 1e-6, (name, e_x)
        report[name] += (len(its), e_x)
        del F, A, b
    print("filters at 1e8 samples vs oracle: " + "; ".join("%s %s" % (k, ["%.1e" % v if isinstance(v, float) else v
                                                                      for v in r]) for k, r in report.items()))
