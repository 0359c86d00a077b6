"""
Round-4 additions to the C ABI, on the GPU: cm2_noise_prepare_tiles (the overlap-save lists of an
(operator, tile plan) pair built ahead of the first application, kept per plan, safe to share between
host threads and inside a stream capture), cm2_set_exact_order, cm2_release_cached_memory.
"""
import threading

import numpy as np
import pytest

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


def _plan(cm, seed, nt=300000, npix=6000, tp=512, pol=3):
    from types import SimpleNamespace
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(seed)
    pairs = rng.integers(0, npix, nt)
    pairs[rng.random(nt) < 0.03] = -1
    phi = 0.3 + 0.0785 * np.arange(nt)
    P = cm.I.SparseLO(npix, nt, pairs, pol=pol,
                      angle_processed=SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi)))
    return P, L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)


def _noise(cm, nt, nblk=4, lam=60):
    kk = np.arange(lam)
    return cm.I.BlockLO(nt // nblk, [(1.0 + 0.1 * b) * np.exp(-kk / 9.0) for b in range(nblk)],
                        offdiag=True, method=3)


def test_prepared_chain_is_captured_into_a_graph(cm):
    """After cm2_tiles_prepare_pt and cm2_noise_prepare_tiles the whole chain P^T N^-1 P
    (cm2_PtNP_tiles_apply: three launches) allocates nothing and never synchronises: it can be
    recorded into a HIP graph, and the replayed graph gives the eager bits."""
    from cosmomap2_amd import _hip, device as D
    t = cm.torch
    P, T = _plan(cm, 1)
    N = _noise(cm, P.nrows)
    pol, n = P.pol, P.pol * P.ncols
    x = D.f64(np.random.default_rng(2).standard_normal(n))
    y, w1, w2 = D.empty(n), D.empty(T.nvalid), D.empty(T.nvalid)
    _hip.call("cm2_noise_prepare_tiles", N._noise.h, T.h, D.stream())
    assert N.tile_kernel_info()["os_lists"] != "not built"
    side = t.cuda.Stream()
    side.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(side):
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g, stream=side):
            _hip.call("cm2_PtNP_tiles_apply", T.h, N._noise.h, D.ptr(x), D.ptr(y), D.ptr(w1), D.ptr(w2),
                      D.stream())
        y.fill_(7.0)
        g.replay()
    side.synchronize()
    t.cuda.current_stream().wait_stream(side)
    got = y.clone()
    _hip.call("cm2_PtNP_tiles_apply", T.h, N._noise.h, D.ptr(x), D.ptr(y), D.ptr(w1), D.ptr(w2), D.stream())
    assert t.equal(got, y)
    x.mul_(-0.5)                                   # the graph reads the same buffers again
    with t.cuda.stream(side):
        g.replay()
    side.synchronize()
    got = y.clone()
    _hip.call("cm2_PtNP_tiles_apply", T.h, N._noise.h, D.ptr(x), D.ptr(y), D.ptr(w1), D.ptr(w2), D.stream())
    assert t.equal(got, y)


def test_one_noise_operator_on_several_plans_from_two_threads(cm):
    """One noise operator shared by tile plans: it keeps the lists of its most recently used plans
    (alternating between two plans builds each list set once), and two host threads that apply it
    to two plans at the same time get the results of the sequential run."""
    from cosmomap2_amd import _hip, device as D
    t = cm.torch
    nt = 300000
    plans = [_plan(cm, s, nt=nt, tp=tp) for s, tp in ((11, 512), (12, 256), (13, 1024), (14, 512))]
    N = _noise(cm, nt)
    tod = D.f64(np.random.default_rng(3).standard_normal(nt))
    ins, want = [], []
    for P, T in plans:
        a, b = D.empty(T.nvalid), D.empty(T.nvalid)
        _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(tod), D.ptr(a), D.stream())
        _hip.call("cm2_noise_apply_tiles", N._noise.h, T.h, D.ptr(a), D.ptr(b), D.stream())
        ins.append(a)
        want.append(b.clone())
    t.cuda.synchronize()
    mem = (__import__("ctypes").c_int64 * 4)()
    _hip.call("cm2_device_memory_info", mem)
    misses0 = int(mem[3])
    for _ in range(3):                              # plans 2 and 3 alternate: their lists are cached
        for k in (2, 3):
            b = D.empty(plans[k][1].nvalid)
            _hip.call("cm2_noise_apply_tiles", N._noise.h, plans[k][1].h, D.ptr(ins[k]), D.ptr(b), D.stream())
            assert t.equal(b, want[k])
    _hip.call("cm2_device_memory_info", mem)
    assert int(mem[3]) == misses0                   # no list was rebuilt (no new driver allocation)
    # the oldest plan (0) fell out of the cache of three: applying it again rebuilds, same result
    b = D.empty(plans[0][1].nvalid)
    _hip.call("cm2_noise_apply_tiles", N._noise.h, plans[0][1].h, D.ptr(ins[0]), D.ptr(b), D.stream())
    assert t.equal(b, want[0])
    # two threads, two plans, one operator, each on its own stream; a fresh operator so that both
    # threads meet in the list build
    N2 = _noise(cm, nt)
    outs, errs = {}, []

    def work(k):
        try:
            st = t.cuda.Stream()
            with t.cuda.stream(st):
                for _ in range(5):
                    b = D.empty(plans[k][1].nvalid)
                    _hip.call("cm2_noise_apply_tiles", N2._noise.h, plans[k][1].h, D.ptr(ins[k]), D.ptr(b),
                              D.stream())
                st.synchronize()
                outs[k] = b
        except Exception as e:                      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not errs, errs
    for k in (0, 1):
        assert t.equal(outs[k], want[k])


def test_set_exact_order_switches_the_hot_pixel_sums(cm, oracle):
    """cm2_set_exact_order(1): every per-pixel sum in the reference's serial order (bit-equal to the
    oracle also for a pixel with 20 000 hits, which the default sums in fixed chunks)."""
    from cosmomap2_amd import _hip
    rng = np.random.default_rng(5)
    nt, npix, pol = 60000, 50, 3
    pairs = rng.integers(0, npix, nt).astype(np.int32)
    pairs[rng.random(nt) < 0.35] = 7
    phi = 0.1 + 0.0785 * np.arange(nt)
    w = rng.random(nt)
    ro = oracle.process_time_samples(pairs.copy(), npix, pol=pol, phi=phi, w=w)
    try:
        _hip.call("cm2_set_exact_order", 1)
        rg = cm.U.ProcessTimeSamples(pairs.copy(), npix, pol=pol, phi=phi, w=w)
        for k in ("counts", "cosine", "sine", "cos2", "sin2", "sincos"):
            np.testing.assert_array_equal(getattr(rg, k), getattr(ro, k), err_msg=k)
        _hip.call("cm2_set_exact_order", 0)
        rd = cm.U.ProcessTimeSamples(pairs.copy(), npix, pol=pol, phi=phi, w=w)
        for k in ("counts", "cosine", "sine", "cos2", "sin2", "sincos"):
            np.testing.assert_allclose(getattr(rd, k), getattr(ro, k), rtol=1e-13, err_msg=k)
        assert any(not np.array_equal(getattr(rd, k), getattr(ro, k)) for k in ("cosine", "sine", "cos2"))
    finally:
        _hip.call("cm2_set_exact_order", -1)


def test_release_cached_memory_returns_everything(cm):
    from cosmomap2_amd import _hip
    import ctypes
    P, T = _plan(cm, 21)
    del P, T
    cm.torch.cuda.synchronize()
    _hip.call("cm2_release_cached_memory")
    mem = (ctypes.c_int64 * 4)()
    _hip.call("cm2_device_memory_info", mem)
    assert int(mem[1]) == 0


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("parts,pol,angles,hot_pixel", [("6000", 3, "full", False), ("20000", 3, "half", False),
                                                      ("6000", 1, "half", False), ("9000", 2, "full", False),
                                                      ("6000", 3, "full", True), (None, 3, "half", False),
                                                      ("6000", 1, "half", True), ("9000", 2, "half", True),
                                                      ("9000", 2, "full", True)])
def test_heavy_tiles_are_shared_out_to_several_workgroups(cm, oracle, monkeypatch, parts, pol, angles, hot_pixel):
    """Uneven hit map, uniform tiles, the slices of the heavy tiles summed by several workgroups and
    their tile copies added in time order (cm2_tiles.h "PARTS").  Against the oracle's serial loop
    (1e-13; the one-workgroup-per-tile plan gives its bits), reproducible from call to call, the same
    bits whether P^T is applied at once or group of tiles by group of tiles, and the exact order
    (cm2_tiles_set_pt_order(t, 2)) back to the serial bits.  A pixel with a large share of the samples
    becomes a tile of its own inside the uniform grid."""
    import ctypes
    from types import SimpleNamespace
    from cosmomap2_amd import _hip, device as D
    from cosmomap2_amd.interfaces import linearoperators as L
    monkeypatch.setenv("CM2_TILE_ANGLES", angles)
    if parts is None:
        monkeypatch.delenv("CM2_PT_PARTS", raising=False)
        nt, npix, tp = 1 << 23, 49152, 256           # large enough for the automatic part length
    else:
        monkeypatch.setenv("CM2_PT_PARTS", parts)
        nt, npix, tp = 1 << 21, 49152, 128
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, npix, nt).astype(np.int32)
    dense = rng.random(nt) < 0.5
    pairs[dense] = pairs[dense] % (npix // 10)
    if hot_pixel:
        pairs[rng.random(nt) < 0.1] = 777
    pairs[rng.random(nt) < 0.02] = -1
    phi = 0.3 + 0.0785 * np.arange(nt)
    c, s = np.cos(2 * phi), np.sin(2 * phi)
    v = rng.standard_normal(nt)
    want = oracle.sparse_rmult(pol, npix, pairs, c, s, v)
    st = D.stream()

    def plan():
        P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=SimpleNamespace(cos=c, sin=s))
        T = L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)
        v_tb = D.empty(T.nvalid)
        _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(D.f64(v)), D.ptr(v_tb), st)
        return P, T, v_tb

    def apply(T, v_tb):
        out = D.empty(pol * npix)
        out.fill_(np.nan)
        _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st)
        return out.cpu().numpy()

    monkeypatch.setenv("CM2_TILE_BALANCE", "0")
    P0, T0, v0 = plan()
    one_wg = apply(T0, v0)
    if angles == "full" and not hot_pixel:
        np.testing.assert_array_equal(one_wg, want)
    monkeypatch.delenv("CM2_TILE_BALANCE")
    P1, T1, v1 = plan()
    info = T1.pt_parts()
    nuniform = (npix + tp - 1) // tp
    assert T1.ntiles == nuniform + (2 if hot_pixel else 0), (T1.ntiles, nuniform)
    assert info["tiles_split"] >= 10 and info["workgroups"] >= T1.ntiles + info["tiles_split"], info
    assert info["copy_bytes"] > 0 and 0.9 < info["simulated_finish_over_ideal"] < 3.0, info
    got = apply(T1, v1)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-13 * scale
    if not hot_pixel:
        assert np.abs(got - one_wg).max() <= 1e-14 * scale      # (two orders of ~850 terms: rounding only)
    np.testing.assert_array_equal(apply(T1, v1), got)                       # reproducible
    # group of tiles by group of tiles (what a sharded run does): the same bits
    cuts = (ctypes.c_int64 * 5)()
    _hip.call("cm2_tiles_group_tiles", T1.h, 4, cuts)
    pieces = D.empty(pol * npix)
    pieces.fill_(np.nan)
    for g in (2, 0, 3, 1):
        _hip.call("cm2_Pt_tiles_apply_range", T1.h, D.ptr(v1), D.ptr(pieces), int(cuts[g]), int(cuts[g + 1]), st)
    np.testing.assert_array_equal(pieces.cpu().numpy(), got)
    # exact order: one workgroup per tile again, the serial bits
    T1.set_pt_order(2)
    T0.set_pt_order(2)
    np.testing.assert_array_equal(apply(T1, v1), apply(T0, v0))
    if angles == "full":
        np.testing.assert_array_equal(apply(T1, v1), want)
    T1.set_pt_order(1)
    np.testing.assert_array_equal(apply(T1, v1), got)
    # round 5: the copies of a split tile are added by the tile's last part to finish and the hot tile's ranges
    # are work items of the same launch (one launch; who adds depends on the dispatch, what is added and in
    # which order does not) -- the separate kernels of round 4 (CM2_PT_FUSE=0) give the same bits, and so do
    # many applications in a row (the arrival counters are zeroed in front of every launch)
    monkeypatch.setenv("CM2_PT_FUSE", "0")
    P2, T2, v2 = plan()
    np.testing.assert_array_equal(apply(T2, v2), got)
    for _ in range(20):
        _hip.call("cm2_Pt_tiles_apply", T1.h, D.ptr(v1), D.ptr(pieces), st)
    np.testing.assert_array_equal(pieces.cpu().numpy(), got)
