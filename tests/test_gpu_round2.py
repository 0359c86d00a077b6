"""
GPU tests added in round 2: the krypy wrappers, the Ritz-vector checkpoint
in a restarted solve (8f row 4), the reference's own HDF5 test inputs through the whole
pipeline (8f row 3), and operators that outlive the pointing they were first used with.
"""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cm():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import cosmomap2_amd.interfaces as I
    import cosmomap2_amd.utilities as U
    import cosmomap2_amd
    from types import SimpleNamespace
    return SimpleNamespace(I=I, U=U, cg=cosmomap2_amd.cg, torch=torch)


def _system(cm, oracle, seed, nt, npix, nb, pol, lam=12):
    rng = np.random.default_rng(seed)
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    k = np.arange(lam)
    bands = [(1.0 + 0.1 * b) * np.where(k == 0, 1.0, -0.3 * np.exp(-k / 4.0)) for b in range(nb)]
    N = cm.I.BlockLO(nt // nb, bands, offdiag=True)
    ces = cm.U.ProcessTimeSamples(pairs, npix, pol=pol, phi=phi)
    n = ces.get_new_pixel[0]
    P = cm.I.SparseLO(n, nt, pairs, pol=pol, angle_processed=ces)
    M = cm.I.BlockDiagonalPreconditionerLO(ces, n, pol=pol)
    A = P.T * N * P
    b = P.T * (N * d)
    return A, M, b, pol * n


# ------------------------------------------------------------------- a14 -----------
def test_run_krypy_arnoldi_relation_and_biorthogonality(cm, oracle):
    """run_krypy_arnoldi (deflationlib.py:187-202) on a map-making system with M = M_BD: the
    Arnoldi relation A V_k = P_{k+1} H_k with P = M^-1 V, bi-orthogonality V^T P = I (= the
    M^-1-orthonormality krypy documents), return shapes (V n x (m+1), H (m+1) x m, m = columns of V).
    PARITY UNPINNED against krypy itself (absent); these are the properties its docstring
    states."""
    A, M, b, n = _system(cm, oracle, 21, 30000, 200, 3, 3)
    m_it = 12
    V, H, m = cm.I.run_krypy_arnoldi(A, b, M, 1e-8, maxiter=m_it)
    assert V.shape == (n, m_it + 1) and H.shape == (m_it + 1, m_it) and m == m_it + 1
    Ad, Md = A.to_array(), M.to_array()
    Pm = np.linalg.solve(Md, V)                               # P = M^-1 V
    np.testing.assert_allclose(V.T.dot(Pm), np.eye(m_it + 1), atol=1e-10)
    np.testing.assert_allclose(Ad.dot(V[:, :m_it]), Pm.dot(H), atol=1e-10 * np.abs(H).max())
    assert np.allclose(np.tril(H, -2), 0.0)                   # upper Hessenberg
    assert np.all(np.diag(H, -1) > 0)
    # H = V^T A V is symmetric for symmetric A and M: tridiagonal to rounding
    Hs = H[:m_it, :m_it]
    assert np.abs(Hs - Hs.T).max() < 1e-9 * np.abs(Hs).max()
    # default maxiter = n (reference: nmax = N) on a tiny system stops when the space is exhausted
    rng = np.random.default_rng(2)
    Q = np.linalg.qr(rng.standard_normal((6, 6)))[0]
    As = (Q * np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])).dot(Q.T)
    Vs, Hs2, ms = cm.I.run_krypy_arnoldi(As, rng.standard_normal(6), None, 1e-8)
    assert Vs.shape[0] == 6 and ms == Vs.shape[1] and ms <= 7
    np.testing.assert_allclose(As.dot(Vs[:, :Hs2.shape[1]]), Vs[:, :Hs2.shape[0]].dot(Hs2), atol=1e-10)


def test_run_krypy_arnoldi_without_M_equals_reference_arnoldi(cm, golden):
    """With M = None the recurrence is the plain Arnoldi the reference implements itself
    (deflationlib.py:17-113), whose executed output is in the golden file: same V and H up to
    the sign of each vector."""
    G = golden
    A, b = G["arn_A"], G["arn_b"]
    j = int(G["arn_j"])
    V, H, m = cm.I.run_krypy_arnoldi(A, b, None, 1e-8, maxiter=j)
    Vg = np.asarray(G["arn_V"])                                # j vectors as rows
    k = min(j, V.shape[1])
    sign = np.sign(np.sum(V[:, :k] * Vg[:k].T, axis=0))
    np.testing.assert_allclose(V[:, :k] * sign, Vg[:k].T, atol=1e-8)
    Hg = G["arn_H"]
    np.testing.assert_allclose(np.abs(H[:k, :k]), np.abs(Hg[:k, :k]), atol=1e-8)


def test_find_ritz_eigenvalues_selection_shapes_and_checkpoint(cm, oracle, tmp_path):
    """find_ritz_eigenvalues (deflationlib.py:204-219): Ritz pairs ordered by residual norm,
    ``r`` = number below the threshold, ``z[:, :r]`` / the masked triple, and the HDF5 dump of
    ALL Ritz vectors when a filename is given."""
    A, M, b, n = _system(cm, oracle, 22, 30000, 200, 3, 3)
    V, H, m = cm.I.run_krypy_arnoldi(A, b, M, 1e-8, maxiter=20)
    Ad, Md = A.to_array(), M.to_array()
    thr = float(np.median(np.linalg.eigvalsh(0.5 * (H[:20, :20] + H[:20, :20].T))))
    Z, r = cm.I.find_ritz_eigenvalues(H, V, threshold=thr)
    assert Z.shape == (n, r) and 0 < r < 20
    Zs, rs, ev = cm.I.find_ritz_eigenvalues(H, V, threshold=thr, eigenvalues=True)
    assert rs == r and Zs.shape == (n, r) and ev.shape == (r,) and np.all(ev < thr)
    # Ritz pairs of M A: residual ||M A z - theta z|| of the selected pairs is bounded by
    # |h_{m+1,m}| |last component|, and the order is by that residual (krypy.utils.ritz)
    MA = Md.dot(Ad)
    theta_all = np.linalg.eigvalsh(0.5 * (H[:20, :20] + H[:20, :20].T))
    assert np.all(np.isin(np.round(ev, 10), np.round(theta_all, 10)))
    Zall, rall, evall = cm.I.find_ritz_eigenvalues(H, V, threshold=np.inf, eigenvalues=True)
    assert rall == 20
    res = np.linalg.norm(MA.dot(Zall) - Zall * evall, axis=0) / np.linalg.norm(Zall, axis=0)
    assert np.all(np.diff(res) >= -1e-9 * res.max()), "Ritz pairs are not ordered by residual"
    # the Ritz values interlace the spectrum of M A (Rayleigh-Ritz in the M^-1 inner product)
    lam = np.sort(np.linalg.eigvals(MA).real)
    assert lam[0] - 1e-9 <= evall.min() and evall.max() <= lam[-1] + 1e-9
    fn = str(tmp_path / "ritz.hdf5")
    cm.I.find_ritz_eigenvalues(H, V, threshold=thr, filename=fn)
    zf, nf, ef = cm.U.read_ritz_eigenvectors_from_hdf5(fn, eigvals=True)
    assert zf.shape == (n, 20) and int(nf) == 20
    np.testing.assert_array_equal(zf[:, :r], Z)
    np.testing.assert_array_equal(ef[:r], ev if np.all(ef[:r] < thr) else ef[:r])


# -------------------------------------------------------------- 8f row 4 -----------
def test_solve_restarted_from_ritz_checkpoint_is_bit_identical(cm, oracle, tmp_path):
    """Deflation basis written with write_ritz_eigenvectors and read back: Z bit for bit, and
    the two-level PCG restarted from the file reproduces iteration count and solution bit for
    bit."""
    A, M, b, n = _system(cm, oracle, 23, 60000, 400, 4, 3, lam=40)
    bd = cm.torch.from_numpy(b).cuda()
    r = 8
    Z, theta = cm.I.ritz_deflation_basis(A, M, bd, r, 30)
    fn = cm.U.write_ritz_eigenvectors(Z, str(tmp_path / "basis"), eigvals=theta)

    def solve(Zm):
        AZ = cm.I.apply_to_columns(A, Zm)
        E = cm.I.CoarseLO(Zm, AZ, r, apply='eig')
        M2 = cm.I.TwoLevelPreconditionerLO(M, cm.I.DeflationLO(Zm), cm.I.DeflationLO(AZ), E)
        its = []
        x, info = cm.cg(A, bd, M=M2, rtol=1e-8, maxiter=500, callback=lambda v: its.append(1))
        assert info == 0
        return x, len(its)
    x1, k1 = solve(Z)
    Z2, th2 = cm.U.read_ritz_eigenvectors(fn, eigvals=True, device=True)
    assert cm.torch.equal(Z2, Z) and np.array_equal(th2, theta)
    x2, k2 = solve(Z2)
    assert k2 == k1 and cm.torch.equal(x2, x1)
    # the column-wise reference construction of AZ gives the same matrix
    AZ = cm.I.apply_to_columns(A, Z)
    for j in (0, r - 1):
        assert cm.torch.equal(AZ[:, j].contiguous(), A * Z[:, j].contiguous())
    # ... and so does the Arnoldi relation A V_m = P_{m+1} H, to the rounding of the recurrence
    Zk, thk, AZk = cm.I.ritz_deflation_basis(A, M, bd, r, 30, with_AZ=True)
    assert cm.torch.equal(Zk, Z) and np.array_equal(thk, theta)
    assert float((AZk - AZ).norm() / AZ.norm()) < 1e-10


@pytest.mark.parametrize("n,rin,rout", [(1000, 32, 32), (4099, 32, 16), (517, 5, 3), (33, 32, 32)])
def test_panel_gemm(cm, n, rin, rout):
    """cm2_panel_gemm: out (+)= P W for a tall row-major panel (fp64 MFMA for 32-column panels)."""
    from cosmomap2_amd import _hip, device as D
    rng = np.random.default_rng(n)
    P, W, O = rng.standard_normal((n, rin)), rng.standard_normal((rin, rout)), rng.standard_normal((n, rout))
    dP, dW, dO = D.f64(P), D.f64(W), D.f64(O.copy())
    _hip.call("cm2_panel_gemm", n, rin, rout, D.ptr(dP), D.ptr(dW), D.ptr(dO), 0, D.stream())
    np.testing.assert_allclose(dO.cpu().numpy(), P.dot(W), rtol=1e-13, atol=1e-13)
    dO = D.f64(O.copy())
    _hip.call("cm2_panel_gemm", n, rin, rout, D.ptr(dP), D.ptr(dW), D.ptr(dO), 1, D.stream())
    np.testing.assert_allclose(dO.cpu().numpy(), O + P.dot(W), rtol=1e-13, atol=1e-13)


# -------------------------------------------------------------- 8f row 3 -----------
@pytest.mark.parametrize("case,pol", [(3, 1), (3, 3), (4, 1), (4, 3), (4, 2)])
def test_reference_hdf5_inputs_through_the_pipeline(cm, oracle, case, pol):
    """The reference's data/testcase_block_diag_{3,4}.hdf5 (pixel, pol_angle, sum, weight):
    file -> read_from_hdf5 -> ProcessTimeSamples -> SparseLO -> BlockLO(weight) -> PCG with M_BD,
    against the oracle on the same arrays.  Case 4 holds two diagonal weights (the fused
    kernel), case 3 two bands of two entries (Toeplitz blocks)."""
    d, pix, phi, weight = cm.U.read_from_hdf5(os.path.join(GOLD, "testcase_block_diag_%d.hdf5" % case))
    nt = d.size
    offdiag = weight.ndim == 2
    nb = weight.shape[0]
    t = [np.array([1.0 + w[0], 0.3 * w[1]]) for w in weight] if offdiag else list(weight)
    npix = int(pix.max()) + 1
    po = pix.copy()
    N = cm.I.BlockLO(nt // nb, t, offdiag=offdiag)
    wdiag = None if offdiag else N.diag
    ces = cm.U.ProcessTimeSamples(pix, npix, pol=pol, phi=phi, w=wdiag)
    ro = oracle.process_time_samples(po, npix, pol=pol, phi=phi, w=wdiag)
    n = ces.get_new_pixel[0]
    assert n == ro.new_npix and np.array_equal(pix, po)
    for k in {1: ("counts",), 2: ("cos2", "sin2", "sincos"), 3: ("counts", "cosine", "sine", "cos2", "sin2", "sincos")}[pol]:
        np.testing.assert_array_equal(getattr(ces, k), getattr(ro, k))
    P = cm.I.SparseLO(n, nt, pix, pol=pol, angle_processed=ces)
    M = cm.I.BlockDiagonalPreconditionerLO(ces, n, pol=pol)
    A = P.T * N * P
    c, s = ro.cos, ro.sin

    def A_o(x):
        return oracle.sparse_rmult(pol, n, po, c, s, oracle.blocklo_mult(
            nt // nb, t, offdiag, oracle.sparse_mult(pol, po, c, s, x)))
    x = np.random.default_rng(case).standard_normal(pol * n)
    np.testing.assert_array_equal(A * x, A_o(x))
    b = P.T * N * d
    b_o = oracle.sparse_rmult(pol, n, po, c, s, oracle.blocklo_mult(nt // nb, t, offdiag, d))
    np.testing.assert_array_equal(b, b_o)
    its_g, its_o = [], []
    xg, ig = cm.cg(A, b, M=M, rtol=1e-8, maxiter=500, callback=lambda v: its_g.append(1))
    xo, io = oracle.cg(A_o, b_o, M=lambda v: oracle.bd_precond_mult(pol, ro, v), rtol=1e-8,
                       maxiter=500, callback=lambda v: its_o.append(1))
    assert ig == 0 and io == 0 and len(its_g) == len(its_o)
    assert rel_l2(xg, xo) < 1e-9
    if not offdiag:
        assert len(its_g) == 1                                 # diagonal N: M_BD = A^-1


# ------------------------------------------- operators that outlive a pointing -----
def test_noise_and_filter_reused_across_pointings(cm, oracle):
    """One BlockLO(method=3) and one FilterLO applied on the tile order of two successive
    pointings of the same length but different flags: their address lists belong to the FIRST
    tile plan and must be rebuilt for the second (they were keyed on a device address that the
    allocator can hand out again)."""
    from cosmomap2_amd.interfaces import linearoperators as L
    pol, npix, nb, ns = 3, 900, 4, 40000
    nt = nb * ns
    rng = np.random.default_rng(31)
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    lam = 50
    k = np.arange(lam)
    bands = [(1.0 + 0.05 * b) * np.where(k == 0, 1.0, 0.2 * np.exp(-k / 12.0)) for b in range(nb)]
    N = cm.I.BlockLO(ns, bands, offdiag=True, method=3)
    Nd = cm.I.BlockLO(ns, bands, offdiag=True, method=1)
    sizes, starts = np.array([9000, 9500, 9000, 9800]), np.array([100, 9400, 19500, 29600])
    x = rng.standard_normal(pol * npix)
    L.set_pointing_mode("tiled")
    try:
        results = []
        for trial, frac in enumerate((0.02, 0.3)):
            p = pairs.copy()
            p[np.random.default_rng(trial).random(nt) < frac] = -1
            ang = type("Ang", (), {"cos": np.cos(2 * phi), "sin": np.sin(2 * phi)})()
            P = cm.I.SparseLO(npix, nt, p, pol=pol, angle_processed=ang)
            exact = oracle.sparse_rmult(pol, npix, p, ang.cos, ang.sin, oracle.blocklo_mult(
                ns, bands, True, oracle.sparse_mult(pol, p, ang.cos, ang.sin, x)))
            got = (P.T * N * P) * x
            assert rel_l2(got, exact) < 1e-12, (trial, rel_l2(got, exact))
            results.append(got)
            del P                                        # frees the plan; the next one may reuse its memory
        assert rel_l2(results[0], results[1]) > 1e-3      # the two pointings really differ
        # FilterLO: flags of its own; it may run on a tile order only if the flags agree, and
        # its window lists must follow the plan
        for trial, frac in enumerate((0.02, 0.3)):
            p = pairs.copy()
            p[np.random.default_rng(10 + trial).random(nt) < frac] = -1
            F = cm.I.FilterLO(nt, [sizes, starts], ns, nb, p, poly_order=1)
            for rep in range(2):
                ang = type("Ang", (), {"cos": np.cos(2 * phi), "sin": np.sin(2 * phi)})()
                P = cm.I.SparseLO(npix, nt, p, pol=pol, angle_processed=ang)
                exact = oracle.sparse_rmult(pol, npix, p, ang.cos, ang.sin, oracle.filter_poly(
                    oracle.sparse_mult(pol, p, ang.cos, ang.sin, x), p, [sizes, starts], ns, nb, 1))
                got = (P.T * F * P) * x
                assert rel_l2(got, exact) < 1e-11, (trial, rep, rel_l2(got, exact))
                del P
    finally:
        L.set_pointing_mode("auto")


def test_ritz_deflation_basis_with_AZ_on_exhausted_krylov_space(cm):
    """A Z from the Arnoldi relation when the Krylov space is exhausted before ``maxiter`` (the
    last basis vector is never stored and H's last sub-diagonal entry is ~0), with and without a
    preconditioner, against the column-by-column product."""
    from cosmomap2_amd import device as D
    rng = np.random.default_rng(5)
    n = 40
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    # 12 distinct eigenvalues: the Krylov space of any start vector has dimension <= 12
    lam = np.repeat(np.linspace(0.05, 3.0, 12), 4)[:n]
    As = (Q * lam).dot(Q.T)
    As = 0.5 * (As + As.T)
    b = rng.standard_normal(n)
    for Ms in (None, np.eye(n) * 0.7):
        Z, theta, AZ = cm.I.ritz_deflation_basis(As, Ms, b, 4, 30, with_AZ=True)
        Zh, AZh = D.to_host(Z), D.to_host(AZ)
        assert Zh.shape == (n, 4) and AZh.shape == (n, 4) and len(theta) == 4
        np.testing.assert_allclose(AZh, As.dot(Zh), atol=1e-9 * np.abs(AZh).max())
        # the Ritz pairs of an exhausted space are eigenpairs of M A
        MAZ = AZh if Ms is None else Ms.dot(AZh)
        assert np.abs(MAZ - Zh * np.asarray(theta)).max() < 1e-7 * np.abs(Zh).max()


@pytest.mark.parametrize("n,r", [(5000, 32), (4099, 16), (777, 64), (1000, 5), (3, 32)])
def test_Z_axpy(cm, n, r):
    """cm2_Z_axpy: w += alpha Z y in one pass over a row-major panel (wide loads for r = 16, 32,
    64; the Arnoldi orthogonalisation's update) against NumPy."""
    from cosmomap2_amd import _hip, device as D
    rng = np.random.default_rng(n + r)
    Z, y, w = rng.standard_normal((n, r)), rng.standard_normal(r), rng.standard_normal(n)
    dZ, dy, dw = D.f64(Z), D.f64(y), D.f64(w.copy())
    _hip.call("cm2_Z_axpy", n, r, D.ptr(dZ), D.ptr(dy), -0.75, D.ptr(dw), D.stream())
    np.testing.assert_allclose(dw.cpu().numpy(), w - 0.75 * Z.dot(y), rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("seed", range(24))
def test_tile_path_against_exact_path_random_configurations(cm, oracle, seed):
    """Randomised configurations of the throughput path (tile order, half angles, fixed-order P^T,
    overlap-save lists) against the exact path (time-order P, pixel-major P^T, direct or rocFFT
    Toeplitz): polarisation, map and TOD sizes, ragged noise blocks, band length, flag fraction,
    tile and slice sizes, and pointings that put a large share of the samples on a few pixels
    (long runs, tail lists).  Also run-to-run bit equality."""
    from types import SimpleNamespace
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(7000 + seed)
    pol = int(rng.integers(1, 4))
    npix = int(rng.integers(64, 6000))
    nt = int(rng.integers(3000, 90000))
    nblk = int(rng.integers(1, 6))
    cuts = np.sort(rng.choice(np.arange(1, nt), size=nblk - 1, replace=False)) if nblk > 1 else np.array([], int)
    sizes = np.diff(np.concatenate([[0], cuts, [nt]])).astype(int).tolist()
    lam = int(rng.choice([1, 2, 7, 33, 130, 300, 700]))
    pairs = rng.integers(0, npix, nt)
    if rng.random() < 0.4:                                  # hot pixels: runs far longer than a group
        hot = rng.integers(0, npix, 3)
        m = rng.random(nt) < rng.uniform(0.2, 0.7)
        pairs[m] = hot[rng.integers(0, 3, int(m.sum()))]
    if rng.random() < 0.5:                                  # coherent scan: consecutive pixels
        pairs = (np.arange(nt) // int(rng.integers(1, 9)) + int(rng.integers(0, npix))) % npix
    pairs[rng.random(nt) < rng.choice([0.0, 0.02, 0.3])] = -1
    phi = rng.uniform(0, np.pi) + 0.0785 * np.arange(nt)
    ang = SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi)) if pol > 1 else None
    P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=ang)
    L._sparse_tiles(P, tile_pixels=int(rng.choice([64, 128, 512, 1536, 4096])),
                    slice_samples=int(rng.choice([256, 1024, 4096])))
    kk = np.arange(lam)
    bands = [(1.0 + 0.1 * b) * np.exp(-kk / (lam / 3.0 + 1.0)) * np.cos(kk / (lam + 2.0)) for b in range(nblk)]
    Nf = cm.I.BlockLO(sizes, bands, offdiag=True, method=3)
    Nd = cm.I.BlockLO(sizes, bands, offdiag=True, method=(1 if lam <= 130 else 2))
    x = rng.standard_normal(pol * npix)
    exact = P.T * (Nd * (P * x))
    A = L._TiledNormalLO(P, Nf)
    y1, y2 = np.asarray(A * x), np.asarray(A * x)
    assert np.array_equal(y1, y2)
    scale = np.linalg.norm(exact)
    assert np.linalg.norm(y1 - exact) <= 1e-12 * scale + 1e-300, (seed, pol, npix, nt, sizes, lam)


@pytest.mark.parametrize("seed", range(8))
def test_tile_path_pcg_iteration_count_random_small_systems(cm, oracle, seed):
    """The tile-order path forced on small random systems (polarisation, sizes, band, flags,
    ragged noise blocks drawn per seed): PCG with M_BD must take exactly the oracle's number of
    iterations and reach its solution."""
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(8100 + seed)
    pol = int(rng.integers(1, 4))
    npix = int(rng.integers(150, 900))
    nb = int(rng.integers(1, 5))
    bs = int(rng.integers(6000, 20000))
    nt = nb * bs
    lam = int(rng.choice([2, 5, 17, 48]))
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    pairs[rng.random(nt) < rng.choice([0.0, 0.05])] = -1
    kk = np.arange(lam)
    bands = [(1.0 + 0.05 * b) * np.where(kk == 0, 1.0, -0.2 * np.exp(-kk / 5.0)) for b in range(nb)]
    po = pairs.copy()
    ro = oracle.process_time_samples(po, npix, pol=pol, phi=phi)
    ces = cm.U.ProcessTimeSamples(pairs, npix, pol=pol, phi=phi)
    n = ces.get_new_pixel[0]
    assert n == ro.new_npix and np.array_equal(pairs, po)
    P = cm.I.SparseLO(n, nt, pairs, pol=pol, angle_processed=ces)
    M = cm.I.BlockDiagonalPreconditionerLO(ces, n, pol=pol)
    N = cm.I.BlockLO(bs, bands, offdiag=True, method=3)
    c, s = ro.cos, ro.sin

    def A_o(x):
        return oracle.sparse_rmult(pol, n, po, c, s, oracle.blocklo_mult(
            bs, bands, True, oracle.sparse_mult(pol, po, c, s, x)))
    M_o = lambda x: oracle.bd_precond_mult(pol, ro, x)
    b_o = oracle.sparse_rmult(pol, n, po, c, s, oracle.blocklo_mult(bs, bands, True, d))
    L.set_pointing_mode("tiled")
    try:
        A = P.T * N * P
        assert any(isinstance(op, L._TiledNormalLO) for op in A._compiled())
        b = P.T * (N * d)
        its_g, its_o = [], []
        xg, info_g = cm.cg(A, b, M=M, rtol=1e-6, maxiter=400, callback=lambda x: its_g.append(1))
        xo, info_o = oracle.cg(A_o, b_o, M=M_o, rtol=1e-6, maxiter=400, callback=lambda x: its_o.append(1))
        assert info_g == 0 and info_o == 0
        assert len(its_g) == len(its_o), (seed, pol, npix, nt, lam, len(its_g), len(its_o))
        assert np.linalg.norm(np.asarray(xg) - xo) <= 1e-8 * np.linalg.norm(xo)
    finally:
        L.set_pointing_mode("auto")


@pytest.mark.parametrize("prog,token", [("matvec_demo", "C-ABI-OK"), ("pcg_demo", "C-PCG-OK")])
def test_c_host_program_through_the_abi(tmp_path, prog, token):
    """Plain C99 hosts (gcc, the HIP runtime's C API, no Python) on include/cosmomap2.h:
    tests/c_abi/matvec_demo.c finds P and P^T bit-identical to the reference's serial loops;
    tests/c_abi/pcg_demo.c runs the whole hot path -- weights, per-pixel blocks, tile-order
    P / Toeplitz N^-1 / P^T, M_BD and scipy's PCG recurrence with device scalars -- and checks the
    true residual of the solution."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / prog)
    libdir = os.path.join(root, "cosmomap2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", prog + ".c"),
                           "-L" + libdir, "-lcosmomap2_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert token in res.stdout, res.stdout


def test_cm2_pcg_driver_equals_python_cg(cm, oracle):
    """cm2_pcg (the C driver with callbacks) against cosmomap2_amd.cg on the same operators:
    identical iteration count, bit-identical solution, callback sequence, maxiter / info."""
    import ctypes
    from cosmomap2_amd import _hip, device as D
    A, M, b, n = _system(cm, oracle, 31, 40000, 300, 4, 3)
    bd = D.f64(b)
    its = []
    xs, info = cm.cg(A, bd, M=M, rtol=1e-8, maxiter=200, callback=lambda v: its.append(1))
    assert info == 0
    APPLY = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    ITER = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double)
    t = cm.torch

    class Raw(object):                                       # zero-copy view of a library buffer
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False),
                                             "version": 2}

    def wrap(op):
        def f(ctx, d_in, d_out, stream):
            x_in = t.as_tensor(Raw(d_in), device="cuda")
            t.as_tensor(Raw(d_out), device="cuda").copy_(D.f64(op * x_in))
            return 0
        return APPLY(f)

    seen = []
    cbA, cbM = wrap(A), wrap(M)
    cbI = ITER(lambda ctx, it, d_x, rn: seen.append((it, rn)))
    x2 = D.empty(n)
    iters, inf = ctypes.c_int64(0), ctypes.c_int(-1)
    _hip.call("cm2_pcg", n, ctypes.cast(cbA, ctypes.c_void_p), None, ctypes.cast(cbM, ctypes.c_void_p), None,
              D.ptr(bd), D.ptr(x2), 1, 1e-8, 0.0, 200, ctypes.cast(cbI, ctypes.c_void_p), None,
              ctypes.byref(iters), ctypes.byref(inf), D.stream())
    assert inf.value == 0 and iters.value == len(its) == len(seen)
    assert [s[0] for s in seen] == list(range(1, len(its) + 1))
    assert t.equal(x2, D.f64(xs))
    # maxiter reached: info = maxiter, like scipy
    x3 = D.empty(n)
    _hip.call("cm2_pcg", n, ctypes.cast(cbA, ctypes.c_void_p), None, None, None, D.ptr(bd), D.ptr(x3), 1,
              1e-14, 0.0, 2, None, None, ctypes.byref(iters), ctypes.byref(inf), D.stream())
    assert inf.value == 2 and iters.value == 2


def test_cm2_arnoldi_driver_equals_python_arnoldi(cm, oracle, golden):
    """cm2_arnoldi (C entry point, operator as a callback) against interfaces.arnoldi on the same
    operator: same number of steps, bit-identical Hessenberg columns and basis vectors; the
    reference's early exit and its failure after inner_m steps; and the reference's own executed
    arnoldi() vectors (golden arn_*) through the C driver."""
    import ctypes
    from cosmomap2_amd import _hip, device as D
    t = cm.torch
    APPLY = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)

    def run(Aop, b, n, tol, inner_m, x0=None):
        class Raw(object):
            def __init__(self, ptr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8",
                                                 "data": (int(ptr), False), "version": 2}

        def f(ctx, d_in, d_out, stream):
            x_in = t.as_tensor(Raw(d_in), device="cuda")
            t.as_tensor(Raw(d_out), device="cuda").copy_(D.f64(Aop * x_in))
            return 0
        cb = APPLY(f)
        V = D.zeros(n * inner_m)
        H = (ctypes.c_double * ((inner_m + 1) * inner_m))()
        steps = ctypes.c_int(-1)
        bd = D.f64(b)                                 # kept alive for the whole call
        x0d = D.f64(x0) if x0 is not None else None
        rc = _hip.load().cm2_arnoldi(n, ctypes.cast(cb, ctypes.c_void_p), None, D.ptr(bd),
                                     D.ptr(x0d), tol, inner_m, D.ptr(V), H, ctypes.byref(steps),
                                     D.stream())
        t.cuda.synchronize()
        return rc, V.reshape(inner_m, n), np.array(H).reshape(inner_m + 1, inner_m), steps.value

    # a small SPD matrix with a few distinct eigenvalues: the stop rule triggers after a few steps
    rng = np.random.default_rng(12)
    n = 60
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Amat = (Q * np.repeat([1.0, 2.0, 3.5, 5.0, 7.0, 9.0], 10)).dot(Q.T)
    Aop = cm.I.lp.LinearOperator(n, n, lambda v: D.f64(Amat).matmul(D.f64(v)) if D.is_tensor(v)
                               else Amat.dot(v), symmetric=True, device_ok=True)
    b = rng.standard_normal(n)
    vs, hs, j = cm.I.arnoldi(Aop, D.f64(b), tol=1e-10, inner_m=30)
    rc, V, H, steps = run(Aop, b, n, 1e-10, 30)
    assert rc == 0 and steps == j and 2 <= j < 30
    for q in range(j):
        np.testing.assert_array_equal(H[:q + 2, q], np.asarray(hs[q]))      # column q has q + 2 entries
        assert t.equal(V[q], D.f64(vs[q]))
    assert not H[j + 1:].any()
    np.testing.assert_array_equal(cm.I.build_hess(hs, j), H[:j, :j])
    # early exit (||r0|| < tol): zero steps; x0 given: r0 = b - A x0
    rc, _, _, steps = run(Aop, 1e-14 * b, n, 1e-5, 10)
    assert rc == 0 and steps == 0
    xsol = np.linalg.solve(Amat, b)
    rc, _, _, steps = run(Aop, b, n, 1e-5, 10, x0=xsol)
    assert rc == 0 and steps == 0
    # failure after inner_m steps: the reference's RuntimeError text
    rc, _, _, steps = run(Aop, b, n, 1e-300, 3)
    assert rc != 0 and steps == 3
    assert b"Convergence not achieved within the Arnoldi algorithm" in _hip.load().cm2_last_error()
    with pytest.raises(RuntimeError):
        cm.I.arnoldi(Aop, D.f64(b), tol=1e-300, inner_m=3)
    # the reference's own executed arnoldi() on its golden system
    Ag, bg = golden["arn_A"], golden["arn_b"]
    ng = bg.size
    Ago = cm.I.lp.LinearOperator(ng, ng, lambda v: D.f64(Ag).matmul(D.f64(v)) if D.is_tensor(v) else Ag.dot(v),
                               symmetric=True, device_ok=True)
    jg = int(golden["arn_j"])
    rc, V, H, steps = run(Ago, bg, ng, 1e-8, 30)          # the golden call: tol 1e-8, inner_m 30
    assert rc == 0 and steps == jg
    np.testing.assert_allclose(V[:jg].cpu().numpy(), golden["arn_V"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(H[:jg, :jg], golden["arn_H"], rtol=0, atol=1e-9)


def test_cm2_PtNP_tiles_apply_is_the_three_calls(cm, oracle):
    """cm2_PtNP_tiles_apply (one call for the tile-order chain) gives the bits of the three
    separate calls."""
    from types import SimpleNamespace
    from cosmomap2_amd import _hip, device as D
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(77)
    pol, npix, nt, nblk, lam = 3, 3000, 200000, 4, 50
    pairs = rng.integers(0, npix, nt)
    pairs[rng.random(nt) < 0.03] = -1
    phi = 0.3 + 0.0785 * np.arange(nt)
    P = cm.I.SparseLO(npix, nt, pairs, pol=pol,
                      angle_processed=SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi)))
    T = L._sparse_tiles(P, tile_pixels=512, slice_samples=4096)
    kk = np.arange(lam)
    N = cm.I.BlockLO(nt // nblk, [(1.0 + 0.1 * b) * np.exp(-kk / 9.0) for b in range(nblk)],
                     offdiag=True, method=3)
    x = D.f64(rng.standard_normal(pol * npix))
    want = D.f64(L._TiledNormalLO(P, N) * x)
    y, w1, w2 = D.empty(pol * npix), D.empty(T.nvalid), D.empty(T.nvalid)
    y.fill_(5.0)
    _hip.call("cm2_PtNP_tiles_apply", T.h, N._noise.h, D.ptr(x), D.ptr(y), D.ptr(w1), D.ptr(w2), D.stream())
    assert cm.torch.equal(y, want)
    with pytest.raises(_hip.HipError):
        _hip.call("cm2_PtNP_tiles_apply", T.h, N._noise.h, D.ptr(x), D.ptr(y), D.ptr(w1), D.ptr(w1), D.stream())


@pytest.mark.parametrize("angles", ["half", "full"])
@pytest.mark.parametrize("nt,npix,tp,hot,slice_len", [(300000, 5000, 2048, 0.0, 1280), (50000, 100, 64, 0.0, 1280),
                                                      (400000, 70000, 1024, 0.0, 1280), (40000, 20, 64, 0.0, 1280),
                                                      (300000, 40000, 1024, 0.05, 1280), (4097, 64, 64, 0.0, 1280),
                                                      (50000, 100, 64, 0.0, 64), (300000, 5000, 2048, 0.0, 2048),
                                                      (40000, 20, 64, 0.0, 2048), (120000, 9000, 1024, 0.05, 192)])
def test_fixed_order_lists_built_per_slice_in_lds_give_the_serial_packers_sums(cm, oracle, monkeypatch, nt,
                                                                             npix, tp, hot, slice_len, angles):
    """The fixed-order P^T lists are built by one workgroup per slice (k_fx_build: bitonic sort in
    LDS, runs placed by class from scans) instead of a global radix sort and a one-thread-per-slice
    walk (CM2_FX_BUILD=serial).  The packing differs, the sums may not: P^T is bit-identical between
    the two, for singles-dominated slices, runs of 5 .. 60 (levels, rows of 64 groups), runs beyond 60
    (tail lists), a hot pixel (chunk sums), both angle storages, and equal to the oracle's serial loop
    bit for bit where that is promised (both angle arrays, pure time order)."""
    from types import SimpleNamespace
    from cosmomap2_amd import _hip, device as D
    from cosmomap2_amd.interfaces import linearoperators as L
    pol = 3
    monkeypatch.setenv("CM2_TILE_ANGLES", angles)
    # (the slice length is tuned to the groups a slice packs into, which differ between the two
    #  packers; the chunk boundaries of a hot run follow the slice length: same length for both)
    monkeypatch.setenv("CM2_PT_SLICE", str(slice_len))      # (64 and 2048: the shortest and longest slices)
    rng = np.random.default_rng(99)
    pairs = rng.integers(0, npix, nt)
    if hot:
        pairs[rng.random(nt) < hot] = npix // 3
    pairs[rng.random(nt) < 0.05] = -1
    phi = rng.uniform(0, np.pi, nt)
    c, s = np.cos(2 * phi), np.sin(2 * phi)
    v = rng.standard_normal(nt)
    outs, groups = [], []
    for build in ("lds", "serial"):
        monkeypatch.setenv("CM2_FX_BUILD", build)
        P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=SimpleNamespace(cos=c, sin=s))
        T = L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)
        assert T.pt_fixed
        st = D.stream()
        v_tb, out = D.empty(T.nvalid), D.empty(pol * npix)
        vd = D.f64(v)
        _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(vd), D.ptr(v_tb), st)
        for order in (True, "exact"):
            T.set_pt_order(order)
            out.fill_(5.0)
            _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st)
            outs.append(out.cpu().numpy().copy())
    np.testing.assert_array_equal(outs[0], outs[2])          # fixed order (chunk sums for hot runs)
    np.testing.assert_array_equal(outs[1], outs[3])          # pure time order
    if angles == "full":
        np.testing.assert_array_equal(outs[1], oracle.sparse_rmult(pol, npix, pairs, c, s, v))


@pytest.mark.parametrize("pol", [1, 2, 3])
def test_weights_of_a_hot_pixel_are_summed_in_fixed_chunks(cm, oracle, monkeypatch, pol):
    """ProcessTimeSamples' per-pixel sums (process_ces.py:480-539) are the reference's serial sums
    bit for bit -- except for a pixel with 8192 samples or more, which one thread would walk for
    seconds (1.9 s for 5 % of 1e8 samples): its sum is regrouped into fixed chunks of 4096 (thread
    sums + halving tree + chunks in time order), reproducible and within 1e-13 of the serial sum;
    CM2_WEIGHTS_ORDER=exact keeps the serial walk.  Two hot pixels here (105 k and 9 k samples)."""
    nt, npix = 1 << 21, 5000
    rng = np.random.default_rng(31)
    pairs = rng.integers(0, npix, nt)
    pairs[rng.random(nt) < 0.05] = 1234
    pairs[rng.random(nt) < 0.004] = 77
    pairs[rng.random(nt) < 0.02] = -1
    pairs = pairs.astype(np.int32)
    phi = rng.uniform(0, np.pi, nt)
    w = rng.random(nt)
    keys = {1: ("counts",), 2: ("cos2", "sin2", "sincos"),
            3: ("counts", "cosine", "sine", "cos2", "sin2", "sincos")}[pol]
    ro = oracle.process_time_samples(pairs.copy(), npix, pol=pol, phi=phi, w=w)
    hot = np.isin(ro.obspix, [1234, 77])
    assert hot.sum() == 2
    runs = []
    for _ in range(2):
        rg = cm.U.ProcessTimeSamples(pairs.astype(np.int64), npix, pol=pol, phi=phi, w=w)
        np.testing.assert_array_equal(rg.get_new_pixel[1], ro.obspix)
        runs.append({k: np.asarray(getattr(rg, k)).copy() for k in keys})
    for k in keys:
        np.testing.assert_array_equal(runs[0][k], runs[1][k], err_msg=k)           # reproducible
        np.testing.assert_array_equal(runs[0][k][~hot], getattr(ro, k)[~hot], err_msg=k)
        ref = getattr(ro, k)[hot]
        assert np.all(np.abs(runs[0][k][hot] - ref) <= 1e-13 * np.abs(ref).max()), k
    monkeypatch.setenv("CM2_WEIGHTS_ORDER", "exact")
    rg = cm.U.ProcessTimeSamples(pairs.astype(np.int64), npix, pol=pol, phi=phi, w=w)
    for k in keys:
        np.testing.assert_array_equal(getattr(rg, k), getattr(ro, k), err_msg=k)
    # unit weights: the hit counts are exact in either order
    if pol != 2:
        monkeypatch.delenv("CM2_WEIGHTS_ORDER")
        r1 = cm.U.ProcessTimeSamples(pairs.astype(np.int64), npix, pol=pol, phi=phi)
        np.testing.assert_array_equal(r1.counts, oracle.process_time_samples(pairs.copy(), npix, pol=pol,
                                                                             phi=phi).counts)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_problems_every_plan_builder_and_list_format_give_the_same_bits(cm, monkeypatch, seed):
    """Random shapes (3 000 .. 600 000 samples, 50 .. 120 000 pixels, tiles of 64 .. 2048 pixels, 1 .. 6
    ragged noise blocks, lambda 2 .. 2049, up to 30 % flagged, sometimes a hot pixel or a tenth of the
    map holding half of the samples): P^T N^-1 P on the tile order is bit-identical between the
    sort-free plan builders and the sorted / serial ones, between run-coded lists cut by time and
    inverse lists cut by address, and between uniform and balanced tiles (pure time order, so that a
    hot run's chunk boundaries do not enter)."""
    from types import SimpleNamespace
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(seed)
    monkeypatch.setenv("CM2_PT_ORDER", "exact")
    builds = (("new", {}), ("old", {"CM2_TILE_BUILD": "sort", "CM2_OS_LIST_BUILD": "sort",
                                    "CM2_FX_BUILD": "serial", "CM2_OS_LISTS": "rc"}),
              ("inv", {"CM2_OS_LISTS": "inv"}), ("bal", {"CM2_TILE_BALANCE": "1"}))
    for case in range(6):
        pol = int(rng.choice([1, 2, 3]))
        nt = int(rng.integers(3000, 600000))
        npix = int(rng.integers(50, 120000))
        tp = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
        nblk = int(rng.integers(1, 7))
        cuts = np.sort(rng.choice(np.arange(1, nt), nblk - 1, replace=False)) if nblk > 1 else np.array([], int)
        sizes = np.diff(np.concatenate([[0], cuts, [nt]])).tolist()
        lam = int(rng.choice([2, 17, 300, 2049]))
        pairs = rng.integers(0, npix, nt)
        if rng.random() < 0.4:
            pairs[rng.random(nt) < rng.uniform(0.01, 0.2)] = int(rng.integers(0, npix))
        if rng.random() < 0.3:
            h = rng.random(nt) < 0.5
            pairs[h] = pairs[h] % max(npix // 10, 1)
        pairs[rng.random(nt) < rng.uniform(0, 0.3)] = -1
        phi = rng.uniform(0, np.pi, nt)
        ang = SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi))
        kk = np.arange(lam)
        bands = [(1.0 + 0.1 * b) * np.exp(-kk / max(lam / 4.0, 1.0)) for b in range(nblk)]
        x = rng.standard_normal(pol * npix)
        outs = {}
        for name, env in builds:
            for k in ("CM2_TILE_BUILD", "CM2_OS_LIST_BUILD", "CM2_FX_BUILD", "CM2_OS_LISTS", "CM2_TILE_BALANCE"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=ang)
            L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)
            N = cm.I.BlockLO(sizes, bands, offdiag=True, method=3)
            outs[name] = np.asarray(L._TiledNormalLO(P, N) * x)
        for name in ("old", "inv", "bal"):
            np.testing.assert_array_equal(outs["new"], outs[name],
                                          err_msg="%s: %r" % (name, dict(pol=pol, nt=nt, npix=npix, tp=tp,
                                                                         sizes=sizes, lam=lam)))


def test_library_device_memory_is_cached_and_released(cm):
    """Plans and their build temporaries come from the library's cache of released device blocks
    (cm2_core.hip): a second build of the same plan is served from the cache, live bytes return to
    where they were once the plan is destroyed, and cm2_release_cached_memory hands everything
    back to the driver.  Reused blocks are not zeroed -- the operator built on them must equal the
    first one bit for bit."""
    import gc
    from types import SimpleNamespace
    from cosmomap2_amd import device as D
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(5)
    nt, npix, pol = 200000, 30000, 3
    pairs = rng.integers(0, npix, nt)
    pairs[rng.random(nt) < 0.05] = -1
    phi = rng.uniform(0, np.pi, nt)
    ang = SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi))
    kk = np.arange(200)
    bands = [np.exp(-kk / 40.0)] * 4
    x = rng.standard_normal(pol * npix)

    def build_and_apply():
        P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=ang)
        L._sparse_tiles(P, tile_pixels=1024, slice_samples=4096)
        N = cm.I.BlockLO([nt // 4] * 4, bands, offdiag=True, method=3)
        return np.asarray(L._TiledNormalLO(P, N) * x)

    gc.collect()
    D.release_cached_memory()
    base = D.memory_info()
    assert base["cached_bytes"] == 0
    y0 = build_and_apply()
    gc.collect()
    cm.torch.cuda.synchronize()
    after = D.memory_info()
    assert after["live_bytes"] == base["live_bytes"]           # everything the build took is back
    assert after["cached_bytes"] > 0
    y1 = build_and_apply()
    gc.collect()
    again = D.memory_info()
    assert again["cache_hits"] > after["cache_hits"]
    assert again["driver_allocations"] - after["driver_allocations"] <= 2
    np.testing.assert_array_equal(y0, y1)
    D.release_cached_memory()
    assert D.memory_info()["cached_bytes"] == 0


@pytest.mark.parametrize("nt,npix,tp,balance", [(400003, 70000, 1024, "0"), (50001, 100, 64, "0"),
                                                (300000, 262144, 64, "0"), (400003, 70000, 1024, "1"),
                                                (8191, 3000, 64, "0"), (70, 500, 64, "0")])
def test_tile_plan_without_a_sort_equals_the_sorted_plan(cm, oracle, monkeypatch, nt, npix, tp, balance):
    """The tile order is built as a stable multisplit (k_tile_rank: rank of a sample among the
    earlier samples of its tile inside a chunk of 8192, per-chunk counts, one scan; k_tile_place)
    instead of a radix sort + gather (CM2_TILE_BUILD=sort).  Both must give the same addresses:
    the time -> tile-order map, the per-sample words and angles (seen through P and P^T on the tile
    order, bit for bit), for ragged sizes, 10 % flagged samples, 1 .. 4096 tiles and re-cut
    (balanced) tiles."""
    from types import SimpleNamespace
    from cosmomap2_amd import _hip, device as D
    from cosmomap2_amd.interfaces import linearoperators as L
    pol = 3
    rng = np.random.default_rng(77)
    pairs = rng.integers(0, npix, nt)
    if balance == "1":
        pairs[: nt // 2] = rng.integers(0, npix // 10, nt // 2)      # half of the samples on a tenth
    pairs[rng.random(nt) < 0.1] = -1
    phi = rng.uniform(0, np.pi, nt)
    c, s = np.cos(2 * phi), np.sin(2 * phi)
    x = rng.standard_normal(pol * npix)
    v = rng.standard_normal(nt)
    monkeypatch.setenv("CM2_TILE_BALANCE", balance)
    got = []
    for build in ("split", "sort"):
        monkeypatch.setenv("CM2_TILE_BUILD", build)
        P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=SimpleNamespace(cos=c, sin=s))
        T = L._sparse_tiles(P, tile_pixels=tp, slice_samples=4096)
        st = D.stream()
        lab = cm.torch.empty(max(T.nvalid, 1), dtype=cm.torch.int32, device="cuda")
        tix = cm.torch.arange(nt, dtype=cm.torch.int32, device="cuda")
        _hip.call("cm2_i32_time_to_tiles", T.h, D.ptr(tix), D.ptr(lab), st)
        d_tb, out = D.empty(T.nvalid), D.empty(pol * npix)
        _hip.call("cm2_P_tiles_apply", T.h, D.ptr(D.f64(x)), D.ptr(d_tb), st)
        v_tb = D.empty(T.nvalid)
        vd = D.f64(v)
        _hip.call("cm2_tod_time_to_tiles", T.h, D.ptr(vd), D.ptr(v_tb), st)
        _hip.call("cm2_Pt_tiles_apply", T.h, D.ptr(v_tb), D.ptr(out), st)
        got.append((T.ntiles, T.nvalid, lab.cpu().numpy()[: T.nvalid].copy(), d_tb.cpu().numpy().copy(),
                    out.cpu().numpy().copy()))
    a, b = got
    assert a[0] == b[0] and a[1] == b[1] == int((pairs >= 0).sum())
    for u, w in zip(a[2:], b[2:]):
        np.testing.assert_array_equal(u, w)
    # and the order is the stable one: inside a tile the time indices ascend
    tmap = a[2]
    assert np.array_equal(np.sort(tmap), np.flatnonzero(pairs >= 0))


def test_uneven_hit_map_gets_balanced_tiles_and_the_same_bits(cm, oracle, monkeypatch):
    """Half of the samples on a tenth of the map.  CM2_TILE_BALANCE=cut re-cuts the pixel ranges to
    equal sample counts (the fixed-order P^T gives a tile to one workgroup): P^T N^-1 P does not
    change by a bit against the uniform tiling.  The default (round 4) keeps the uniform tiles and
    shares the slices of the heavy ones out to several workgroups whose tile copies are added in time
    order: a fixed regrouping, within 1e-14 of the serial order.  The groups of tiles a sharded run
    reduces cover the same pixels in every plan."""
    import ctypes
    from types import SimpleNamespace
    from cosmomap2_amd import _hip
    from cosmomap2_amd.interfaces import linearoperators as L
    rng = np.random.default_rng(99)
    pol, nside, nt, nblk, lam = 3, 64, 1 << 21, 4, 40
    npix = 12 * nside * nside
    pairs = rng.integers(0, npix, nt)
    hot = rng.random(nt) < 0.5
    pairs[hot] = pairs[hot] % (npix // 10)
    pairs[rng.random(nt) < 0.02] = -1
    phi = 0.3 + 0.0785 * np.arange(nt)
    ang = SimpleNamespace(cos=np.cos(2 * phi), sin=np.sin(2 * phi))
    kk = np.arange(lam)
    bands = [(1.0 + 0.1 * b) * np.exp(-kk / 9.0) for b in range(nblk)]
    x = rng.standard_normal(pol * npix)
    out, tiles, groups, split = {}, {}, {}, {}
    monkeypatch.setenv("CM2_PT_PARTS", "9000")      # (parts of 9000 samples: the automatic choice needs
    for mode in ("0", "cut", None):                 #  more samples per workgroup than this problem has)
        if mode is None:
            monkeypatch.delenv("CM2_TILE_BALANCE", raising=False)
        else:
            monkeypatch.setenv("CM2_TILE_BALANCE", mode)
        P = cm.I.SparseLO(npix, nt, pairs, pol=pol, angle_processed=ang)
        T = L._sparse_tiles(P, tile_pixels=128, slice_samples=4096)
        N = cm.I.BlockLO(nt // nblk, bands, offdiag=True, method=3)
        out[mode] = np.asarray(L._TiledNormalLO(P, N) * x)
        tiles[mode] = T.ntiles
        split[mode] = T.pt_parts()
        cuts = (ctypes.c_int64 * 5)()
        _hip.call("cm2_tiles_group_tiles", T.h, 4, cuts)
        groups[mode] = [T.pixel_range(int(cuts[g]), int(cuts[g + 1])) for g in range(4)]
    assert tiles["0"] == npix // 128 and tiles["cut"] != tiles["0"] and tiles[None] == tiles["0"]
    assert np.array_equal(out["0"], out["cut"])
    assert split["0"]["tiles_split"] == 0 and split["cut"]["tiles_split"] == 0
    assert split[None]["tiles_split"] >= 30 and split[None]["workgroups"] > tiles[None] + 60, split[None]
    assert np.abs(out[None] - out["0"]).max() <= 1e-14 * np.abs(out["0"]).max()
    assert groups["0"] == groups["cut"] == groups[None] and groups[None][0][0] == 0 and groups[None][-1][1] == npix
